#!/bin/bash
# background_prep (the CImg chain per sample): kernel statistics and step rate, several libs on one box
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for lib in "$@"; do
  out=$R/gpurun_out/prof_bg_${lib%.so}
  OFDG_LIB=$R/optical-flow-2d-data-generation_amd/lib/$lib rocprofv3 --kernel-trace --stats --output-format csv -d $out -o t -- python3 $R/bench.py --background-prep --steps 300 --no-cpu-baseline > $out.json 2>/dev/null
  python3 $R/tools/kstats.py $out | grep "bgprep\|compose" | sed "s/^/$lib: /"
  for r in 1 2; do
  echo "$lib: $(OFDG_LIB=$R/optical-flow-2d-data-generation_amd/lib/$lib python3 $R/bench.py --background-prep --steps 800 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys;d=json.loads(sys.stdin.read());print('%.0f samples/s %.1f us/step'%(d['value'],d['ms_per_step']*1e3))")"
  done
done
