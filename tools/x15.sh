#!/bin/bash
# kernel statistics of bench --config N for several libs: tools/x15.sh <config> <lib> ...
cfg=$1; shift
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for lib in "$@"; do
  out=$R/gpurun_out/prof_${lib%.so}
  OFDG_LIB=$R/optical-flow-2d-data-generation_amd/lib/$lib rocprofv3 --kernel-trace --stats --output-format csv -d $out -o t -- python3 $R/bench.py --config $cfg --steps 300 --no-cpu-baseline > $out.json 2>/dev/null
  echo "== $lib: $(python3 -c "import json;d=json.load(open('$out.json'));print('%.0f samples/s %.1f us/step'%(d['value'],d['ms_per_step']*1e3))")"
  python3 $R/tools/kstats.py $out | grep -v "pool_synth\|wf_\|elementwise\|fill\|Memcpy\|vectorized"
done
