#!/bin/bash
# Round profile, everything the bench line's numbers are backed by:
#   bench line (default = config 2 with the reference's per-sample background preparation, 2000 steps, with cpu_baseline and the
#   centre-crop secondary), the driver's form of it (--gpus 1 --steps 20 --warmup 5), the other BASELINE configurations,
#   rocprofv3 kernel statistics of the default command, and HBM traffic of the dominant kernel from SEPARATE --pmc FETCH_SIZE /
#   WRITE_SIZE passes (config 2 with background_prep 1 and 0, config 3).
# Usage (on the GPU box): bash tools/profile_round.sh r05 [bench|prof|all]      -> gpurun_out/<tag>/
: ${GRAFT_REPO_ROOT:?}  # (set by gpurun; refuse to run from an unknown place)
cd "$GRAFT_REPO_ROOT" || exit 1
tag=${1:-r05}
part=${2:-all}   # bench: the bench lines; prof: the rocprofv3 passes (each fits one gpurun call); all: both
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
if [ "$part" != "prof" ]; then
echo "[bench] default"; timeout -k 10 500 python3 bench.py > $out/bench_line.json 2> $out/bench_stderr.txt
echo "[bench] driver-like"; timeout -k 10 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_line_driver_like_20_steps.json 2>> $out/bench_stderr.txt
for cfg in 1 3 4 5; do
  echo "[bench] config $cfg"; timeout -k 10 400 python3 bench.py --config $cfg > $out/bench_line_config$cfg.json 2>> $out/bench_stderr.txt
done
echo "[bench] centre crops as the headline"; timeout -k 10 300 python3 bench.py --background-prep 0 --no-cpu-baseline --no-secondary > $out/bench_line_config2_centre_crop.json 2>> $out/bench_stderr.txt
echo "[bench] resident"; timeout -k 10 300 python3 bench.py --sampler resident --no-cpu-baseline --no-secondary > $out/bench_line_config2_resident.json 2>> $out/bench_stderr.txt
fi
if [ "$part" != "bench" ]; then
cd /tmp && export TMPDIR=/tmp
echo "[rocprofv3] kernel trace"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o t -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-secondary > $out/bench_line_under_rocprof.json 2>/dev/null
for arm in "2 1" "2 0" "3 1"; do
  set -- $arm; cfg=$1; bgp=$2
  echo "[rocprofv3] pmc config $cfg background_prep $bgp"
  timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch_c${cfg}_p$bgp -o p -- python3 $GRAFT_REPO_ROOT/bench.py --config $cfg --background-prep $bgp --no-cpu-baseline --no-secondary --steps 60 > /dev/null 2>&1
  timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_write_c${cfg}_p$bgp -o p -- python3 $GRAFT_REPO_ROOT/bench.py --config $cfg --background-prep $bgp --no-cpu-baseline --no-secondary --steps 60 > /dev/null 2>&1
done
echo "[rocprofv3] pmc instruction counts, config 2 background_prep 1"
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --output-format csv -d $out/pmcdir_valu_c2_p1 -o p -- python3 $GRAFT_REPO_ROOT/bench.py --config 2 --background-prep 1 --no-cpu-baseline --no-secondary --steps 60 > /dev/null 2>&1
cd "$GRAFT_REPO_ROOT"
python3 tools/pmcstats.py $out/pmcdir_valu_c2_p1 > $out/pmc_valu_config2_background_prep_1.txt
python3 tools/kstats.py $out/trace > $out/kernel_stats.txt
cp $(ls $out/trace/*kernel_stats.csv $out/trace/*/*kernel_stats.csv 2>/dev/null | head -1) $out/kernel_stats.csv 2>/dev/null
for arm in "2 1" "2 0" "3 1"; do
  set -- $arm; cfg=$1; bgp=$2
  python3 tools/pmcstats.py $out/pmc_fetch_c${cfg}_p$bgp > $out/pmc_fetch_size_config${cfg}_background_prep_$bgp.txt
  python3 tools/pmcstats.py $out/pmc_write_c${cfg}_p$bgp > $out/pmc_write_size_config${cfg}_background_prep_$bgp.txt
done
rm -rf $out/trace $out/pmc_fetch_c* $out/pmc_write_c* $out/pmcdir_valu_c*
tail -n 12 $out/kernel_stats.txt; grep -A3 compose $out/pmc_fetch_size_config2_background_prep_1.txt $out/pmc_write_size_config2_background_prep_1.txt
fi
cd "$GRAFT_REPO_ROOT"
python3 -c "
import json,glob
for f in sorted(glob.glob('$out/bench_line*.json')):
    try: d=json.load(open(f))
    except Exception as e: print(f, 'unreadable', e); continue
    r=d.get('centre_crop_backgrounds') or d.get('reference_equivalent') or {}
    print('%-55s prep %d %9.0f samples/s %7.1f us/step frac %.3f | other mode %s' % (f.split('/')[-1], d['config']['background_prep'], d['value'], d['ms_per_step']*1e3, d['roofline']['whole_step_frac'], ('%.0f' % r['value']) if r else '-'))"
