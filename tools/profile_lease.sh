#!/bin/bash
# One lease, one box, one binary: the driver's bench command, then rocprofv3 --kernel-trace --stats and the PMC passes
# (each in its own run, program directly after "--") on the same workload, summarised into profiles/<tag>_*.
#   tools/grun.sh --timeout 900 -- 'tools/profile_lease.sh r03'                 (the bench workload, config 3)
#   tools/grun.sh --timeout 900 -- 'tools/profile_lease.sh r03 8 c5 262144'     (another BASELINE shape: tag, steps, workload, reads per launch)
# Writes gpurun_out/lease_<tag>/ (raw) and gpurun_out/lease_<tag>/profiles/ (the files to copy into profiles/).
set -u
TAG=${1:-rXX}
STEPS=${2:-20}
WL=${3:-c3}
READS=${4:-1048576}
SFX=""; [ "$WL" != c3 ] && SFX="_$WL"
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/lease_$TAG$SFX
P=$OUT/profiles
mkdir -p $P
export TMPDIR=/tmp
export TD_SPEC_CACHE_DIR=/tmp/td_spec_cache
mkdir -p $TD_SPEC_CACHE_DIR
cd $ROOT
echo "== bench (the driver's command)"
if [ "$WL" = c3 ]; then BENCH_ARGS=""; else BENCH_ARGS="--workload $WL --reads $READS --extras 0 --cpu-sample 0"; fi
timeout -k 10 500 python3 bench.py --gpus 1 --steps $STEPS --warmup 5 $BENCH_ARGS > $P/${TAG}${SFX}_bench_line.json 2> $OUT/bench.err || { echo "bench failed"; tail -5 $OUT/bench.err; exit 1; }
PROF_ARGS="--workload $WL --reads $READS --steps $STEPS --warmup 5 --extras 0 --cpu-sample 0 --check 0"
run() { local name=$1; shift
	timeout -k 10 300 rocprofv3 "$@" --output-format csv -d $OUT/$name -o td -- python3 $ROOT/bench.py $PROF_ARGS > $OUT/$name.log 2>&1 \
		|| { echo "rocprofv3 pass $name failed"; tail -5 $OUT/$name.log; return 1; }
	echo "pass $name done"; }
# The bench pipeline overlaps consecutive decode launches (two streams); a launch's traced duration then includes the time it
# shares the machine with its neighbour.  The passes the per-launch figures come from run with TD_OVERLAP=0 (one stream, launches
# one after the other -- what bench.py's isolated kernel_ms measures); the overlapping pipeline is traced once more beside them.
export TD_OVERLAP=0
run trace --kernel-trace --stats &&
run fetch --pmc FETCH_SIZE &&
run write --pmc WRITE_SIZE &&
run pmc1 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS &&
run pmc2 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS &&
run pmc3 --pmc SQ_INSTS_FLAT SQ_INSTS_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT GRBM_GUI_ACTIVE
export TD_OVERLAP=1
run trace_ov --kernel-trace
# what roofline.kernel_ms is: isolated launches on a resident batch, nothing else on the device -- traced by itself, so that
# rocprofv3's own duration of the dominant kernel can be held against the HIP-event figure of the bench line
PROF_ARGS="--workload $WL --reads $READS --isolated 20 --warmup 0 --check 0"
run trace_iso --kernel-trace --stats
python3 tools/summarize_lease.py $OUT $TAG $WL $READS
ls $P
