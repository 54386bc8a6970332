#!/bin/bash
# Profile the bench workload with rocprofv3 on the GPU box (run through gpurun):
#   tools/profile_gpu.sh <tag> [bench args...]
# Writes gpurun_out/prof_<tag>/{trace,pmc1,pmc2,fetch,write}/ ; summarise with tools/summarize_prof.py.
# Counters go in their own passes (never combined with --sys-trace etc.), program directly after "--".
set -u
TAG=${1:-run}; shift || true
ARGS=${@:---reads 262144 --steps 3 --warmup 1 --cpu-sample 0 --check 0}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $ROOT
run() { # name, rocprof options...
	local name=$1; shift
	timeout -k 10 400 rocprofv3 "$@" --output-format csv -d $OUT/$name -o td -- python3 $ROOT/bench.py $ARGS > $OUT/$name.log 2>&1 \
		|| { echo "rocprofv3 pass $name failed"; tail -5 $OUT/$name.log; return 1; }
	echo "pass $name done"
}
run trace --kernel-trace --stats &&
run pmc1 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS &&
run pmc2 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS &&
run fetch --pmc FETCH_SIZE &&
run write --pmc WRITE_SIZE
find $OUT -name "*.csv" | head -30
