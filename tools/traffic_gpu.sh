#!/bin/bash
# HBM traffic (FETCH_SIZE / WRITE_SIZE passes) and kernel time of the bench workload: tools/traffic_gpu.sh <tag> [bench args]
set -u
TAG=${1:-x}; shift || true
ARGS=${@:---reads 1048576 --steps 3 --warmup 1 --cpu-sample 0 --check 0}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/traffic_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $ROOT
run() { local name=$1; shift
	timeout -k 10 400 rocprofv3 "$@" --output-format csv -d $OUT/$name -o td -- python3 $ROOT/bench.py $ARGS > $OUT/$name.log 2>&1 || { echo "$name failed"; tail -3 $OUT/$name.log; return 1; }; }
run trace --kernel-trace --stats && run fetch --pmc FETCH_SIZE && run write --pmc WRITE_SIZE
python3 tools/summarize_prof.py $OUT | python3 -c "
import sys, json
s = json.load(sys.stdin)
h = s.get('hbm_bytes_per_launch', {}); k = s.get('kernel_trace', {})
print('$TAG', 'kernel_ms %.2f' % k.get('avg_ms', 0), 'traffic_GB %.1f' % (h.get('total', 0) / 1e9), 'TB/s %.2f' % (h.get('total', 0) / 1e9 / max(k.get('avg_ms', 1), 1e-9)))"
