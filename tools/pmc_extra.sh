#!/bin/bash
# Extra PMC passes for td_spec_kernel: memory/LDS latency levels and instruction-cache behaviour.
#   tools/pmc_extra.sh <tag> [bench args...]
set -u
TAG=${1:-x}; shift || true
ARGS=${@:---reads 262144 --steps 3 --warmup 1 --cpu-sample 0 --check 0}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmcx_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $ROOT
run() {
	local name=$1; shift
	timeout -k 10 300 rocprofv3 "$@" --output-format csv -d $OUT/$name -o td -- python3 $ROOT/bench.py $ARGS > $OUT/$name.log 2>&1 \
		|| { echo "rocprofv3 pass $name failed"; tail -5 $OUT/$name.log; return 1; }
	echo "pass $name done"
}
run lat --pmc SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES &&
run icache --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_IFETCH_LEVEL SQ_ACTIVE_INST_VALU SQ_INSTS_VALU &&
run misc --pmc SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_LDS_MEM_VIOLATIONS
python3 - <<PY
import csv, glob, collections
for name in ("lat", "icache", "misc"):
    tot = collections.Counter()
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % name, recursive=True):
        for r in csv.DictReader(open(f)):
            if "td_spec_kernel" in r.get("Kernel_Name", ""):
                tot[r["Counter_Name"]] += float(r["Counter_Value"])
    print(name, {k: "%.4g" % v for k, v in tot.items()})
PY
