#!/bin/bash
# Unit-busy counters for td_spec_kernel on the bench workload: tools/pmc_busy.sh <tag>
set -u
TAG=${1:-x}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmcbusy_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $ROOT
pass() {
	local name=$1; shift
	timeout -k 10 300 rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -o td -- python3 $ROOT/bench.py --reads 1048576 --steps 2 --warmup 1 --cpu-sample 0 --check 0 > $OUT/$name.log 2>&1 || { echo "$name failed"; tail -3 $OUT/$name.log; return 1; }
}
pass a GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_ANY &&
pass b GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_THREAD_CYCLES_VALU

python3 - <<PY
import csv, glob, collections
for name in "ab":
    tot = collections.Counter(); cnt = collections.Counter()
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % name, recursive=True):
        for r in csv.DictReader(open(f)):
            if "td_spec_kernel" in r.get("Kernel_Name", ""):
                tot[r["Counter_Name"]] += float(r["Counter_Value"]); cnt[r["Counter_Name"]] += 1
    print("$TAG", name, {k: "%.4g (n=%d)" % (v, cnt[k]) for k, v in sorted(tot.items())})
PY
