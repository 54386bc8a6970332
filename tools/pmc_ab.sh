#!/bin/bash
# Instruction-mix counters for one build/knob setting: tools/pmc_ab.sh <tag>   (knobs via environment)
set -u
TAG=${1:-x}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmcab_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $ROOT
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES \
	--output-format csv -d $OUT -o td -- python3 $ROOT/bench.py --reads 262144 --steps 3 --warmup 1 --cpu-sample 0 --check 0 > $OUT/run.log 2>&1 || { echo failed; tail -3 $OUT/run.log; exit 1; }
python3 - <<PY
import csv, glob, collections
tot = collections.Counter(); n = 0
for f in glob.glob("$OUT/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "td_spec_kernel" in r.get("Kernel_Name", ""):
            tot[r["Counter_Name"]] += float(r["Counter_Value"]); n += 1
print("$TAG", n, {k: "%.4g" % v for k, v in sorted(tot.items())})
PY
grep -o '"value": [0-9.]*' $OUT/run.log | head -1
