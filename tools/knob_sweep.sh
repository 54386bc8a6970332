#!/bin/bash
# A/B of kernel build knobs inside one lease (same box): tools/knob_sweep.sh "<VAR=val ...>" "<VAR=val ...>" ...
# each argument is one configuration (environment assignments); prints kernel_ms of the bench workload for each.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
mkdir -p gpurun_out/sweep
export TD_SPEC_CACHE_DIR=/tmp/td_spec_cache; mkdir -p $TD_SPEC_CACHE_DIR
i=0
WL=${SWEEP_WORKLOAD:-c3}
for cfg in "$@"; do
	i=$((i+1))
	env $cfg timeout -k 10 200 python3 bench.py --workload $WL --steps 6 --warmup 2 --extras 0 --cpu-sample 0 --check 512 --reads ${SWEEP_READS:-1048576} > gpurun_out/sweep/$i.json 2> gpurun_out/sweep/$i.err
	python3 - "$cfg" gpurun_out/sweep/$i.json gpurun_out/sweep/$i.err <<'PY'
import json, sys
cfg, f, e = sys.argv[1:4]
try:
    d = json.load(open(f))
    print("%-90s kernel_ms %.2f  Mreads/s(kernel) %.2f  host-incl %.2f" % (cfg, d["roofline"]["kernel_ms"], d["roofline"]["kernel_reads_per_s"] / 1e6, d["value"] / 1e6))
except Exception as ex:
    print("%-90s FAILED: %s" % (cfg, open(e).read()[-300:].replace("\n", " | ")))
PY
done
