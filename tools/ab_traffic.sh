#!/bin/bash
# Same-box A/B of two kernel builds: kernel time (HIP events, unprofiled) and HBM bytes per read (FETCH_SIZE x2 + WRITE_SIZE,
# separate rocprofv3 --pmc passes) for each environment setting.
#   tools/ab_traffic.sh <tag> "<VAR=val ...>" "<VAR=val ...>" [workload] [reads]
TAG=$1; A=$2; B=$3; WL=${4:-c3}; N=${5:-1048576}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/ab_$TAG
mkdir -p $OUT
export TMPDIR=/tmp TD_SPEC_CACHE_DIR=/tmp/td_spec_cache
mkdir -p $TD_SPEC_CACHE_DIR
cd $ROOT
ARGS="--workload $WL --reads $N --steps 8 --warmup 3 --extras 0 --cpu-sample 0 --check 256"
i=0
for cfg in "$A" "$B" "$A" "$B"; do
	i=$((i+1))
	env $cfg timeout -k 10 200 python3 bench.py $ARGS > $OUT/time_$i.json 2> $OUT/time_$i.err || echo "timing run $i failed"
done
j=0
for cfg in "$A" "$B"; do
	j=$((j+1))
	for c in FETCH_SIZE WRITE_SIZE; do
		env $cfg timeout -k 10 300 rocprofv3 --pmc $c --output-format csv -d $OUT/pmc_${j}_$c -o td -- python3 $ROOT/bench.py $ARGS > $OUT/pmc_${j}_$c.log 2>&1 || echo "pmc pass $j $c failed"
	done
done
python3 - "$OUT" "$A" "$B" "$N" "$WL" <<'PY'
import csv, glob, json, sys
out, A, B, n, wl = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4]), sys.argv[5]
res = {"workload": wl, "reads_per_launch": n, "method": "kernel_ms: HIP events around the decode kernel in unprofiled bench runs (two per setting, "
       "interleaved A B A B); bytes: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, KiB -> bytes, FETCH_SIZE x2 (gfx950)", "settings": {}}
for j, cfg in ((1, A), (2, B)):
    ms = []
    for i in (j, j + 2):
        try:
            ms.append(json.load(open("%s/time_%d.json" % (out, i)))["roofline"]["kernel_ms"])
        except Exception:
            pass
    p = {}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        v = []
        for f in glob.glob("%s/pmc_%d_%s/**/*counter_collection.csv" % (out, j, c), recursive=True):
            for r in csv.DictReader(open(f)):
                if "td_spec_kernel" in r["Kernel_Name"] and r["Counter_Name"] == c:
                    v.append(float(r["Counter_Value"]))
        p[c] = sum(v[-8:]) / max(len(v[-8:]), 1) if v else None
    fetch = p["FETCH_SIZE"] * 2048 if p["FETCH_SIZE"] else None
    write = p["WRITE_SIZE"] * 1024 if p["WRITE_SIZE"] else None
    res["settings"][cfg] = {"kernel_ms_runs": ms, "fetch_bytes_per_read": fetch / n if fetch else None, "write_bytes_per_read": write / n if write else None,
                            "hbm_bytes_per_read": (fetch + write) / n if fetch and write else None}
print(json.dumps(res, indent=1))
json.dump(res, open(out + "/ab.json", "w"), indent=1)
PY
