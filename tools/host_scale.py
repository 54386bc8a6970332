#!/usr/bin/env python3
"""What do N ranks' HOST halves cost one host?  Runs the host side of td_submit / td_wait -- the staging copy into page-locked
memory, the rebuilding of rewritten sequences from keep bits, the expansion of labels from runs, the copy of the records;
td_host_halves_bench in the library: the routines td_api.hip calls around its device calls, no device involved -- back to back in
1, 2, 4 and 8 processes side by side, each with its share of the CPUs this job may use (as bench.py's bind_rank_to_numa gives
every rank its share), for both kinds of caller buffers.  Prints and writes (second argument) a JSON record: aggregate reads/s
and GB/s of host memory traffic per process count -- the ceiling the host sets for the 8-GPU run, measured without eight GPUs.
usage: tools/host_scale.py [reads per batch] [out.json]"""
import ctypes as C
import json
import os
import subprocess
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def worker(n, L, threads, iters, mode, cpus):
    if cpus:
        try:
            os.sched_setaffinity(0, cpus)
        except OSError:
            pass
    from tagdust_amd import lib as tdlib
    lib = tdlib.load_library()
    lib.td_host_halves_bench.argtypes = [C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_double)]
    out = (C.c_double * 3)()
    rc = lib.td_host_halves_bench(n, L, threads, iters, mode, out)
    assert rc == 0
    print(json.dumps({"s_per_batch": out[0], "bytes_per_batch": out[1], "iters": out[2]}))


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--worker":
        n, L, threads, iters, mode = (int(x) for x in sys.argv[2:7])
        cpus = [int(x) for x in sys.argv[7].split(",")] if len(sys.argv) > 7 and sys.argv[7] else []
        worker(n, L, threads, iters, mode, cpus)
        return
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
    out_path = sys.argv[2] if len(sys.argv) > 2 else None
    L = 150
    avail = sorted(os.sched_getaffinity(0))
    try:     # a cgroup CPU quota caps what the affinity mask promises
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        quota = None if q == "max" else float(q) / float(p)
    except Exception:
        quota = None
    rec = {"what": "host halves of td_submit / td_wait (td_host_halves_bench), no device; config-3 batch shape",
           "reads_per_batch": n, "read_len": L, "cpus_in_affinity_mask": len(avail), "cgroup_cpu_quota": quota,
           "note": "threads per process = min(8, CPUs the job may use / processes): with a cgroup quota the N processes SHARE that quota -- the figures then say what that many CPUs sustain, not what an N-GPU node with N such allotments does",
           "host": os.uname().nodename, "collected": time.strftime("%Y-%m-%dT%H:%MZ", time.gmtime()), "runs": []}
    for mode, mname in ((0, "pageable caller buffers (staging copy in, records copied out)"), (1, "page-locked caller buffers, stable_input (no staging copies)")):
        for procs in (1, 2, 4, 8):
            share = max(1, len(avail) // procs)
            # threads: what bench.py gives a rank's copy pool (its share of the CPUs, at most 8) -- of the CPUs this job may really
            # use: a cgroup quota below the affinity mask's width (the pool's 1-GPU boxes: 16 of 256) is what N ranks share here
            budget = int(quota) if quota else len(avail)
            threads = max(1, min(8, share, budget // procs))
            iters = 12
            ps = []
            t0 = time.time()
            for r in range(procs):
                cpus = avail[r * share:(r + 1) * share] if share * procs <= len(avail) else avail
                ps.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), "--worker", str(n), str(L), str(threads), str(iters), str(mode),
                                            ",".join(map(str, cpus))], stdout=subprocess.PIPE))
            outs = [json.loads(p.communicate()[0].decode().strip().splitlines()[-1]) for p in ps]
            wall = time.time() - t0
            per = [o["s_per_batch"] for o in outs]
            agg_reads = sum(n / s for s in per)
            agg_gb = sum(o["bytes_per_batch"] / o["s_per_batch"] for o in outs) / 1e9
            run = {"buffers": mname, "processes": procs, "threads_per_process": threads, "cpus_per_process": share,
                   "ms_per_batch_per_process": [round(1e3 * s, 2) for s in per], "aggregate_reads_per_s": agg_reads,
                   "aggregate_host_GB_per_s": agg_gb, "per_process_reads_per_s": agg_reads / procs, "wall_s": round(wall, 1)}
            rec["runs"].append(run)
            print("%-14s %d proc x %d thr: %6.2f ms/batch/proc  aggregate %7.1f M reads/s  %6.1f GB/s" % (
                "pageable" if mode == 0 else "page-locked", procs, threads, 1e3 * sum(per) / len(per), agg_reads / 1e6, agg_gb), flush=True)
    if out_path:
        json.dump(rec, open(out_path, "w"), indent=1)


if __name__ == "__main__":
    main()
