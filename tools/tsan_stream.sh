#!/bin/bash
# The streaming pipeline (csrc/td_stream.cpp: reader/parser thread, writer thread, two thread pools, the caller's thread) under
# ThreadSanitizer on the CPU: td_stream.cpp + td_fastq.cpp as they are, the device entry points replaced by stand-ins
# (tools/tsan_stream/stub.cpp), a 60 000-read ragged FASTQ file in batches of 777 reads and blocks of 20 KB -- parse-only and
# with the stubbed decode + real formatting / appends, each twice (the second run reuses the cached batch buffers), and the
# multi-file pipeline (td_stream_run_multi: two input files in lock-step, two "devices", one file not decoded).
# usage: tools/tsan_stream.sh      (prints the result lines, the last one of a run into a full disk; any ThreadSanitizer report goes to stderr)
set -e
cd "$(dirname "$0")/.."
OUT=/tmp/td_tsan
mkdir -p $OUT
g++ -std=c++17 -O1 -g -fsanitize=thread -pthread -Iinclude -w -o $OUT/tsan_stream tools/tsan_stream/stub.cpp tagdust_amd/csrc/td_stream.cpp tagdust_amd/csrc/td_fastq.cpp
python3 - <<'PY'
import numpy as np
rng = np.random.RandomState(1)
with open('/tmp/td_tsan/in.fq', 'wb') as f:
    for i in range(60000):
        L = int(rng.randint(20, 120)); s = ''.join(rng.choice(list('ACGTN'), L)); f.write(('@r%d x\n%s\n+\n%s\n' % (i, s, 'I' * L)).encode())
PY
cd $OUT
for mode in "" decode multi; do ./tsan_stream in.fq 777 4 20000 $mode; done
# a full disk: every output file a link to /dev/full -- the appender thread's pwrite fails, the run must end with TD_FAIL and
# "write failed" (not hang, not report success)
for f in out_*.fq; do rm -f $f; ln -s /dev/full $f; done
timeout 120 ./tsan_stream in.fq 777 4 20000 decode | head -1
rm -f out_*.fq
