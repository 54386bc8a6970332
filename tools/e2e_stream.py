#!/usr/bin/env python3
"""Streaming end to end (td_stream_run): a FASTQ file of COPIES x 2^20 bench reads -> demultiplexed files; stage rates as JSON.
usage: tools/e2e_stream.py [copies] [n_threads]     (default 16 copies = 5.3 GB of FASTQ)"""
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench

copies = int(sys.argv[1]) if len(sys.argv) > 1 else 16
nt = int(sys.argv[2]) if len(sys.argv) > 2 else 0
print(json.dumps(bench.e2e_stream(0, copies=copies, n_threads=nt), indent=1))
