#!/usr/bin/env python3
"""Experiment: N independent single-chain bench processes on one GPU at the same time (host-threading check for
run_concurrent: if processes interleave much better than threads, the host side is the limit)."""
import json
import subprocess
import sys
import time
import os

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
cmd = [sys.executable, os.path.join(root, 'bench.py'), '--concurrent', '1', '--no-profile', '--cpu-rows', '0', '--no-search',
       '--steps', '3', '--warmup', '1']
t0 = time.time()
procs = [subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True) for _ in range(n)]
outs = [p.communicate()[0] for p in procs]
wall = time.time() - t0
vals = [json.loads(o.strip().splitlines()[-1])['value'] for o in outs]
print('procs', n, 'per-process ms/sweep', [round(v) for v in vals], 'aggregate ms/sweep', round(sum(vals) / n / n), 'wall', round(wall, 1))
