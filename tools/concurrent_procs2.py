#!/usr/bin/env python3
"""Experiment: P bench processes on one GPU at the same time, C interleaved chains each (P x C = rotations in flight).  Do processes
interleave better than the threads / streams of one process?  Usage: concurrent_procs2.py P C [steps]"""
import json
import os
import subprocess
import sys
import time

P = int(sys.argv[1]) if len(sys.argv) > 1 else 2
Cc = int(sys.argv[2]) if len(sys.argv) > 2 else 2
steps = sys.argv[3] if len(sys.argv) > 3 else '10'
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
cmd = [sys.executable, os.path.join(root, 'bench.py'), '--concurrent', str(Cc), '--no-profile', '--cpu-rows', '0', '--no-search', '--steps', steps, '--warmup', '2']
t0 = time.time()
procs = [subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True) for _ in range(P)]
outs = [p.communicate()[0] for p in procs]
wall = time.time() - t0
res = [json.loads(o.strip().splitlines()[-1]) for o in outs]
print('procs', P, 'chains each', Cc, 'step ms per process', [round(r['ms_per_step']) for r in res], 'median steps', [sorted(r['config']['step_ms_rank0'])[len(r['config']['step_ms_rank0']) // 2] for r in res], 'wall', round(wall, 1))
