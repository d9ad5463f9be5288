#!/bin/bash
# Step time of the bench sweep against the number of chains interleaved on one GPU (1 .. 4 rotations of the same instance).
for n in 1 2 3 4; do
  timeout -k 10 300 python bench.py --concurrent $n --steps 4 --warmup 2 --no-search --cpu-rows 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('chains $n: ms_per_step %.1f' % d['ms_per_step'])"
done
