# A/B of environment settings on BOTH forms of the bench: four interleaved chains (the headline) and one chain alone.
# usage  bash tools/ab_both.sh "TN_X=1" "" ...   ("" = defaults)
for st in "$@"; do
  for c in 4 1; do
    env $st timeout -k 10 280 python bench.py --steps 6 --warmup 2 --cpu-rows 0 --no-search --no-profile --concurrent $c 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('[%s] chains %s  value %.1f median %.1f' % ('$st', '$c', d['value'], d['config']['median_ms_per_sweep']))"
  done
done
