# quick A/B of environment settings on the 4-chain bench: usage  bash tools/ab_quick.sh "TN_X=1" "TN_X=2 TN_Y=3" ...   ("" = defaults)
for st in "$@"; do
  env $st timeout -k 10 280 python bench.py --steps 10 --warmup 3 --cpu-rows 0 --no-search --no-profile 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('[%s] value %.1f median %.1f' % ('$st', d['value'], d['config']['median_ms_per_sweep']))"
done
