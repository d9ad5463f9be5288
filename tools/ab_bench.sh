# A/B runs of the 4-chain bench under an environment switch:  ab_bench.sh VAR v1 v2 ...
VAR=$1; shift
for v in "$@"; do
  env $VAR=$v python bench.py --steps 3 --warmup 1 --no-search --cpu-rows 0 --no-profile > gpurun_out/ab_${VAR}_$v.json 2> gpurun_out/ab_${VAR}_$v.err
  echo "$VAR=$v: $(python -c "import json;d=json.load(open('gpurun_out/ab_${VAR}_$v.json'));print('value %.1f ms/sweep  steps %s' % (d['value'], d['config']['step_ms_rank0']))")"
done
