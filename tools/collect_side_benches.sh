set -e
cd $GRAFT_REPO_ROOT
bash tools/chain_scaling.sh > gpurun_out/chain_scaling.txt 2>&1
cat gpurun_out/chain_scaling.txt
timeout -k 10 300 python bench.py --workload chimera512 --steps 3 --warmup 1 --no-search --cpu-rows 0 > gpurun_out/bench_chimera512.json 2> gpurun_out/bench_chimera512.err
timeout -k 10 400 python bench.py --workload rmf64 --steps 1 --warmup 0 --no-search --cpu-rows 0 > gpurun_out/bench_rmf64.json 2> gpurun_out/bench_rmf64.err
timeout -k 10 300 python bench.py --force-dist --steps 3 --warmup 1 --no-search --cpu-rows 0 > gpurun_out/bench_force_dist.json 2> gpurun_out/bench_force_dist.err
timeout -k 10 400 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err
python - <<PY
import json
for f in ('bench_chimera512','bench_rmf64','bench_force_dist','bench_default'):
    d=json.loads([x for x in open('gpurun_out/%s.json'%f) if x.startswith('{')][-1]); print(f, d['value'])
PY
