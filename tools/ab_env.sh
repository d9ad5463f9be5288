# A/B of one environment switch on the bench: usage  VAR=TN_PIVOT_DEVICE VALUES="0 1" bash tools/ab_env.sh
# per value: pass clocks of a single chain (TN_CHAIN_PASSES=1), then the 4-chain bench (8 steps)
set -e
for v in $VALUES; do
env $VAR=$v TN_CHAIN_PASSES=1 timeout -k 10 280 python bench.py --concurrent 1 --steps 1 --warmup 1 --cpu-rows 0 --no-search --no-profile > gpurun_out/ab_pass_$v.log 2> gpurun_out/ab_pass_$v.err
grep -A9 "tn_compress_mps passes" gpurun_out/ab_pass_$v.err
env $VAR=$v timeout -k 10 280 python bench.py --steps 8 --warmup 2 --cpu-rows 0 --no-search > gpurun_out/ab_$v.log 2> gpurun_out/ab_$v.err
python - <<PY
import json
d=json.loads([x for x in open("gpurun_out/ab_$v.log") if x.startswith("{")][-1])
print("$VAR=$v value %.1f median %.1f single %.1f launches %d disc %.6e ovl %.3e" % (d["value"], d["config"]["median_ms_per_sweep"], d["config"]["single_chain_sweep_latency_ms"], d["launches_per_sweep"], d["config"]["rhoT_discarded_max"], 1-d["config"]["rhoT_overlap_min"]), d["config"]["bond_dims_mid_row"])
PY
done
