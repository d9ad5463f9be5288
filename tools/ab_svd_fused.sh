set -e
for mode in 1 0 1 0; do
TN_SVD_FUSED=$mode timeout -k 10 300 python bench.py --steps 8 --warmup 2 > gpurun_out/ab_$mode.log 2>&1
python - <<PY
import json
d=json.loads([x for x in open("gpurun_out/ab_$mode.log") if x.startswith("{")][-1])
print("TN_SVD_FUSED=$mode value %.1f median %.1f single %.1f svd %.1f" % (d["value"], d["config"]["median_ms_per_sweep"], d["config"]["single_chain_sweep_latency_ms"], d["phase_kernel_time_single_chain_sweep"]["svd_trunc"]["kernel_ms"]))
PY
done
