# A/B of the intermediate-pass decompositions that cannot truncate (chain.hip: gauge_svd_skippable): pass clocks of a single chain,
# then the 4-chain bench; TN_GAUGE_SVD=1: every decomposition (the round-4 behaviour), 2: those of the 2 chi pass only, 0: none (default)
set -e
for mode in ${MODES:-1 2 0}; do
TN_GAUGE_SVD=$mode TN_CHAIN_PASSES=1 timeout -k 10 280 python bench.py --concurrent 1 --steps 1 --warmup 1 --cpu-rows 0 --no-search --no-profile > gpurun_out/gs_pass_$mode.log 2> gpurun_out/gs_pass_$mode.err
grep -A9 "tn_compress_mps passes" gpurun_out/gs_pass_$mode.err
TN_GAUGE_SVD=$mode timeout -k 10 280 python bench.py --steps 8 --warmup 2 --cpu-rows 0 --no-search > gpurun_out/gs_$mode.log 2> gpurun_out/gs_$mode.err
python - <<PY
import json
d=json.loads([x for x in open("gpurun_out/gs_$mode.log") if x.startswith("{")][-1])
print("TN_GAUGE_SVD=$mode value %.1f median %.1f single %.1f svd %.1f disc %.6e ovl %.3e" % (d["value"], d["config"]["median_ms_per_sweep"], d["config"]["single_chain_sweep_latency_ms"], d["phase_kernel_time_single_chain_sweep"]["svd_trunc"]["kernel_ms"], d["config"]["rhoT_discarded_max"], 1-d["config"]["rhoT_overlap_min"]), d["config"]["bond_dims_mid_row"])
PY
done
