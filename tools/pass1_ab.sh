# A/B of the weighted rank-revealing first canonisation pass on the headline workload (single chain, one sweep each)
for rv in 0 1; do
TN_PASS1_WEIGHTED=$rv TN_PASS1_TRACE=1 python bench.py --L 2048 --no-search --cpu-rows 0 --steps 1 --warmup 0 --concurrent 1 --no-profile > gpurun_out/r2_pass1_$rv.json 2> gpurun_out/r2_pass1_$rv.err
echo weighted=$rv ms: $(python -c "import json;d=json.load(open('gpurun_out/r2_pass1_$rv.json'));print(d['value'], d['config']['rhoT_discarded_max'], d['config']['rhoT_overlap_min'], d['config']['bond_dims_mid_row'])")
done
grep pass1 gpurun_out/r2_pass1_1.err | head -16
echo fallbacks: $(grep -c "accept=False" gpurun_out/r2_pass1_1.err)
