# A/B of the Jacobi convergence measure on the headline workload (one single-chain sweep each)
for rel in 0 1; do
TN_SVD_RELEVANT=$rel TN_SVD_TRACE=1 python bench.py --L 2048 --no-search --cpu-rows 0 --steps 1 --warmup 0 --concurrent 1 --no-profile > gpurun_out/r2_svdrel_$rel.json 2> gpurun_out/r2_svdrel_$rel.err
echo relevant=$rel sweeps: $(grep -c tn_svd gpurun_out/r2_svdrel_$rel.err) $(python -c "import json;d=json.load(open('gpurun_out/r2_svdrel_$rel.json'));print(d['value'], d['config']['rhoT_discarded_max'], d['config']['rhoT_overlap_min'], d['config']['bond_dims_mid_row'])")
done
