#!/bin/bash
# Copy what tools/collect_profiles.sh, tools/collect_side_benches.sh and the driver-form bench run left under gpurun_out/ into profiles/
# under the round's names (ROUND=r05 by default) and rebuild profiles/${ROUND}_pmc_traffic.json.  Run here, after the gpurun call.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
ROUND=${ROUND:-r05}
P=$ROOT/gpurun_out/prof_$ROUND
D=$ROOT/profiles
cp $P/kt/bench_kernel_stats.csv $D/${ROUND}_rocprofv3_kernel_stats_bench_L2048_chi64_4chains.csv
cp $P/kt/bench_domain_stats.csv $D/${ROUND}_rocprofv3_domain_stats_bench_L2048_chi64_4chains.csv
cp $P/kt1/bench_kernel_stats.csv $D/${ROUND}_rocprofv3_kernel_stats_bench_L2048_chi64_single_chain.csv
cp $P/kt1/bench_domain_stats.csv $D/${ROUND}_rocprofv3_domain_stats_bench_L2048_chi64_single_chain.csv
cp $P/bench_under_rocprof.json $D/${ROUND}_bench_under_rocprofv3_4chains.json
cp $P/bench_c1_under_rocprof.json $D/${ROUND}_bench_under_rocprofv3_single_chain.json
cp $P/pmc_FETCH_SIZE_summary.csv $D/${ROUND}_pmc_FETCH_SIZE_probe.csv
cp $P/pmc_WRITE_SIZE_summary.csv $D/${ROUND}_pmc_WRITE_SIZE_probe.csv
cp "$P/pmc_SQ_INSTS_VALU_MFMA_F64+SQ_VALU_MFMA_BUSY_CYCLES+SQ_BUSY_CYCLES_summary.csv" $D/${ROUND}_pmc_SQ_INSTS_VALU_MFMA_F64_probe.csv
cp "$P/pmc_SQ_INSTS_VALU_MFMA_MOPS_F64+SQ_WAVE_CYCLES+SQ_WAIT_INST_LDS_summary.csv" $D/${ROUND}_pmc_SQ_INSTS_VALU_MFMA_MOPS_F64_probe.csv
for f in chimera512 rmf64 default; do [ -f $ROOT/gpurun_out/bench_$f.json ] && cp $ROOT/gpurun_out/bench_$f.json $D/${ROUND}_bench_$f.json; done
[ -f $ROOT/gpurun_out/bench_force_dist.json ] && cp $ROOT/gpurun_out/bench_force_dist.json $D/${ROUND}_bench_force_dist_1rank.json
[ -f $ROOT/gpurun_out/bench_driver_form.json ] && cp $ROOT/gpurun_out/bench_driver_form.json $D/${ROUND}_bench_driver_form_steps20_warmup5.json
[ -f $ROOT/gpurun_out/chain_scaling.txt ] && cp $ROOT/gpurun_out/chain_scaling.txt $D/${ROUND}_chain_scaling.txt
ROUND=$ROUND python3 $ROOT/tools/make_pmc_json.py "$(sed 's/PROBE_TIMES //' $P/probe_times.txt)" > /dev/null
echo "published $ROUND"
