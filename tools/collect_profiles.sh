#!/bin/bash
# Profile collection (ROUND=r05 by default) on the GPU box (run through gpurun): rocprofv3 kernel statistics of the bench command and
# PMC passes (FETCH_SIZE, WRITE_SIZE, MFMA counters: separate runs, counters never combined with tracing domains) over
# tools/pmc_probe.py.  Summaries land in gpurun_out/prof_$ROUND/ and are copied into profiles/ by hand.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
ROUND=${ROUND:-r05}
OUT=$ROOT/gpurun_out/prof_$ROUND
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
echo "== counters" > $OUT/log.txt
rocprofv3 -L 2>/dev/null | grep -i -E "mfma|FETCH_SIZE|WRITE_SIZE|SQ_BUSY_CYCLES|GRBM_GUI_ACTIVE" | head -60 > $OUT/counters_available.txt
if [ -z "$PMC_ONLY" ]; then
echo "== kernel trace of the bench command (4 interleaved chains)" >> $OUT/log.txt
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o bench -- python3 $ROOT/bench.py --steps 3 --warmup 1 --no-search --cpu-rows 0 > $OUT/bench_under_rocprof.json 2> $OUT/bench_under_rocprof.err
echo "rc=$?" >> $OUT/log.txt
echo "== kernel trace of the single-chain form" >> $OUT/log.txt
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt1 -o bench -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-search --cpu-rows 0 --concurrent 1 > $OUT/bench_c1_under_rocprof.json 2> $OUT/bench_c1_under_rocprof.err
echo "rc=$?" >> $OUT/log.txt
fi
python3 $ROOT/tools/pmc_probe.py 2>/dev/null | grep PROBE_TIMES > $OUT/probe_times.txt
for ctr in FETCH_SIZE WRITE_SIZE "SQ_INSTS_VALU_MFMA_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_WAVE_CYCLES SQ_WAIT_INST_LDS"; do
  tag=$(echo $ctr | tr ' ' '+')
  echo "== pmc $ctr" >> $OUT/log.txt
  timeout -k 10 300 rocprofv3 --pmc $ctr --output-format csv -d $OUT/pmc_$tag -o probe -- python3 $ROOT/tools/pmc_probe.py > $OUT/pmc_$tag.log 2>&1
  echo "rc=$?" >> $OUT/log.txt
  f=$(ls $OUT/pmc_$tag/*/*counter_collection.csv $OUT/pmc_$tag/*counter_collection.csv 2>/dev/null | head -1)
  if [ -n "$f" ]; then python3 $ROOT/tools/pmc_summary.py $f $OUT/pmc_${tag}_summary.csv > /dev/null; rm -f $f; fi
done
# keep only the statistics (the raw traces are hundreds of MB)
find $OUT -name "*kernel_trace.csv" -delete
find $OUT -name "*.db" -delete
ls -R $OUT | head -50 >> $OUT/log.txt
cat $OUT/log.txt
