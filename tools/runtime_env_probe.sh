#!/bin/bash
# HIP runtime knobs against the step time of the bench sweep (4 interleaved chains unless CHAINS is set).  None of them helped
# (round 3): HIP_FORCE_DEV_KERNARG=1 and device-scope fences are already the defaults.
C=${CHAINS:-4}
run() { "$@" timeout -k 10 200 python bench.py --concurrent $C --steps 3 --warmup 1 --no-search --cpu-rows 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"; }
echo "base: $(run env)"
for kv in AMD_OPT_FLUSH=0 ROC_SYSTEM_SCOPE_SIGNAL=0 ROC_USE_FGS_KERNARG=0 ROC_USE_FGS_KERNARG=1 DEBUG_HIP_KERNARG_COPY_OPT=0 GPU_STREAMOPS_CP_WAIT=1 AMD_DIRECT_DISPATCH=0 HIP_FORCE_DEV_KERNARG=0 ROC_ACTIVE_WAIT_TIMEOUT=1000 ROC_AQL_QUEUE_SIZE=16384 DEBUG_HIP_DYNAMIC_QUEUES=0; do
  echo "$kv: $(run env $kv)"
done
echo "base again: $(run env)"
