run() { "$@" timeout -k 10 200 python bench.py --concurrent 1 --steps 3 --warmup 1 --no-search --cpu-rows 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"; }
echo "base: $(run env)"
echo "HIP_FORCE_DEV_KERNARG=1: $(run env HIP_FORCE_DEV_KERNARG=1)"
echo "HIP_FORCE_DEV_KERNARG=0: $(run env HIP_FORCE_DEV_KERNARG=0)"
echo "ROC_ACTIVE_WAIT_TIMEOUT=100: $(run env ROC_ACTIVE_WAIT_TIMEOUT=100)"
echo "ROC_ACTIVE_WAIT_TIMEOUT=1000: $(run env ROC_ACTIVE_WAIT_TIMEOUT=1000)"
echo "both: $(run env HIP_FORCE_DEV_KERNARG=1 ROC_ACTIVE_WAIT_TIMEOUT=1000)"
echo "base again: $(run env)"
