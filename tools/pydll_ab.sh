for v in 0 1; do echo "== TN_PYDLL=$v"; TN_PYDLL=$v python tools/host_overhead.py 2>&1 | tail -11; done
