cd "$(dirname "$0")"
for d in 0 1 2 4 3 5 6 7; do TN_EIG_DBG=$d timeout -k 10 120 python eig_diag.py 2>&1 | grep TN_EIG_DBG; done
