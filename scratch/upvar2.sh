for sk in 0 1024 2048 4096 8192 15360; do
SMCP_SKIP=$sk bash scratch/trace_up.sh > /dev/null 2>&1
echo "skip $sk"; python scratch/uptimes.py k_hess_up_n16 | sort -k2 -n | tail -1
done
