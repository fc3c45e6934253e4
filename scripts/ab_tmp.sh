for v in 1 0 1 0; do echo "PQ_BN_ATOMIC=$v"; PQ_BN_ATOMIC=$v timeout -k 10 200 python bench.py --no-cpu-baseline 2>/dev/null | cut -c100-200; done
