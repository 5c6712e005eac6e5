for lb in 20 26 32 40; do for im in 6 12 20; do echo "== leaf_batch $lb inner_min $im"; PRT_TUNE_LEAF_BATCH=$lb PRT_TUNE_INNER_MIN=$im timeout -k 10 100 python tools/exp_ab.py || exit 1; done; done
