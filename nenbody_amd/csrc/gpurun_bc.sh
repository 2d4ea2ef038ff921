for d in 0 1 2; do echo "debug=$d"; NB_BC_DEBUG=$d NB_STRICT_BC=1 python -u tools/shard_run.py 16384 5 2>&1 | tail -2; done
