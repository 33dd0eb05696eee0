cd $GRAFT_REPO_ROOT
timeout -k 5 120 python3 tools/_dbg.py 2>&1 | tail -10
timeout -k 10 900 python -m pytest tests -m gpu -q > gpurun_out/r04_t11.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r04_t11.log
tail -8 gpurun_out/r04_t11.log
