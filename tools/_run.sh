cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests -m gpu -q -x --durations=5 > gpurun_out/r04_t12.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r04_t12.log
tail -14 gpurun_out/r04_t12.log
