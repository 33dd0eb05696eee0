cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_baseline_configs.py tests/test_gpu_dense_api.py tests/test_gpu_fusion_parity.py -m gpu -q -x -k "merge_pointclouds or dense or centroid or sparse" --durations=3 2>&1 | tail -12
