cd $GRAFT_REPO_ROOT
timeout -k 5 45 python -m pytest tests/test_gpu_orient_parity.py tests/test_gpu_search_parity.py -m gpu -q -x 2>&1 | tail -1
