cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 5 60 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout -k 5 60 python -m pytest tests/test_gpu_physics_parity.py -m gpu -q -x 2>&1 | tail -1
