cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3l
timeout 200 python -m pytest -q -m gpu tests/test_gpu_linear_kernel.py -x 2>&1 | tail -4 | tee gpurun_out/r3l/tests.log
timeout 120 python tools/time_linear.py 1024 3 7 1 8 2>&1 | grep -v amdgpu | tee gpurun_out/r3l/time_linear_dual.txt
