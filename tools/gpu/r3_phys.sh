cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3p
timeout 300 python -m pytest tests/test_gpu_physics_parity.py tests/test_gpu_fullsize_properties.py tests/test_gpu_fullsize_tasks.py -q -m gpu -x 2>&1 | tail -3
timeout 100 python tools/time_physics.py 1024 8 | tee gpurun_out/r3p/tp0.json | cut -c1-500
for n in 512 2048 4096; do timeout 100 python tools/time_physics.py $n 8 | cut -c150-420; done
