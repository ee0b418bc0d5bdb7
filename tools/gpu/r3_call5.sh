cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3c5
timeout 500 python tools/pile_penetration.py 1024 250 0.0:1.0 0.8:1.0:warm_age=16 2>&1 | grep -v Warning | tee gpurun_out/r3c5/pile_pen.txt
timeout 300 python -m pytest tests/test_gpu_physics_parity.py -q -m gpu -x 2>&1 | tail -5
timeout 100 python tools/time_physics.py 1024 8 | cut -c1-900
