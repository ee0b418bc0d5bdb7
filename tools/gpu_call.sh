cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(time timeout 900 python -m pytest tests/test_gpu_physics_parity.py tests/test_gpu_fullsize_properties.py -m gpu -q -x) > gpurun_out/gputests.log 2>&1
grep -E "passed|failed" gpurun_out/gputests.log | tail -2
for nt in 512; do SDX_PHYS_NT=$nt timeout 300 python tools/time_physics.py 1024 24 > gpurun_out/phys_nt$nt.json 2> gpurun_out/phys_nt$nt.err; cat gpurun_out/phys_nt$nt.json; done
SDX_PHYS_NT=512 timeout 300 python tools/time_physics.py 256 24 2>/dev/null
