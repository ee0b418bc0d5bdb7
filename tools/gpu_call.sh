cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout -k 5 300 python -m pytest tests/test_gpu_physics_parity.py tests/test_gpu_fullsize_properties.py -m gpu -q -x -k "physics or kinematics or free_fall or teacher") 2>&1 | grep -E "passed|failed|Error" | tail -3
SDX_PHYS_NT=512 timeout -k 5 120 python tools/time_physics.py 1024 24 > gpurun_out/phys_nt512.json 2>/dev/null; cat gpurun_out/phys_nt512.json
SDX_TP_ITERS=1 timeout -k 5 120 python tools/time_physics.py 1024 24 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print(d['solver_iters'], round(d['k_physics_ms'],4), d['contacts_mean'])"
