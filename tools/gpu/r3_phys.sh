cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_physics_parity.py -q -m gpu 2>&1 | grep -E "passed|failed"
timeout 100 python tools/time_physics.py 1024 8 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('production', d['k_physics_ms'])"
for e in 19; do SDX_LIB_PATH=$PWD/seqdex_amd/lib/libseqdex_prof.so SDX_DEBUG_ENV=$e timeout 100 python tools/time_physics.py 1024 8 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); p=d['phase_cycles_env0_substep0']; print(d['k_physics_ms'], p)"; done
