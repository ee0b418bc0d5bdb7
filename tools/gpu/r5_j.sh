cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5j
mkdir -p $O
./tools/gpu/probe/mfma_branch | tail -6
./tools/gpu/probe/mfma_branch_nobranch | tail -3
timeout 120 python tools/time_physics.py 1024 24 > $O/time_hip.json 2>$O/time_hip.err; python -c "
import json; d=json.load(open('$O/time_hip.json')); print('hip', round(d['k_physics_ms'],4), 'contacts', d['contacts_mean'], d['contacts_max'])"
for e in 0 19; do SDX_LIB_PATH=$PWD/seqdex_amd/lib/libseqdex_prof.so SDX_DEBUG_ENV=$e timeout 120 python tools/time_physics.py 1024 24 > $O/phase_env$e.json 2>/dev/null; python -c "
import json; d=json.load(open('$O/phase_env$e.json')); print(d['k_physics_ms'], d['debug_env_contacts'], d['debug_env_has_robot_contact'], d['phase_cycles_env0_substep0'])"; done
timeout 200 python -m pytest tests/test_gpu_physics_parity.py tests/test_gpu_scripted_lift.py -q -m gpu 2>&1 | tail -3
