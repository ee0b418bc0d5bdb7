# round 5, first GPU pass after the compound shapes: physics parity + known-answer tests on the device, k_physics timing (production, rolled
# variants, phase clock), scripted-grasp lift diagnostic
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5a
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_physics_parity.py tests/test_gpu_seams.py tests/test_gpu_task_parity.py -q -m gpu -x 2>&1 | tail -25 > $O/tests1.txt; tail -8 $O/tests1.txt
for v in hip u1 u2; do SDX_LIB_PATH=$PWD/seqdex_amd/lib/libseqdex_$v.so timeout 120 python tools/time_physics.py 1024 24 > $O/time_$v.json 2>$O/time_$v.err; python -c "
import json; d=json.load(open('$O/time_$v.json')); print('$v', round(d['k_physics_ms'],4), 'contacts', d['contacts_mean'], d['contacts_max'])"; done
for e in 0 19; do SDX_LIB_PATH=$PWD/seqdex_amd/lib/libseqdex_prof.so SDX_DEBUG_ENV=$e timeout 120 python tools/time_physics.py 1024 24 > $O/phase_env$e.json 2>/dev/null; python -c "
import json; d=json.load(open('$O/phase_env$e.json')); print(d['k_physics_ms'], d['debug_env_contacts'], d['debug_env_has_robot_contact'], d['phase_cycles_env0_substep0'])"; done
timeout 300 python tools/lift_diag.py 1024 --dump 2 > $O/lift_diag.txt 2>$O/lift_diag.err; grep -v "^trace" $O/lift_diag.txt | head -30; tail -3 $O/lift_diag.err
timeout 600 python -m pytest tests/test_gpu_fullsize_properties.py -q -m gpu -x -k "not persistent_update" 2>&1 | tail -25 > $O/tests2.txt; tail -8 $O/tests2.txt
