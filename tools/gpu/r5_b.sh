# round 5, second GPU pass: greedy manifold + scripted controller with grip detection
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5b
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_physics_parity.py -q -m gpu 2>&1 | tail -25 > $O/tests1.txt; tail -6 $O/tests1.txt
timeout 120 python tools/time_physics.py 1024 24 > $O/time_hip.json 2>$O/time_hip.err; python -c "
import json; d=json.load(open('$O/time_hip.json')); print('hip', round(d['k_physics_ms'],4), 'contacts', d['contacts_mean'], d['contacts_max'])"
timeout 300 python tools/lift_diag.py 1024 --dump 4 > $O/lift_diag.txt 2>$O/lift_diag.err; grep -v "^trace" $O/lift_diag.txt | cut -c1-900 | head -30; tail -3 $O/lift_diag.err
timeout 900 python -m pytest tests/test_gpu_fullsize_properties.py -q -m gpu -k "not persistent_update" 2>&1 | tail -25 > $O/tests2.txt; tail -8 $O/tests2.txt
