cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5c
mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_physics_parity.py -q -m gpu 2>&1 | tail -25 > $O/tests1.txt; tail -4 $O/tests1.txt
timeout 600 python tools/lift_diag.py 1024 --dump 3 --variants "0.9,14,0.05;0.9,14,0.0;0.8,14,0.05;0.7,14,0.05;0.9,8,0.05;0.9,20,0.03;0.6,14,0.05" > $O/lift_diag.txt 2>$O/lift_diag.err; grep -v "^trace" $O/lift_diag.txt | cut -c1-700 | head -30; tail -3 $O/lift_diag.err
