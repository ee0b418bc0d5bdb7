# training curves of BlockAssemblyGraspSim on the compound-shape engine: a short probe of three schedules first
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5e
mkdir -p $O
timeout 200 python tools/grasp_train_r5.py 1024 1500 100 2048 adaptive $O/probe_mb2048_adaptive.txt 150 > $O/p1.log 2>&1; tail -8 $O/probe_mb2048_adaptive.txt | cut -c1-220
timeout 200 python tools/grasp_train_r5.py 1024 1500 100 8192 adaptive $O/probe_mb8192_adaptive.txt 150 > $O/p2.log 2>&1; tail -8 $O/probe_mb8192_adaptive.txt | cut -c1-220
timeout 200 python tools/grasp_train_r5.py 1024 1500 100 2048 1e-4 $O/probe_mb2048_lr1e-4.txt 150 > $O/p3.log 2>&1; tail -8 $O/probe_mb2048_lr1e-4.txt | cut -c1-220
tail -3 $O/p1.log
