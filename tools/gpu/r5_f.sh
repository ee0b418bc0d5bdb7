# the 19 000-epoch BlockAssemblyGraspSim run (large-minibatch path, minibatch 2048, shipped adaptive LR) and the shipped
# minibatch-4 schedule for as long as the time box allows
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5f
mkdir -p $O
timeout 700 python tools/grasp_train_r5.py 1024 19000 250 2048 adaptive $O/r5_grasp_train_curve_mb2048_19000_epochs.txt 640 > $O/t1.log 2>&1; tail -4 $O/r5_grasp_train_curve_mb2048_19000_epochs.txt | cut -c1-250
timeout 560 python tools/grasp_train_r5.py 1024 3000 50 4 adaptive $O/r5_grasp_train_curve_shipped_minibatch4.txt 500 > $O/t2.log 2>&1; tail -4 $O/r5_grasp_train_curve_shipped_minibatch4.txt | cut -c1-250
tail -2 $O/t1.log $O/t2.log | cut -c1-300
