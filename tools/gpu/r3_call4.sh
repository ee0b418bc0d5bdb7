cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3c4
V="0.0:1.0 0.8:1.0 0.8:1.0:warm_age=4 0.8:1.0:warm_age=16 0.8:1.0:warm_age=64 0.9:1.0:warm_age=16 0.9:1.0:warm_age=64"
timeout 500 python tools/drop_bricks.py 1024 250 $V 2>&1 | grep -v Warning | cut -c1-330 | tee gpurun_out/r3c4/drop_age.txt
timeout 300 python tools/stack_lab.py 240 $V 2>&1 | grep -v Warning | tee gpurun_out/r3c4/stack_age.txt
