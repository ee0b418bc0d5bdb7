cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3c2
timeout 500 python tools/drop_bricks.py 1024 250 0.0:1.0 0.0:1.4 0.0:1.8 0.8:1.0 0.8:1.4 0.8:1.8 0.5:1.4 2>&1 | grep -v Warning | cut -c1-700 | tee gpurun_out/r3c2/drop_grid.txt
