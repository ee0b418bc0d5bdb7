cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3c3
timeout 500 python tools/drop_bricks.py 1024 250 0.0:1.0:contact_offset=0.004 0.8:1.0:contact_offset=0.004 0.0:1.0:contact_offset=0.008 0.8:1.0:contact_offset=0.008 2>&1 | grep -v Warning | cut -c1-700 | tee gpurun_out/r3c3/drop_offset.txt
