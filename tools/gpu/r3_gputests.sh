cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3t
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -25 | tee gpurun_out/r3t/gputests.txt
