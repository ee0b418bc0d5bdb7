cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(time timeout 2000 python -m pytest tests -m gpu -q) > gpurun_out/gputests.log 2>&1
tail -40 gpurun_out/gputests.log
