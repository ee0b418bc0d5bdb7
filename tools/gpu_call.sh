cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(time timeout -k 5 600 python -m pytest tests/test_gpu_fullsize_tasks.py -m gpu -q -k deterministic) > gpurun_out/gputests.log 2>&1
grep -E "passed|failed|Error|error|assert|Mismatch|Max |step" gpurun_out/gputests.log | tail -25
