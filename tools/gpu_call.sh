cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(time timeout 900 python -m pytest tests/test_gpu_ppo_parity.py -m gpu -q -s -k "bf16_gradients") > gpurun_out/gputests.log 2>&1
grep -E "passed|failed|bf16 |Error|error|ACTUAL|DESIRED|Max" gpurun_out/gputests.log | tail -16
