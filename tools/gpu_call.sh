cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(time timeout -k 5 600 python -m pytest tests/test_gpu_fullsize_properties.py -m gpu -q -x) > gpurun_out/gputests_piles.log 2>&1
grep -E "passed|failed|Error " gpurun_out/gputests_piles.log | tail -4
timeout -k 5 200 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-large-minibatch > gpurun_out/bench_quick.json 2>/dev/null; head -c 200 gpurun_out/bench_quick.json; echo
