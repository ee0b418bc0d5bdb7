cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(time timeout -k 5 600 python -m pytest tests/test_gpu_fullsize_properties.py tests/test_gpu_ppo_parity.py -m gpu -q -x) > gpurun_out/gputests_ppo.log 2>&1
grep -E "passed|failed|Error " gpurun_out/gputests_ppo.log | tail -4
timeout -k 5 120 python tools/time_multi_rank_parts.py > gpurun_out/mr_parts.txt 2>&1; tail -12 gpurun_out/mr_parts.txt
SDX_FORCE_MULTI_RANK=1 timeout -k 5 200 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-large-minibatch 2>/dev/null | head -1 > gpurun_out/bench_fmr.json; head -c 230 gpurun_out/bench_fmr.json; echo
SDX_FORCE_MULTI_RANK=1 SDX_MULTI_RANK_GRAPH=0 timeout -k 5 200 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-large-minibatch 2>/dev/null | head -1 > gpurun_out/bench_fmr_eager.json; head -c 230 gpurun_out/bench_fmr_eager.json; echo
