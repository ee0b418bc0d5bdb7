cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(time timeout -k 5 900 python -m pytest tests -m gpu -q) > gpurun_out/gputests.log 2>&1
grep -E "passed|failed|Error " gpurun_out/gputests.log | tail -4
timeout -k 5 400 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; head -c 200 gpurun_out/bench_default.json; echo
timeout -k 5 200 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-large-minibatch > gpurun_out/bench_protocol.json 2>/dev/null; head -c 200 gpurun_out/bench_protocol.json; echo
SDX_FORCE_MULTI_RANK=1 timeout -k 5 200 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-large-minibatch 2>/dev/null | head -1 > gpurun_out/bench_fmr.json; head -c 200 gpurun_out/bench_fmr.json; echo
timeout -k 5 200 python bench.py --num-envs 4096 --minibatch 32768 --mixed-precision --steps 5 --warmup 2 --no-cpu-baseline --no-large-minibatch > gpurun_out/bench_n4096_bf16.json 2>/dev/null; head -c 200 gpurun_out/bench_n4096_bf16.json; echo
timeout -k 5 200 python bench.py --num-envs 4096 --minibatch 32768 --steps 5 --warmup 2 --no-cpu-baseline --no-large-minibatch > gpurun_out/bench_n4096_fp32.json 2>/dev/null; head -c 200 gpurun_out/bench_n4096_fp32.json; echo
timeout -k 5 300 python bench.py --pretrain-epochs 200 --steps 5 --warmup 2 --no-cpu-baseline --no-large-minibatch > gpurun_out/bench_trained200.json 2>/dev/null; head -c 200 gpurun_out/bench_trained200.json; echo
timeout -k 5 300 python tools/bench_config3.py 1024 12 > gpurun_out/config3.json 2>/dev/null; tail -c 300 gpurun_out/config3.json
