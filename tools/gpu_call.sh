cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 5 200 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-large-minibatch > gpurun_out/bench_protocol.json 2>/dev/null; head -c 300 gpurun_out/bench_protocol.json; echo
SDX_FORCE_MULTI_RANK=1 timeout -k 5 200 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-large-minibatch 2>/dev/null | head -1 > gpurun_out/bench_fmr.json; head -c 300 gpurun_out/bench_fmr.json; echo
timeout -k 5 200 python bench.py --num-envs 4096 --minibatch 32768 --mixed-precision --steps 5 --warmup 2 --no-cpu-baseline --no-large-minibatch > gpurun_out/bench_n4096_bf16.json 2>gpurun_out/bench_n4096_bf16.err; head -c 300 gpurun_out/bench_n4096_bf16.json; echo; tail -2 gpurun_out/bench_n4096_bf16.err
timeout -k 5 200 python bench.py --num-envs 4096 --minibatch 32768 --steps 5 --warmup 2 --no-cpu-baseline --no-large-minibatch > gpurun_out/bench_n4096_fp32.json 2>/dev/null; head -c 300 gpurun_out/bench_n4096_fp32.json; echo
timeout -k 5 300 python bench.py --pretrain-epochs 200 --steps 5 --warmup 2 --no-cpu-baseline --no-large-minibatch > gpurun_out/bench_trained200.json 2>/dev/null; head -c 300 gpurun_out/bench_trained200.json; echo
