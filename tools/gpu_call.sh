cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(time timeout -k 5 900 python -m pytest tests -m gpu -q) > gpurun_out/gputests.log 2>&1
grep -E "passed|failed|Error " gpurun_out/gputests.log | tail -6
timeout -k 5 120 python tools/time_physics.py > gpurun_out/time_physics.txt 2>&1; tail -1 gpurun_out/time_physics.txt | cut -c1-200
SDX_WARM_START=0.8 timeout -k 5 120 python tools/time_physics.py > gpurun_out/time_physics_warm.txt 2>&1; tail -1 gpurun_out/time_physics_warm.txt | cut -c1-200
bash tools/prof_r2_cmd.sh > gpurun_out/prof_all.log 2>&1; tail -25 gpurun_out/prof_all.log | cut -c1-220
cd $GRAFT_REPO_ROOT
timeout -k 5 400 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; head -c 200 gpurun_out/bench_default.json; echo
timeout -k 5 200 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-large-minibatch > gpurun_out/bench_protocol.json 2>/dev/null; head -c 200 gpurun_out/bench_protocol.json; echo
SDX_FORCE_MULTI_RANK=1 timeout -k 5 200 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-large-minibatch 2>/dev/null | head -1 > gpurun_out/bench_fmr.json; head -c 200 gpurun_out/bench_fmr.json; echo
timeout -k 5 300 python bench.py --pretrain-epochs 200 --steps 5 --warmup 2 --no-cpu-baseline --no-large-minibatch > gpurun_out/bench_trained200.json 2>/dev/null; head -c 200 gpurun_out/bench_trained200.json; echo
SDX_WARM_START=0.8 timeout -k 5 300 python bench.py --pretrain-epochs 200 --steps 5 --warmup 2 --no-cpu-baseline --no-large-minibatch > gpurun_out/bench_trained200_warm.json 2>/dev/null; head -c 200 gpurun_out/bench_trained200_warm.json; echo
timeout -k 5 300 python tools/bench_config3.py 1024 12 > gpurun_out/config3.json 2>/dev/null; tail -c 300 gpurun_out/config3.json
