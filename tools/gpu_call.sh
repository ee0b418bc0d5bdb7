cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(time timeout -k 5 900 python -m pytest tests -m gpu -q) > gpurun_out/gputests.log 2>&1
grep -E "passed|failed|Error " gpurun_out/gputests.log | tail -6
timeout -k 5 120 python tools/time_physics.py > gpurun_out/time_physics.txt 2>&1; tail -12 gpurun_out/time_physics.txt
timeout -k 5 200 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-large-minibatch > gpurun_out/bench_quick.json 2>/dev/null; head -c 200 gpurun_out/bench_quick.json; echo
timeout -k 5 300 python bench.py --pretrain-epochs 200 --steps 5 --warmup 2 --no-cpu-baseline --no-large-minibatch > gpurun_out/bench_trained200.json 2>/dev/null; head -c 200 gpurun_out/bench_trained200.json; echo
