cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4y
timeout 100 python -m pytest tests/test_gpu_ppo_parity.py -q -m gpu -k "large" 2>&1 | grep -E "passed|failed|^E  " | head -6
timeout 60 python bench.py --num-envs 4096 --minibatch 32768 --no-cpu-baseline --no-large-minibatch --steps 4 --warmup 2 2>/dev/null | grep "^{" > gpurun_out/r4y/bench_n4096_fp32_tt.json
timeout 60 python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | grep "^{" > gpurun_out/r4y/bench_tt.json
python - <<PY
import json
d = json.loads(open("gpurun_out/r4y/bench_n4096_fp32_tt.json").read()); r = d["roofline_update"]
print("n4096 fp32 value %.0f update ms %.2f TFLOP/s %.1f (executed %.1f)" % (d["value"], d["update_ms_per_epoch"], r["achieved"], r["achieved_on_executed_flops"]))
d = json.loads(open("gpurun_out/r4y/bench_tt.json").read()); v = d["large_minibatch_variant"]
print("value %.0f; variant %.0f env-steps/s %.1f TFLOP/s" % (d["value"], v["value"], v["roofline"]["achieved"]))
PY
