cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4y
timeout 600 python -m pytest tests/test_gpu_ppo_parity.py tests/test_gpu_bi_optimization_fullsize.py -q -m gpu -k "large or bf16_update_stays" 2>&1 | grep -E "passed|failed|^E  " | head -8
for mb in 8192 32768; do timeout 120 python tools/time_gemm_nt.py --mb $mb --out gpurun_out/r4y/products_fp32_mb$mb.txt 2>&1 | grep -v amdgpu | cut -c1-118; done
BIG="--num-envs 4096 --minibatch 32768 --no-cpu-baseline --no-large-minibatch --steps 5 --warmup 2"
timeout 200 python bench.py $BIG 2>/dev/null | grep "^{" > gpurun_out/r4y/bench_n4096_fp32.json
timeout 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | grep "^{" > gpurun_out/r4y/bench.json
python - <<PY
import json
d = json.loads(open("gpurun_out/r4y/bench_n4096_fp32.json").read()); r = d["roofline_update"]
print("n4096 fp32 value %.0f update ms %.2f TFLOP/s %.1f (executed %.1f)" % (d["value"], d["update_ms_per_epoch"], r["achieved"], r["achieved_on_executed_flops"]))
d = json.loads(open("gpurun_out/r4y/bench.json").read()); v = d["large_minibatch_variant"]
print("value %.0f; variant %.0f env-steps/s %.1f TFLOP/s" % (d["value"], v["value"], v["roofline"]["achieved"]))
PY
