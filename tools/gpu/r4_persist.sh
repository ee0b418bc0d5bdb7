cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4h
timeout 600 python -m pytest tests/test_gpu_ppo_parity.py tests/test_gpu_fullsize_properties.py -q -m gpu -k "update_matches or persistent or fault or explicit or step_for_step or replicas or reproduc or emulated or nccl" 2>&1 | grep -E "passed|failed|^E  " | head -8
SDXP_PERSIST_STAMPS=1 timeout 120 python tools/prof_persist.py 1024 2>&1 | grep -v amdgpu | tee gpurun_out/r4h/phase_clock.txt | head -24
timeout 120 python tools/prof_persist.py 1024 2>&1 | grep "update"
SDX_FORCE_MULTI_RANK=1 timeout 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-large-minibatch 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][0]); print('forced multi-rank value', d['value'], d['roofline_update']['us_per_optimiser_step'])"
