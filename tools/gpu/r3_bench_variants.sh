cd $GRAFT_REPO_ROOT
O=gpurun_out/r3bv; mkdir -p $O
timeout 200 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-large-minibatch 2>/dev/null | tail -1 > $O/r3_bench_protocol.json
timeout 300 python bench.py --pretrain-epochs 200 --steps 5 --warmup 2 --no-cpu-baseline --no-large-minibatch 2>/dev/null | tail -1 > $O/r3_bench_trained200.json
timeout 200 python bench.py --num-envs 4096 --minibatch 32768 --steps 5 --warmup 2 --no-cpu-baseline --no-large-minibatch 2>/dev/null | tail -1 > $O/r3_bench_n4096_fp32.json
timeout 200 python bench.py --num-envs 4096 --minibatch 32768 --mixed-precision --steps 5 --warmup 2 --no-cpu-baseline --no-large-minibatch 2>/dev/null | tail -1 > $O/r3_bench_n4096_bf16.json
SDX_FORCE_MULTI_RANK=1 timeout 200 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-large-minibatch 2>/dev/null | tail -1 > $O/r3_bench_fmr.json
for f in protocol trained200 n4096_fp32 n4096_bf16 fmr; do python -c "
import json; d=json.load(open('$O/r3_bench_$f.json')); rp=d['roofline_physics']; ru=d['roofline_update']
print('$f', round(d['value']), round(d['fps_step']), round(d['fps_step_and_inference']), round(d['update_ms_per_epoch'],1), round(d['rollout_ms_per_epoch'],2), round(rp['avg_launch_ms'],3), rp['contacts_per_env_max_since_create'], rp['env_substeps_over_capacity_since_create'], rp['env_substeps_rebuilt_without_speculative_contacts'], ru.get('frac'), ru.get('achieved'), ru.get('us_per_optimiser_step'))"; done
