# (the SDXP_NT_STAGGER sweep below belonged to an experiment that was measured - no effect - and removed from csrc/sdx_gemm_nt.h: DESIGN.md section 12;
# kept as the record of what was run)
mkdir -p gpurun_out/r4g
for q in 0 8 16 32; do echo "== bf16 stagger $q"; SDXP_NT_STAGGER=$q timeout 120 python tools/time_gemm_nt.py --mb 32768 --bf16 2>&1 | grep -E "forward|data grad|all eight" | cut -c1-110; done
for q in 0 16 32 64; do echo "== fp32 stagger $q"; SDXP_NT_STAGGER=$q timeout 120 python tools/time_gemm_nt.py --mb 32768 2>&1 | grep -E "forward|data grad|all eight" | cut -c1-110; done
timeout 300 python -m pytest tests/test_gpu_chain.py -q -m gpu 2>&1 | tail -5
timeout 200 python tools/bench_config3.py 1024 1000 --out gpurun_out/r4g/config3_chain.json 2>/dev/null | cut -c1-300
timeout 300 python tools/chain_repeat.py --reps 5 --out gpurun_out/r4g/chain_repeat5.json 2>&1 | tail -4 | cut -c1-500
