cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
CMD="python bench.py --num-envs 4096 --minibatch 32768 --mixed-precision --steps 1 --warmup 1 --no-cpu-baseline --no-large-minibatch"
timeout -k 5 150 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_VALU_MFMA_BUSY_CYCLES -d gpurun_out/p_sq -o r2 -- $CMD > gpurun_out/p_sq.log 2>&1; echo "sq rc=$?"
timeout -k 5 150 rocprofv3 --pmc FETCH_SIZE WRITE_SIZE -d gpurun_out/p_fs -o r2 -- $CMD > gpurun_out/p_fs.log 2>&1; echo "fs rc=$?"
for n in sq fs; do db=$(find gpurun_out/p_$n -name "*_results.db" | head -1); if [ -n "$db" ]; then python tools/rocpd_summary.py pmc $db gpurun_out/r2_bigmb_bf16_pmc_$n.csv; else tail -3 gpurun_out/p_$n.log; fi; rm -rf gpurun_out/p_$n; done
grep "k_gemm" gpurun_out/r2_bigmb_bf16_pmc_sq.csv | cut -c1-160 | head -40
grep "k_gemm" gpurun_out/r2_bigmb_bf16_pmc_fs.csv | cut -c1-160 | head -20
