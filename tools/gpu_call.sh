cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 5 200 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_bf16 -o r2 -- python bench.py --num-envs 4096 --minibatch 32768 --mixed-precision --steps 2 --warmup 1 --no-cpu-baseline --no-large-minibatch > gpurun_out/prof_bf16.log 2>&1; echo "rc=$?"
db=$(find gpurun_out/prof_bf16 -name "*_results.db" | head -1)
if [ -n "$db" ]; then python tools/rocpd_summary.py stats $db gpurun_out/r2_bigmb_bf16_kernel_stats.csv; head -16 gpurun_out/r2_bigmb_bf16_kernel_stats.csv | cut -c1-140; else tail -5 gpurun_out/prof_bf16.log; fi
rm -rf gpurun_out/prof_bf16
grep '"metric"' gpurun_out/prof_bf16.log | head -c 300
