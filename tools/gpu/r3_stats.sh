cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out/r3s
timeout -k 5 300 rocprofv3 --kernel-trace --stats -d gpurun_out/r3s/p -o r3 -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-large-minibatch > gpurun_out/r3s/prof.log 2>&1; echo rc=$?
db=$(find gpurun_out/r3s/p -name "*_results.db" | head -1)
python tools/rocpd_summary.py stats $db gpurun_out/r3s/bench_kernel_stats.csv; head -24 gpurun_out/r3s/bench_kernel_stats.csv | cut -c1-150
rm -rf gpurun_out/r3s/p
