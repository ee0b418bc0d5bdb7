cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
timeout 600 python bench.py > gpurun_out/bench_r1_v4.json 2> gpurun_out/bench_r1_v4.err; tail -c 600 gpurun_out/bench_r1_v4.json
timeout 400 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_r1v4_stats -o r1 -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/prof_stats.log 2>&1; echo stats rc=$?
timeout 400 rocprofv3 --pmc FETCH_SIZE -d gpurun_out/prof_r1v4_fetch -o r1 -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/prof_fetch.log 2>&1; echo fetch rc=$?
timeout 400 rocprofv3 --pmc WRITE_SIZE -d gpurun_out/prof_r1v4_write -o r1 -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/prof_write.log 2>&1; echo write rc=$?
find gpurun_out/prof_r1v4_stats gpurun_out/prof_r1v4_fetch gpurun_out/prof_r1v4_write -type f | head -30
