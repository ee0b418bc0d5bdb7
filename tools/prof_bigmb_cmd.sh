cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
timeout 300 python -m pytest tests/test_gpu_ppo_parity.py -q -m gpu -k "large" 2>&1 | grep -E "^E|passed|failed|Error" | head -20
timeout 200 python tools/quick_train.py 1024 4 BlockAssemblyInsertSim 2>&1 | tail -3
timeout 200 python tools/quick_train.py 1024 4 BlockAssemblyGraspSim 8192 2>&1 | tail -3
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_bigmb -o big -- python tools/quick_train.py 1024 3 BlockAssemblyInsertSim > gpurun_out/prof_bigmb.log 2>&1; echo stats rc=$?
DB=$(find gpurun_out/prof_bigmb -name "*_results.db" | head -1)
python tools/rocpd_summary.py stats $DB gpurun_out/bigmb_kernel_stats.csv && head -16 gpurun_out/bigmb_kernel_stats.csv
