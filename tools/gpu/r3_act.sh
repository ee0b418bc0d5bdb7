# round 3, rollout-side kernels after the load-batching / pipeline fixes: direct kernel tests, sdxp_act per tile shape, kernel stats of a short bench
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out/r3a
timeout -k 5 240 python -m pytest -q -m gpu tests/test_gpu_linear_kernel.py tests/test_gpu_vectask_reset.py tests/test_gpu_task_parity.py tests/test_gpu_tvalue_train.py "tests/test_gpu_ppo_parity.py::test_param_layout_and_init" "tests/test_gpu_ppo_parity.py::test_rollout_gae_and_dataset" "tests/test_gpu_ppo_parity.py::test_update_with_narrow_padded_observation_186" -x 2>&1 | tail -15 > gpurun_out/r3a/tests.log; cat gpurun_out/r3a/tests.log
timeout -k 5 120 python tools/time_act.py 1024 2>&1 | tail -10 | tee gpurun_out/r3a/time_act.txt
timeout -k 5 300 rocprofv3 --kernel-trace --stats -d gpurun_out/r3a/p -o r3 -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-large-minibatch > gpurun_out/r3a/prof.log 2>&1; echo rc=$?
tail -1 gpurun_out/r3a/prof.log | cut -c1-900
db=$(find gpurun_out/r3a/p -name "*_results.db" | head -1)
python tools/rocpd_summary.py stats $db gpurun_out/r3a/bench_kernel_stats.csv; head -16 gpurun_out/r3a/bench_kernel_stats.csv | cut -c1-150
rm -rf gpurun_out/r3a/p
