cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 5 200 python tools/drop_test.py 1024 250 0.0 0.8 0.6 > gpurun_out/drop_test.txt 2>&1; tail -3 gpurun_out/drop_test.txt
timeout -k 5 200 python tools/train_curve.py BlockAssemblyInsertSim 2048 1000 250 > gpurun_out/insert_curve_cold.txt 2>&1; tail -4 gpurun_out/insert_curve_cold.txt
SDX_WARM_START=0.8 timeout -k 5 200 python tools/train_curve.py BlockAssemblyInsertSim 2048 1000 250 > gpurun_out/insert_curve_warm.txt 2>&1; tail -4 gpurun_out/insert_curve_warm.txt
