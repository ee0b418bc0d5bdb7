cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 -d gpurun_out/prof_mfma -o big -- python tools/quick_train.py 1024 2 BlockAssemblyInsertSim > gpurun_out/prof_mfma.log 2>&1; echo mfma rc=$?
tail -2 gpurun_out/prof_mfma.log | cut -c1-200
DB=$(find gpurun_out/prof_mfma -name "*_results.db" | head -1)
python tools/rocpd_summary.py pmc $DB gpurun_out/bigmb_pmc_mfma.csv && grep -E "k_gemm|kernel" gpurun_out/bigmb_pmc_mfma.csv | head -40
