# SQ counters of the rollout's trunk layers launched alone (tools/time_linear.py, default tile shape): two passes, no tracing flags
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out/r3pl
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_INSTS_VALU SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_THREAD_CYCLES_VALU"; do
  i=$((i+1))
  timeout -k 5 150 rocprofv3 --pmc $set -d gpurun_out/r3pl/p$i -o r3 -- python tools/time_linear.py 1024 3 > gpurun_out/r3pl/prof$i.log 2>&1; echo "pass $i rc=$?"
  db=$(find gpurun_out/r3pl/p$i -name "*_results.db" | head -1)
  python tools/rocpd_summary.py pmc $db gpurun_out/r3pl/linear_pmc_$i.csv; grep "k_linear" gpurun_out/r3pl/linear_pmc_$i.csv | cut -c1-150
  rm -rf gpurun_out/r3pl/p$i
done
