cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out/r3l
SDXP_LINEAR_TILE=1 timeout -k 5 200 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT -d gpurun_out/r3l/p -o r3 -- python tools/time_act.py 1024 > gpurun_out/r3l/prof.log 2>&1; echo rc=$?
db=$(find gpurun_out/r3l/p -name "*_results.db" | head -1)
python tools/rocpd_summary.py pmc $db gpurun_out/r3l/act_pmc.csv; grep -E "k_linear|k_act" gpurun_out/r3l/act_pmc.csv | cut -c1-170
rm -rf gpurun_out/r3l/p
