# round 3, call 1: diagnostic SQ counters of k_physics (LDS bank conflicts, instruction mix) before the restructuring
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out/r3c1
rocprofv3 -L 2>/dev/null | grep -oE "SQ_[A-Z_0-9]+" | sort -u > gpurun_out/r3c1/sq_counters.txt
wc -l gpurun_out/r3c1/sq_counters.txt
pass() {  # name, rocprof args..., -- command
  name=$1; shift
  timeout -k 5 200 rocprofv3 "$@" > gpurun_out/r3c1/prof_$name.log 2>&1; echo "$name rc=$?"
}
pass lds --pmc SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS -d gpurun_out/r3c1/p_lds -o r3 -- python tools/time_physics.py 1024 8
pass mix --pmc SQ_WAVE_CYCLES SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU -d gpurun_out/r3c1/p_mix -o r3 -- python tools/time_physics.py 1024 8
for n in lds mix; do
  db=$(find gpurun_out/r3c1/p_$n -name "*_results.db" | head -1)
  if [ -n "$db" ]; then python tools/rocpd_summary.py pmc $db gpurun_out/r3c1/kphysics_pmc_$n.csv; grep "k_physics" gpurun_out/r3c1/kphysics_pmc_$n.csv | grep 524288 | cut -c1-160
  else echo "no db for $n"; tail -5 gpurun_out/r3c1/prof_$n.log; fi
  rm -rf gpurun_out/r3c1/p_$n
done
for it in 16 8 1; do SDX_TP_ITERS=$it timeout 120 python tools/time_physics.py 1024 8 2>&1 | tail -1 | cut -c1-200; done
