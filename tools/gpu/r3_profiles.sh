# round-3 profiles: every rocprofv3 pass under its own timeout; PMC passes carry no tracing flags.  Outputs -> gpurun_out/r3prof/ (copied to profiles/ by hand)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
O=gpurun_out/r3prof
mkdir -p $O
pass() {  # name, mode(stats|pmc), rocprof args..., -- command
  name=$1; mode=$2; shift; shift
  timeout -k 5 240 rocprofv3 "$@" > $O/prof_$name.log 2>&1; rc=$?
  db=$(find $O/p_$name -name "*_results.db" 2>/dev/null | head -1)
  if [ -n "$db" ]; then python tools/rocpd_summary.py $mode $db $O/$name.csv; else echo "no db for $name (rc=$rc)"; tail -3 $O/prof_$name.log; fi
  rm -rf $O/p_$name
}
BENCH="python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-large-minibatch"
pass r3_bench_kernel_stats stats --kernel-trace --stats -d $O/p_r3_bench_kernel_stats -o r3 -- $BENCH
pass r3_bench_pmc_fetch pmc --pmc FETCH_SIZE -d $O/p_r3_bench_pmc_fetch -o r3 -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-large-minibatch
pass r3_bench_pmc_write pmc --pmc WRITE_SIZE -d $O/p_r3_bench_pmc_write -o r3 -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-large-minibatch
TP="python tools/time_physics.py 1024 8"
pass r3_kphysics_pmc_sq pmc --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU -d $O/p_r3_kphysics_pmc_sq -o r3 -- $TP
pass r3_kphysics_pmc_lds pmc --pmc SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VALU SQ_WAIT_INST_LDS SQ_THREAD_CYCLES_VALU -d $O/p_r3_kphysics_pmc_lds -o r3 -- $TP
pass r3_kphysics_pmc_fetch pmc --pmc FETCH_SIZE -d $O/p_r3_kphysics_pmc_fetch -o r3 -- $TP
pass r3_kphysics_pmc_write pmc --pmc WRITE_SIZE -d $O/p_r3_kphysics_pmc_write -o r3 -- $TP
export SDX_WARM_START=0
pass r3_kphysics_cold_pmc_fetch pmc --pmc FETCH_SIZE -d $O/p_r3_kphysics_cold_pmc_fetch -o r3 -- $TP
pass r3_kphysics_cold_pmc_write pmc --pmc WRITE_SIZE -d $O/p_r3_kphysics_cold_pmc_write -o r3 -- $TP
unset SDX_WARM_START
BIG="python bench.py --num-envs 4096 --minibatch 32768 --steps 1 --warmup 1 --no-cpu-baseline --no-large-minibatch"
pass r3_bigmb_fp32_kernel_stats stats --kernel-trace --stats -d $O/p_r3_bigmb_fp32_kernel_stats -o r3 -- $BIG
pass r3_bigmb_fp32_pmc_mfma pmc --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU -d $O/p_r3_bigmb_fp32_pmc_mfma -o r3 -- $BIG
pass r3_bigmb_bf16_pmc_mfma pmc --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU -d $O/p_r3_bigmb_bf16_pmc_mfma -o r3 -- $BIG --mixed-precision
pass r3_bigmb_bf16_pmc_fetch pmc --pmc FETCH_SIZE -d $O/p_r3_bigmb_bf16_pmc_fetch -o r3 -- $BIG --mixed-precision
pass r3_bigmb_bf16_pmc_write pmc --pmc WRITE_SIZE -d $O/p_r3_bigmb_bf16_pmc_write -o r3 -- $BIG --mixed-precision
# phase clocks
SDXP_PERSIST_STAMPS=1 timeout 120 python tools/prof_persist.py 1024 > $O/r3_persist_phase_clock.txt 2>&1
for e in 0 19; do SDX_LIB_PATH=$R/seqdex_amd/lib/libseqdex_prof.so SDX_DEBUG_ENV=$e timeout 100 python tools/time_physics.py 1024 8 > $O/r3_kphysics_phase_clock_env$e.json 2>/dev/null; done
timeout 100 python tools/time_physics.py 1024 8 > $O/r3_kphysics_time_n1024.json 2>/dev/null
for n in 512 2048 4096 16384; do timeout 100 python tools/time_physics.py $n 8 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['n_envs'], d['k_physics_ms'], d['env_steps_per_s'], d['contacts_mean'])"; done > $O/r3_kphysics_scaling_with_n.txt
ls -la $O | head -40
grep -h "k_physics\|k_update_persistent\|k_gemm" $O/r3_bench_pmc_fetch.csv $O/r3_bench_pmc_write.csv $O/r3_kphysics_pmc_fetch.csv $O/r3_kphysics_pmc_write.csv $O/r3_kphysics_cold_pmc_fetch.csv $O/r3_kphysics_cold_pmc_write.csv | cut -c1-140
