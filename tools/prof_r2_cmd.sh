# round-2 profiles (each rocprofv3 pass under its own tight timeout; PMC passes carry no tracing flags)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out
pass() {  # name, rocprof args..., -- command
  name=$1; shift
  timeout -k 5 240 rocprofv3 "$@" > gpurun_out/prof_$name.log 2>&1; echo "$name rc=$?"
}
pass stats --kernel-trace --stats -d gpurun_out/prof_r2_stats -o r2 -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-large-minibatch
pass sq --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU -d gpurun_out/prof_r2_sq -o r2 -- python tools/time_physics.py 1024 8
pass fetch --pmc FETCH_SIZE -d gpurun_out/prof_r2_fetch -o r2 -- python tools/time_physics.py 1024 8
pass write --pmc WRITE_SIZE -d gpurun_out/prof_r2_write -o r2 -- python tools/time_physics.py 1024 8
for n in stats sq fetch write; do
  db=$(find gpurun_out/prof_r2_$n -name "*_results.db" | head -1)
  if [ -n "$db" ]; then
    if [ $n = stats ]; then python tools/rocpd_summary.py stats $db gpurun_out/r2_bench_kernel_stats.csv; else python tools/rocpd_summary.py pmc $db gpurun_out/r2_kphysics_pmc_$n.csv; fi
  else echo "no db for $n"; tail -3 gpurun_out/prof_$n.log; fi
  rm -rf gpurun_out/prof_r2_$n
done
head -12 gpurun_out/r2_bench_kernel_stats.csv | cut -c1-150
grep "k_physics" gpurun_out/r2_kphysics_pmc_sq.csv | cut -c1-200 | head -20
grep "k_physics" gpurun_out/r2_kphysics_pmc_fetch.csv gpurun_out/r2_kphysics_pmc_write.csv | cut -c1-200
