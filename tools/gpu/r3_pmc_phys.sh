# FETCH_SIZE / WRITE_SIZE of k_physics at N = 1024 (separate passes, no tracing flags)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out/r3pmc
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 5 200 rocprofv3 --pmc $c -d gpurun_out/r3pmc/p_$c -o r3 -- python tools/time_physics.py 1024 8 > gpurun_out/r3pmc/prof_$c.log 2>&1; echo "$c rc=$?"
  db=$(find gpurun_out/r3pmc/p_$c -name "*_results.db" | head -1)
  python tools/rocpd_summary.py pmc $db gpurun_out/r3pmc/kphysics_pmc_$c.csv; grep "k_physics" gpurun_out/r3pmc/kphysics_pmc_$c.csv | grep 524288 | cut -c1-160
  rm -rf gpurun_out/r3pmc/p_$c
done
