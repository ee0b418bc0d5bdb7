cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out
(timeout -k 5 300 python -m pytest tests/test_gpu_physics_parity.py tests/test_gpu_fullsize_properties.py -m gpu -q -x -k "physics or kinematics or free_fall or teacher") 2>&1 | grep -E "passed|failed|Error" | tail -3
timeout -k 5 120 python tools/time_physics.py 1024 24 > gpurun_out/phys_nt512.json 2>/dev/null; cat gpurun_out/phys_nt512.json | cut -c1-200
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 5 200 rocprofv3 --pmc $c -d gpurun_out/prof_r2_$c -o r2 -- python tools/time_physics.py 1024 8 > gpurun_out/prof_$c.log 2>&1; echo "$c rc=$?"
  db=$(find gpurun_out/prof_r2_$c -name "*_results.db" | head -1)
  [ -n "$db" ] && python tools/rocpd_summary.py pmc $db gpurun_out/r2b_kphysics_pmc_$c.csv
  rm -rf gpurun_out/prof_r2_$c
  grep "k_physics" gpurun_out/r2b_kphysics_pmc_$c.csv | cut -c1-150
done
