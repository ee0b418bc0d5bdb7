# round 5 profiles: bench lines, kernel trace of the bench command, FETCH / WRITE of the bench's kernels, k_physics counters (fetch, write, SQ, LDS)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
O=gpurun_out/r5prof
mkdir -p $O
timeout 400 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc $?"
timeout 300 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-large-minibatch > $O/bench_protocol.json 2>> $O/bench.err; echo "protocol rc $?"
python - <<PY
import json
for n in ("bench", "bench_protocol"):
    try:
        d = json.loads([l for l in open("$O/%s.json" % n) if l.startswith("{")][0])
        r = d.get("roofline_update") or {}
        print(n, "value %.0f ms/step %.1f us/opt-step %.2f fps_step %.0f fps_inf %.0f physics ms %.4f" % (d["value"], d["ms_per_step"], r.get("us_per_optimiser_step", 0), d.get("fps_step", 0), d.get("fps_step_and_inference", 0), d.get("roofline_physics", {}).get("avg_launch_ms", 0)))
    except Exception as ex:
        print(n, "unreadable:", ex)
PY
pass() { name=$1; shift; timeout -k 5 240 rocprofv3 "$@" > $O/prof_$name.log 2>&1; echo "$name rc=$?"; }
summ() { db=$(find $O/prof_$1 -name "*_results.db" | head -1); if [ -n "$db" ]; then python tools/rocpd_summary.py $2 $db $O/$3; else echo "no db for $1"; tail -3 $O/prof_$1.log; fi; rm -rf $O/prof_$1; }
SHORT="--steps 1 --warmup 1 --no-cpu-baseline --no-large-minibatch"
pass stats --kernel-trace --stats -d $O/prof_stats -o r5 -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline
summ stats stats bench_kernel_stats.csv
pass bfetch --pmc FETCH_SIZE -d $O/prof_bfetch -o r5 -- python bench.py $SHORT
summ bfetch pmc bench_pmc_fetch.csv
pass bwrite --pmc WRITE_SIZE -d $O/prof_bwrite -o r5 -- python bench.py $SHORT
summ bwrite pmc bench_pmc_write.csv
for c in FETCH_SIZE WRITE_SIZE; do
  pass k$c --pmc $c -d $O/prof_k$c -o r5 -- python tools/time_physics.py 1024 8
  summ k$c pmc kphysics_pmc_$c.csv
done
pass ksq --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU -d $O/prof_ksq -o r5 -- python tools/time_physics.py 1024 8
summ ksq pmc kphysics_pmc_sq.csv
pass klds --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_THREAD_CYCLES_VALU -d $O/prof_klds -o r5 -- python tools/time_physics.py 1024 8
summ klds pmc kphysics_pmc_lds.csv
head -8 $O/bench_kernel_stats.csv | cut -c1-120
grep -E "k_update_persistent|k_physics" $O/bench_pmc_fetch.csv $O/bench_pmc_write.csv | cut -c1-200
grep -h "k_physics" $O/kphysics_pmc_*.csv | grep 524288 | cut -c1-200
