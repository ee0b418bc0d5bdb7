# closing GPU pass of round 4: whole -m gpu suite, bench lines, kernel trace and traffic counters of the closing library
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
O=gpurun_out/r4zz
mkdir -p $O
timeout 120 python -m pytest tests/test_gpu_fullsize_properties.py -q -m gpu -s -k "pump_energy" 2>&1 | grep -E "fastest|bricks outside|passed|failed|^E " | head
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -15 > $O/gputests.txt; tail -6 $O/gputests.txt
timeout 400 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc $?"
timeout 300 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-large-minibatch > $O/bench_protocol.json 2>> $O/bench.err; echo "protocol rc $?"
SDX_FORCE_MULTI_RANK=1 timeout 200 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-large-minibatch > $O/bench_fmr.json 2>> $O/bench.err; echo "fmr rc $?"
python - <<PY
import json
for n in ("bench", "bench_protocol", "bench_trained200", "bench_fmr"):
    try:
        d = json.loads([l for l in open("$O/%s.json" % n) if l.startswith("{")][0])
        r = d.get("roofline_update") or {}
        print(n, "value %.0f ms/step %.1f us/opt-step %.2f fps_step %.0f fps_inf %.0f" % (d["value"], d["ms_per_step"], r.get("us_per_optimiser_step", 0), d.get("fps_step", 0), d.get("fps_step_and_inference", 0)))
        if "large_minibatch_variant" in d:
            v = d["large_minibatch_variant"]; print("   large-minibatch variant %.0f env-steps/s, %.1f TFLOP/s" % (v["value"], v["roofline"]["achieved"]))
        if n == "bench":
            print("   frac_of_floor", d["roofline"]["bound_actual"].get("frac_of_floor"), "physics", json.dumps(d.get("roofline_physics", {}))[:400])
    except Exception as ex:
        print(n, "unreadable:", ex)
PY
pass() { name=$1; shift; timeout -k 5 240 rocprofv3 "$@" > $O/prof_$name.log 2>&1; echo "$name rc=$?"; }
summ() { db=$(find $O/prof_$1 -name "*_results.db" | head -1); if [ -n "$db" ]; then python tools/rocpd_summary.py $2 $db $O/$3; else echo "no db for $1"; tail -3 $O/prof_$1.log; fi; rm -rf $O/prof_$1; }
SHORT="--steps 1 --warmup 1 --no-cpu-baseline --no-large-minibatch"
pass stats --kernel-trace --stats -d $O/prof_stats -o r4 -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline
summ stats stats bench_kernel_stats.csv
pass bfetch --pmc FETCH_SIZE -d $O/prof_bfetch -o r4 -- python bench.py $SHORT
summ bfetch pmc bench_pmc_fetch.csv
pass bwrite --pmc WRITE_SIZE -d $O/prof_bwrite -o r4 -- python bench.py $SHORT
summ bwrite pmc bench_pmc_write.csv
head -8 $O/bench_kernel_stats.csv | cut -c1-120
grep -E "k_update_persistent" $O/bench_pmc_fetch.csv $O/bench_pmc_write.csv | cut -c1-200
