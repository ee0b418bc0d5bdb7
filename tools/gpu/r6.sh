#!/bin/bash
# Round-6 GPU jobs, one parameterised script (replaces the one-off r5_[a-x].sh files):  gpurun -- 'bash tools/gpu/r6.sh <job> [args...]'
# Every job writes under gpurun_out/r6_<job>/ and prints a short digest; the files that are kept go to profiles/r6_*.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
job=$1; shift
O=gpurun_out/r6_$job
mkdir -p $O
digest() { python - "$@" <<'PY'
import json, sys
for f in sys.argv[1:]:
    try:
        d = json.loads([l for l in open(f) if l.startswith("{")][-1])
        print(f.split("/")[-1], round(d["k_physics_ms"], 4), "contacts", round(d["contacts_mean"], 1), d["contacts_max"], "robot envs", d["envs_with_robot_contact"])
        if "counts_env_substep0" in d and any(d["counts_env_substep0"].values()):
            print("   counts", d["counts_env_substep0"])
            print("   phases", {k: v for k, v in d["phase_cycles_env0_substep0"].items()})
    except Exception as ex:
        print(f, "unreadable:", ex)
PY
}
case $job in
phys)   # k_physics variants side by side: libseqdex_<v>.so for v in "$@" (default: what is in seqdex_amd/lib), parity tests on the default library first
  timeout 900 python -m pytest tests/test_gpu_physics_parity.py tests/test_gpu_seams.py -q -m gpu -x 2>&1 | tail -6 > $O/tests.txt; tail -3 $O/tests.txt
  V="$@"; [ -z "$V" ] && V=$(ls seqdex_amd/lib | sed -n 's/^libseqdex_\(.*\)\.so$/\1/p' | grep -v prof)
  for v in $V; do
    SDX_LIB_PATH=$PWD/seqdex_amd/lib/libseqdex_$v.so timeout 150 python tools/time_physics.py 1024 24 > $O/time_$v.json 2> $O/time_$v.err || tail -2 $O/time_$v.err
    digest $O/time_$v.json
  done
  for v in $(ls seqdex_amd/lib | sed -n 's/^libseqdex_\(.*prof\)\.so$/\1/p'); do for e in 0 1; do
    SDX_LIB_PATH=$PWD/seqdex_amd/lib/libseqdex_$v.so SDX_DEBUG_ENV=$e timeout 150 python tools/time_physics.py 1024 24 > $O/phase_${v}_env$e.json 2> /dev/null
    digest $O/phase_${v}_env$e.json
  done; done
  ;;
ablate)   # launch-level attribution of k_physics (profiling build)
  SDX_LIB_PATH=$PWD/seqdex_amd/lib/libseqdex_prof.so timeout 300 python tools/ablate_physics.py 1024 24 > $O/ablate.json 2> $O/ablate.err || tail -3 $O/ablate.err
  cat $O/ablate.json
  ;;
bigmb_trace)   # per-kernel times of the large-minibatch update at minibatch $1 (default 2048): bench line + rocprofv3 kernel trace
  MB=${1:-2048}; shift
  timeout 300 python bench.py --minibatch $MB --steps 5 --warmup 2 --no-cpu-baseline "$@" > $O/bench_mb$MB.json 2> $O/bench_mb$MB.err; echo "bench rc $?"
  python - <<PY
import json
d = json.loads([l for l in open("$O/bench_mb$MB.json") if l.startswith("{")][0])
print("value %.0f ms/step %.2f update_path %s" % (d["value"], d["ms_per_step"], d.get("update_path")), json.dumps(d.get("roofline_update"))[:600])
PY
  timeout -k 5 300 rocprofv3 --kernel-trace --stats -d $O/prof -o r6 -- python bench.py --minibatch $MB --steps 2 --warmup 1 --no-cpu-baseline "$@" > $O/prof.log 2>&1; echo "prof rc $?"
  db=$(find $O/prof -name "*_results.db" | head -1); [ -n "$db" ] && python tools/rocpd_summary.py stats $db $O/kernel_stats_mb$MB.csv; rm -rf $O/prof
  head -40 $O/kernel_stats_mb$MB.csv | cut -c1-150
  ;;
ppo_tests)   # PPO parity tests on the device (-k "$1" optional)
  timeout 1200 python -m pytest tests/test_gpu_ppo_parity.py -q -m gpu -x ${1:+-k "$1"} 2>&1 | tail -15 > $O/tests.txt; tail -8 $O/tests.txt
  ;;
bench)   # the default bench line (+ args)
  timeout 600 python bench.py "$@" > $O/bench.json 2> $O/bench.err; echo "bench rc $?"
  python - <<PY
import json
d = json.loads([l for l in open("$O/bench.json") if l.startswith("{")][0])
r = d.get("roofline_update") or {}
print("value %.0f ms/step %.2f us/opt-step %.2f fps_step %.0f fps_inf %.0f physics ms %.4f" % (d["value"], d["ms_per_step"], r.get("us_per_optimiser_step", 0), d.get("fps_step", 0), d.get("fps_step_and_inference", 0), (d.get("roofline_physics") or {}).get("avg_launch_ms", 0)))
v = d.get("large_minibatch_variant") or {}
print("large-minibatch variant:", {k: v.get(k) for k in ("value", "minibatch", "ms_per_step")}, json.dumps(v.get("roofline"))[:400])
print("cpu_baseline:", json.dumps(d.get("cpu_baseline"))[:300])
PY
  ;;
*) echo "unknown job $job"; exit 2 ;;
esac
