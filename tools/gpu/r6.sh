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
  db=$(find $O/prof -name "*_results.db" | head -1); [ -n "$db" ] && python tools/rocpd_summary.py stats $db $O/kernel_stats_mb$MB.csv && python tools/rocpd_summary.py bygrid $db $O/kernel_bygrid_mb$MB.csv; rm -rf $O/prof
  grep -i "gemm\|reduce\|adam\|stage\|heads\|sqnorm\|fin" $O/kernel_bygrid_mb$MB.csv | head -40 | cut -c1-150
  head -40 $O/kernel_stats_mb$MB.csv | cut -c1-150
  ;;
ppo_tests)   # PPO parity tests on the device (-k "$1" optional)
  timeout 1200 python -m pytest tests/test_gpu_ppo_parity.py -q -m gpu -x ${1:+-k "$1"} 2>&1 | tail -15 > $O/tests.txt; tail -8 $O/tests.txt
  ;;
pclock)   # phase clock of the persistent update kernel only ($1 = tag, further args = variant libraries)
  tagp=${1:-head}; shift
  SDXP_PERSIST_STAMPS=1 timeout 200 python tools/prof_persist.py 1024 > $O/phase_clock_$tagp.txt 2> /dev/null; cat $O/phase_clock_$tagp.txt
  for v in "$@"; do
    SDX_LIB_PATH=$PWD/seqdex_amd/lib/libseqdex_$v.so SDXP_PERSIST_STAMPS=1 timeout 200 python tools/prof_persist.py 1024 > $O/phase_clock_${tagp}_$v.txt 2> /dev/null; echo "== $v"; cat $O/phase_clock_${tagp}_$v.txt
  done
  ;;
multirank)   # the multi-rank optimiser step on one GPU: its tests, then the forced world-1 bench line with the one-launch and the three-launch apply ($1 = tag)
  tagp=${1:-head}
  timeout 900 python -m pytest tests/test_gpu_fullsize_properties.py tests/test_gpu_two_ranks_one_gpu.py -q -m gpu -x -k "one_launch_apply or factor_exchange or rccl_world1 or two_processes" 2>&1 | tail -60 > $O/tests_$tagp.txt; tail -4 $O/tests_$tagp.txt
  SDXP_APPLY_IMPL=fused SDX_FORCE_MULTI_RANK=1 timeout 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-large-minibatch > $O/bench_fused_$tagp.json 2> $O/bench_fused_$tagp.err; echo "fused rc $?"
  SDX_FORCE_MULTI_RANK=1 timeout 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-large-minibatch > $O/bench_three_$tagp.json 2> $O/bench_three_$tagp.err; echo "three-launch rc $?"
  for t in fused three; do python -c "
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][0])
print(sys.argv[2], 'value %.0f ms/step %.2f' % (d['value'], d['ms_per_step']), 'update_path', d.get('update_path'), json.dumps(d.get('roofline_update') or d.get('roofline'))[:300])
" $O/bench_${t}_$tagp.json $t; done
  if [ -n "$2" ]; then   # $2 = trace: kernel durations of both forms (rocprofv3 kernel trace of a one-epoch run)
    for t in fused three; do
      if [ $t = fused ]; then export SDXP_APPLY_IMPL=fused; else unset SDXP_APPLY_IMPL; fi
      SDX_FORCE_MULTI_RANK=1 timeout -k 5 300 rocprofv3 --kernel-trace --stats -d $O/prof_$t -o r6 -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-large-minibatch > $O/prof_$t.log 2>&1; echo "prof $t rc $?"
      db=$(find $O/prof_$t -name "*_results.db" | head -1); [ -n "$db" ] && python tools/rocpd_summary.py stats $db $O/kernel_stats_${t}_$tagp.csv; rm -rf $O/prof_$t
      head -8 $O/kernel_stats_${t}_$tagp.csv | cut -c1-160
    done
    unset SDXP_APPLY_IMPL
  fi
  ;;
persist)   # the persistent update kernel: its oracle / determinism / fault tests, phase clock of CU 0, short bench line ($1 = tag of the output files)
  tagp=${1:-head}
  timeout 900 python -m pytest tests/test_gpu_ppo_parity.py tests/test_gpu_fullsize_properties.py -q -m gpu -x -k "update_matches_autograd_adam or persistent or narrow_padded or explicit_gradient_path" 2>&1 | tail -8 > $O/tests_$tagp.txt; tail -4 $O/tests_$tagp.txt
  SDXP_PERSIST_STAMPS=1 timeout 200 python tools/prof_persist.py 1024 > $O/phase_clock_$tagp.txt 2> /dev/null; cat $O/phase_clock_$tagp.txt
  shift; for v in "$@"; do   # further libraries (seqdex_amd/lib/libseqdex_<v>.so): phase clock only
    SDX_LIB_PATH=$PWD/seqdex_amd/lib/libseqdex_$v.so SDXP_PERSIST_STAMPS=1 timeout 200 python tools/prof_persist.py 1024 > $O/phase_clock_${tagp}_$v.txt 2> /dev/null; echo "== $v"; cat $O/phase_clock_${tagp}_$v.txt
  done
  timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-large-minibatch > $O/bench_$tagp.json 2> $O/bench_$tagp.err; echo "bench rc $?"
  python - <<PY
import json
d = json.loads([l for l in open("$O/bench_$tagp.json") if l.startswith("{")][0])
r = d.get("roofline_update") or d.get("roofline") or {}
print("value %.0f ms/step %.2f" % (d["value"], d["ms_per_step"]), json.dumps(r)[:500])
PY
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
sweep)   # training runs: $1 = name of the output, $2 = run list "mb:lr:seed:epochs,...", $3 = per-run time box (s)
  timeout 3400 python tools/train_sweep.py --runs "$2" --out $O/$1.json --max-seconds ${3:-100000} 2> $O/$1.err | cut -c1-600; tail -2 $O/$1.err | cut -c1-300
  ;;
profiles)   # the round's evidence: bench lines, kernel trace of the bench command, FETCH / WRITE of its kernels, k_physics counters + phase clocks
  timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc $?"
  timeout 300 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-large-minibatch > $O/bench_protocol.json 2>> $O/bench.err; echo "protocol rc $?"
  SDX_FORCE_MULTI_RANK=1 timeout 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-large-minibatch > $O/bench_forced_multi_rank_world1.json 2>> $O/bench.err; echo "forced multi-rank rc $?"
  pass() { name=$1; shift; timeout -k 5 240 rocprofv3 "$@" > $O/prof_$name.log 2>&1; echo "$name rc=$?"; }
  summ() { db=$(find $O/prof_$1 -name "*_results.db" | head -1); if [ -n "$db" ]; then python tools/rocpd_summary.py $2 $db $O/$3; else echo "no db for $1"; tail -3 $O/prof_$1.log; fi; rm -rf $O/prof_$1; }
  SHORT="--steps 1 --warmup 1 --no-cpu-baseline --no-large-minibatch"
  pass stats --kernel-trace --stats -d $O/prof_stats -o r6 -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline
  summ stats stats bench_kernel_stats.csv
  pass bfetch --pmc FETCH_SIZE -d $O/prof_bfetch -o r6 -- python bench.py $SHORT
  summ bfetch pmc bench_pmc_fetch.csv
  pass bwrite --pmc WRITE_SIZE -d $O/prof_bwrite -o r6 -- python bench.py $SHORT
  summ bwrite pmc bench_pmc_write.csv
  for c in FETCH_SIZE WRITE_SIZE; do
    pass k$c --pmc $c -d $O/prof_k$c -o r6 -- python tools/time_physics.py 1024 8
    summ k$c pmc kphysics_pmc_$c.csv
  done
  pass ksq --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU -d $O/prof_ksq -o r6 -- python tools/time_physics.py 1024 8
  summ ksq pmc kphysics_pmc_sq.csv
  pass klds --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_THREAD_CYCLES_VALU -d $O/prof_klds -o r6 -- python tools/time_physics.py 1024 8
  summ klds pmc kphysics_pmc_lds.csv
  timeout 150 python tools/time_physics.py 1024 24 > $O/kphysics_time_n1024.json 2>/dev/null; digest $O/kphysics_time_n1024.json
  for e in 0 1; do SDX_LIB_PATH=$PWD/seqdex_amd/lib/libseqdex_prof.so SDX_DEBUG_ENV=$e timeout 150 python tools/time_physics.py 1024 24 > $O/kphysics_phase_clock_env$e.json 2>/dev/null; done
  SDX_LIB_PATH=$PWD/seqdex_amd/lib/libseqdex_prof.so timeout 300 python tools/ablate_physics.py 1024 24 > $O/kphysics_ablation.json 2>/dev/null
  head -6 $O/bench_kernel_stats.csv | cut -c1-120
  grep -E "k_update_persistent|k_physics" $O/bench_pmc_fetch.csv $O/bench_pmc_write.csv | cut -c1-200
  grep -h "k_physics" $O/kphysics_pmc_*.csv | grep 524288 | cut -c1-200
  ;;
chain_closed)   # the all-learned chain over seeds: $1 = output name, rest = arguments of tools/chain_closed.py
  name=$1; shift
  timeout 3400 python tools/chain_closed.py "$@" --out $O/$name.json 2> $O/$name.err | grep -v "^Setting\|amdgpu" | cut -c1-1500; tail -3 $O/$name.err | cut -c1-400
  ;;
biopt_long)   # bi-optimisation rounds at a length that inserts: $1 = output name, rest = arguments of tools/biopt_long.py
  name=$1; shift
  timeout 3400 python tools/biopt_long.py "$@" --out $O/$name.json 2> $O/$name.err | grep -v "^Setting\|amdgpu\|^fps step" | cut -c1-400 | tail -60; tail -3 $O/$name.err | cut -c1-400
  ;;
suite)   # the whole -m gpu suite as the driver runs it, then smoke()
  timeout 3000 python -m pytest tests/ -q -m gpu --durations=15 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl" | tail -60 > $O/gputests.txt; grep -E "passed|failed|error" $O/gputests.txt | tail -3
  timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
  ;;
*) echo "unknown job $job"; exit 2 ;;
esac
