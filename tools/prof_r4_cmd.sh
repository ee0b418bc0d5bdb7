# round-4 measurement pass on one MI355X (gpurun): bench lines, the trunk products alone, rocprofv3 kernel stats, and the PMC passes
# (each rocprofv3 pass under its own tight timeout; PMC passes carry no tracing flags; FETCH_SIZE and WRITE_SIZE in separate passes).
# Everything lands under gpurun_out/r4f/ and the summaries are copied to profiles/r4_* by hand.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
O=gpurun_out/r4f
mkdir -p $O
export SDX_TEST_ARTIFACTS=$R/$O
BIG="--num-envs 4096 --minibatch 32768 --no-cpu-baseline --no-large-minibatch"

if [ "${1:-all}" != "prof" ]; then
  timeout 600 python -m pytest tests/test_gpu_ppo_parity.py tests/test_gpu_fullsize_properties.py -q -m gpu -k "large or capacity or step_for_step" 2>&1 | grep -E "^E|passed|failed|Error" | head -20
  for dt in fp32 bf16; do
    for mb in 8192 32768; do
      f=""; [ $dt = bf16 ] && f="--bf16"
      timeout 120 python tools/time_gemm_nt.py --mb $mb $f --out $O/bigmb_products_${dt}_mb$mb.txt 2>&1 | tail -12
    done
  done
  timeout 400 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc $?"
  timeout 200 python bench.py $BIG --steps 5 --warmup 2 > $O/bench_n4096_fp32.json 2>> $O/bench.err; echo "n4096 fp32 rc $?"
  timeout 200 python bench.py $BIG --mixed-precision --steps 5 --warmup 2 > $O/bench_n4096_bf16.json 2>> $O/bench.err; echo "n4096 bf16 rc $?"
  SDX_FORCE_MULTI_RANK=1 timeout 200 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-large-minibatch > $O/bench_fmr.json 2>> $O/bench.err; echo "fmr rc $?"
  python - <<EOF
import json
for n in ("bench", "bench_n4096_fp32", "bench_n4096_bf16", "bench_fmr"):
    try:
        d = json.loads(open("$O/%s.json" % n).read().strip().splitlines()[-1])
        r = d.get("roofline", {})
        print(n, "value %.0f ms/step %.1f" % (d["value"], d["ms_per_step"]), json.dumps(r.get("bound_actual", r))[:600])
        if "large_minibatch_variant" in d:
            print("   large-minibatch variant", json.dumps(d["large_minibatch_variant"])[:500])
    except Exception as ex:
        print(n, "unreadable:", ex)
EOF
fi

[ "${1:-all}" = "bench" ] && exit 0
timeout 120 python tools/time_gemm_nt.py --mb 32768 --bf16 --out $O/bigmb_products_bf16_mb32768.txt 2>&1 | tail -10
pass() {  # name, rocprof args..., -- command
  name=$1; shift
  timeout -k 5 240 rocprofv3 "$@" > $O/prof_$name.log 2>&1; echo "$name rc=$?"
}
summ() {  # name, mode, csv
  db=$(find $O/prof_$1 -name "*_results.db" | head -1)
  if [ -n "$db" ]; then python tools/rocpd_summary.py $2 $db $O/$3; else echo "no db for $1"; tail -3 $O/prof_$1.log; fi
  rm -rf $O/prof_$1
}
SHORT="--steps 1 --warmup 1 --no-cpu-baseline --no-large-minibatch"
pass stats --kernel-trace --stats -d $O/prof_stats -o r4 -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline
summ stats stats bench_kernel_stats.csv
pass bfetch --pmc FETCH_SIZE -d $O/prof_bfetch -o r4 -- python bench.py $SHORT
summ bfetch pmc bench_pmc_fetch.csv
pass bwrite --pmc WRITE_SIZE -d $O/prof_bwrite -o r4 -- python bench.py $SHORT
summ bwrite pmc bench_pmc_write.csv
pass pfetch --pmc FETCH_SIZE -d $O/prof_pfetch -o r4 -- python tools/time_physics.py 1024 8
summ pfetch pmc kphysics_pmc_fetch.csv
pass pwrite --pmc WRITE_SIZE -d $O/prof_pwrite -o r4 -- python tools/time_physics.py 1024 8
summ pwrite pmc kphysics_pmc_write.csv
MF="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU"
pass mf32 --pmc $MF -d $O/prof_mf32 -o r4 -- python bench.py $BIG --steps 1 --warmup 1
summ mf32 pmc bigmb_fp32_pmc_mfma.csv
pass mbf --pmc $MF -d $O/prof_mbf -o r4 -- python bench.py $BIG --mixed-precision --steps 1 --warmup 1
summ mbf pmc bigmb_bf16_pmc_mfma.csv
pass gfetch --pmc FETCH_SIZE -d $O/prof_gfetch -o r4 -- python bench.py $BIG --mixed-precision --steps 1 --warmup 1
summ gfetch pmc bigmb_bf16_pmc_fetch.csv
pass gstats --kernel-trace --stats -d $O/prof_gstats -o r4 -- python bench.py $BIG --mixed-precision --steps 2 --warmup 1
summ gstats stats bigmb_bf16_kernel_stats.csv
head -14 $O/bench_kernel_stats.csv | cut -c1-150
head -16 $O/bigmb_bf16_kernel_stats.csv | cut -c1-150
grep -E "k_update_persistent|k_physics" $O/bench_pmc_fetch.csv $O/bench_pmc_write.csv $O/kphysics_pmc_fetch.csv $O/kphysics_pmc_write.csv | cut -c1-220 | head -20
grep -E "k_gemm_nt" $O/bigmb_fp32_pmc_mfma.csv $O/bigmb_bf16_pmc_mfma.csv $O/bigmb_bf16_pmc_fetch.csv | cut -c1-260 | head -30
