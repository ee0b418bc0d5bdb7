cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5w
mkdir -p $O
SDX_FORCE_MULTI_RANK=1 timeout 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-large-minibatch > $O/bench_forced_multi.json 2> $O/err.txt; echo rc $?
python - <<PY
import json
d = json.loads([l for l in open("$O/bench_forced_multi.json") if l.startswith("{")][0])
print(d["value"], d["ms_per_step"], d["update_path"], d.get("roofline_update", {}).get("us_per_optimiser_step"))
PY
tail -3 $O/err.txt | cut -c1-300
