# learned chain exploration + k_physics timing after the parallel box-pair stage
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5g
mkdir -p $O
timeout 500 python tools/chain_learned.py 1024 1500 1500 --orient-gate 0.99 0.9 0.8 0.5 0.3 0.0 --out $O/chain_learned.json > $O/chain.log 2> $O/chain.err; grep -E "grasp policy|insert policy" $O/chain.err | cut -c1-700; python - <<PY
import json
try:
    d=json.load(open("$O/chain_learned.json"))
    print("chain_error:", d.get("chain_error"))
    c=d.get("chain") or {}
    for k in ("orient","grasp","insert"):
        if k in c: print(k, json.dumps(c[k])[:900])
    print("value", d.get("value"), "total_wall_s", d.get("total_wall_s"))
except Exception as ex:
    print("unreadable", ex)
PY
tail -5 $O/chain.err | cut -c1-400
