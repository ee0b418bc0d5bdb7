cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5n
mkdir -p $O
timeout 900 python tools/chain_learned.py 1024 1500 1500 --out $O/chain_learned.json > $O/chain_learned.txt 2> $O/chain_learned.err
tail -3 $O/chain_learned.err | cut -c1-300
python - <<PY
import json
d = json.load(open("$O/chain_learned.json"))
print(json.dumps(d["insert_policy_refit(untimed)"]))
print(json.dumps(d["chain"]["orient"])[:600])
print(json.dumps(d["chain"]["grasp"])[:700])
print(json.dumps(d["chain"]["insert"]))
print(d["value"], d["total_wall_s"])
PY
