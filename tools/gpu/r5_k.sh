cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5k
mkdir -p $O
python - > $O/policy_lift.json 2> $O/policy_lift.err <<PY
import json, sys
sys.path.insert(0, ".")
from seqdex_amd.scripts.evaluation import train_grasp_policy
path, task, st = train_grasp_policy(1024, 3000, seed=22, lift_statistics=True)
task.sim.close()
print(json.dumps(st))
PY
cut -c1-1200 $O/policy_lift.json; tail -2 $O/policy_lift.err | cut -c1-300
SDXP_NT_TILE=1 SDX_LIB_PATH=$PWD/seqdex_amd/lib/libseqdex_branchy.so timeout 120 python tools/diag_gemm_nt.py > $O/gemm_nt_branchy_build.txt 2>&1; grep -E "^TN bf0|^=== " $O/gemm_nt_branchy_build.txt | head -12
