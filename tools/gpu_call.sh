cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(time timeout -k 5 900 python -m pytest tests -m gpu -q) > gpurun_out/gputests.log 2>&1
grep -E "passed|failed|Error " gpurun_out/gputests.log | tail -4
for n in 512 1024 2048 4096 16384; do timeout -k 5 150 python tools/time_physics.py $n 8 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(json.dumps({k:d[k] for k in ('n_envs','k_physics_ms','env_steps_per_s','contacts_mean','contacts_max')}))"; done > gpurun_out/kphysics_scaling.txt; cat gpurun_out/kphysics_scaling.txt
timeout -k 5 100 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
