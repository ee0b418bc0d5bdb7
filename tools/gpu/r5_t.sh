cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5t
mkdir -p $O
timeout 200 python tests/helpers/parity_stats.py 8 > $O/parity_stats.txt 2>> $O/err.txt
timeout 200 python tests/helpers/parity_keys.py 3 > $O/parity_keys.txt 2>> $O/err.txt
timeout 120 python tools/time_physics.py 1024 8 > $O/time.json 2>> $O/err.txt
python -c "import json;d=json.load(open('$O/time.json'));print(d['k_physics_ms'], d['contacts_mean'])"
grep -v amdgpu $O/parity_stats.txt | cut -c1-330
grep "only on the gpu [1-9]\|gpu only\|oracle only\|^step" $O/parity_keys.txt | cut -c1-300
tail -3 $O/err.txt
