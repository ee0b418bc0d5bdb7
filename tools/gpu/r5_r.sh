cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5r
mkdir -p $O
for v in hip nofma; do
  export SDX_LIB_PATH=$PWD/seqdex_amd/lib/libseqdex_$v.so
  echo "== $v" >> $O/parity_stats.txt
  timeout 200 python tests/helpers/parity_stats.py 6 >> $O/parity_stats.txt 2>> $O/err.txt
  timeout 120 python tools/time_physics.py 1024 8 > $O/time_$v.json 2>> $O/err.txt
  python -c "import json;d=json.load(open('$O/time_$v.json'));print('$v', d['k_physics_ms'], d['contacts_mean'])"
done
grep -v amdgpu $O/parity_stats.txt | cut -c1-330
tail -3 $O/err.txt
