cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5m
mkdir -p $O
timeout 600 python tools/insert_refit_probe.py 1024 1000 200 --gate 0.5 > $O/insert_refit.txt 2> $O/insert_refit.err
grep -v "^Setting\|amdgpu.ids" $O/insert_refit.txt | cut -c1-420
tail -3 $O/insert_refit.err | cut -c1-300
