cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5o
mkdir -p $O
timeout 800 python tools/tvalue_sharpness_probe.py 1024 4000 > $O/tvalue_sharpness.txt 2> $O/tvalue_sharpness.err
grep -v "^Setting\|amdgpu.ids" $O/tvalue_sharpness.txt | cut -c1-330
tail -3 $O/tvalue_sharpness.err | cut -c1-300
