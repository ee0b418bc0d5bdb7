cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5q
mkdir -p $O
timeout 560 python tools/loop_rounds_probe.py 1024 2 > $O/loop_rounds.txt 2> $O/loop_rounds.err
grep -v "^Setting\|amdgpu.ids" $O/loop_rounds.txt | grep "^round\|^CHAIN" | cut -c1-900
tail -3 $O/loop_rounds.err | cut -c1-300
