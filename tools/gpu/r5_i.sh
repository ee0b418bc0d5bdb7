cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5i
mkdir -p $O
timeout 400 python tools/orient_probe.py 1024 1500 600 0.99 > $O/orient_probe.txt 2>&1; grep -E "^stage 0|^T over|^epoch" $O/orient_probe.txt | cut -c1-330; tail -2 $O/orient_probe.txt | cut -c1-300
