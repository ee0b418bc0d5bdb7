cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5s
mkdir -p $O
timeout 300 python tests/helpers/parity_keys.py 2 > $O/parity_keys.txt 2> $O/err.txt
grep -v amdgpu $O/parity_keys.txt | cut -c1-260 | head -120
tail -3 $O/err.txt
