cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5x
mkdir -p $O
timeout 280 python -m pytest tests/test_gpu_chain.py -x -q -m gpu -s 2>&1 | grep -v "^Setting\|amdgpu" | tail -8 > $O/chain_test.txt; cat $O/chain_test.txt | cut -c1-300
