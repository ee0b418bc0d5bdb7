cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5u
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_chain.py tests/test_gpu_physics_parity.py -x -q -m gpu -s 2>&1 | grep -v "^Setting\|amdgpu" | tail -12 > $O/chain_test.txt; cat $O/chain_test.txt | cut -c1-300
