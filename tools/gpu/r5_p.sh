cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5p
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_two_ranks_one_gpu.py -x -q -m gpu -s --durations=5 > $O/two_ranks.txt 2>&1
tail -40 $O/two_ranks.txt | cut -c1-300
