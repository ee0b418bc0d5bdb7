cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5l
mkdir -p $O
SDX_TEST_ARTIFACTS=$PWD/$O timeout 900 python -m pytest tests/test_gpu_bi_optimization_fullsize.py -x -q -m gpu -k round_at_4096 -s > $O/biopt_test.txt 2>&1
tail -5 $O/biopt_test.txt | cut -c1-400
grep -E "^(forward|backward) " $O/biopt_test.txt | cut -c1-200
