cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 400 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_r1_multi -o r1 -- python tools/time_multi_rank_path.py 256 > gpurun_out/prof_multi.log 2>&1; echo rc=$?
grep -i "multi-rank" gpurun_out/prof_multi.log
