cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 5 120 python tools/time_multi_rank_parts.py 1024 > gpurun_out/mr_parts.txt 2>&1; grep "us per call\|Error\|error" gpurun_out/mr_parts.txt | head
