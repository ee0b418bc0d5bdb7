cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(time timeout -k 5 900 python -m pytest tests -m gpu -q) > gpurun_out/gputests.log 2>&1
grep -E "passed|failed|Error " gpurun_out/gputests.log | tail -4
timeout -k 5 100 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
