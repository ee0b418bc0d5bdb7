# round 3, closing run: the whole -m gpu suite, smoke(), and the default bench line
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3f
timeout 900 python -m pytest tests -q -m gpu -x 2>&1 | tail -12 | tee gpurun_out/r3f/gputests.txt
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee gpurun_out/r3f/smoke.txt
timeout 600 python bench.py > gpurun_out/r3f/bench.json 2> gpurun_out/r3f/bench.err; echo bench rc=$?
tail -c 1500 gpurun_out/r3f/bench.json
