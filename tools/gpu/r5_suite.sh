# the whole -m gpu suite + smoke + default bench line
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5suite
mkdir -p $O
timeout 1500 python -m pytest tests -q -m gpu --durations=15 2>&1 | tail -60 > $O/gputests.txt; tail -40 $O/gputests.txt
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
timeout 400 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc $?"; cut -c1-1500 $O/bench.json
