cd $GRAFT_REPO_ROOT
O=gpurun_out/r4c; mkdir -p $O
timeout 200 python tools/bench_config3.py 1024 1000 --out $O/config3_chain.json 2>/dev/null | cut -c1-200
timeout 300 python tools/chain_repeat.py --reps 5 --out $O/chain_repeat5.json 2>&1 | tail -1 | cut -c1-400
timeout 300 python tools/bench_config5.py --out $O/config5.json 2>&1 | tail -3 | cut -c1-400
