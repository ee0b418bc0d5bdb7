cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_ppo_parity.py -q -m gpu -k "update_matches or persistent or fault or explicit" 2>&1 | grep -E "passed|failed|^E  " | head -5
SDXP_PERSIST_STAMPS=1 timeout 120 python tools/prof_persist.py 1024 2>&1 | grep -v amdgpu | head -22
timeout 120 python tools/prof_persist.py 1024 2>&1 | grep "update"
