cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3d
(SDX_LIB_PATH=$PWD/seqdex_amd/lib/libseqdex_old.so timeout 200 python tools/chain_diag.py 1000 2>&1 | grep -v amdgpu.ids | tail -6) | tee gpurun_out/r3d/old.txt
(timeout 200 python tools/chain_diag.py 1000 2>&1 | grep -v amdgpu.ids | tail -6) | tee gpurun_out/r3d/new.txt
