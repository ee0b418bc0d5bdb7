cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU -d gpurun_out/prof_r1_sq -o r1 -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/prof_sq.log 2>&1; echo sq rc=$?
tail -3 gpurun_out/prof_sq.log | cut -c1-300
