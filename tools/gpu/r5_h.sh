cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5h
mkdir -p $O
timeout 150 python tools/insert_probe.py 1024 1500 250 > $O/ins_synth.txt 2>&1; grep -E "^epoch|wall_s" $O/ins_synth.txt | cut -c1-260
timeout 150 python tools/insert_probe.py 1024 1500 250 --no-hollow > $O/ins_synth_nohollow.txt 2>&1; grep -E "^epoch|wall_s" $O/ins_synth_nohollow.txt | cut -c1-260
timeout 200 python tools/insert_probe.py 1024 1500 250 --grasp 1500 > $O/ins_real.txt 2>&1; grep -E "^epoch|wall_s" $O/ins_real.txt | cut -c1-260
tail -3 $O/ins_synth.txt | cut -c1-300
