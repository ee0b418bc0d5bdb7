cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5d
mkdir -p $O
V=""
for z in 0.135 0.145 0.155 0.165 0.175; do V="$V;0.9,8,0.05,$z"; done
for z in 0.155 0.165 0.175; do V="$V;1.0,6,0.1,$z"; done
for z in 0.155 0.165; do for x in 0.115 0.135; do V="$V;0.9,8,0.05,$z,$x"; done; done
for z in 0.155 0.165; do for y in 0.01 0.03; do V="$V;0.9,8,0.05,$z,0.125,$y"; done; done
timeout 900 python tools/lift_diag.py 1024 --variants "${V:1}" > $O/lift_scan2.txt 2>$O/lift_scan2.err; grep "variant" $O/lift_scan2.txt | cut -c1-330; tail -3 $O/lift_scan2.err
