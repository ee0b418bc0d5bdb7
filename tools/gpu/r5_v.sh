# closing library: kernel trace of the bench command + the protocol line
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
O=gpurun_out/r5v
mkdir -p $O
pass() { name=$1; shift; timeout -k 5 240 rocprofv3 "$@" > $O/prof_$name.log 2>&1; echo "$name rc=$?"; }
summ() { db=$(find $O/prof_$1 -name "*_results.db" | head -1); if [ -n "$db" ]; then python tools/rocpd_summary.py $2 $db $O/$3; else echo "no db for $1"; tail -3 $O/prof_$1.log; fi; rm -rf $O/prof_$1; }
pass stats --kernel-trace --stats -d $O/prof_stats -o r5 -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline
summ stats stats bench_kernel_stats.csv
head -8 $O/bench_kernel_stats.csv | cut -c1-120
timeout 200 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-large-minibatch > $O/bench_protocol.json 2> $O/bench.err; echo "protocol rc $?"; cut -c1-400 $O/bench_protocol.json
