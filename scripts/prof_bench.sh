# usage: prof.sh TAG [extra env assignments are exported by the caller]
TAG=$1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/${TAG}_prof -o $TAG -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --radiation-freq 0 --no-class-layer $BENCH_EXTRA > $GRAFT_REPO_ROOT/gpurun_out/${TAG}_prof_bench.json 2> $GRAFT_REPO_ROOT/gpurun_out/${TAG}_prof.err
cd $GRAFT_REPO_ROOT
f=$(find gpurun_out/${TAG}_prof -name "*kernel_trace.csv" | head -1)
python scripts/step_breakdown.py $f > gpurun_out/${TAG}_last_step.txt
s=$(find gpurun_out/${TAG}_prof -name "*kernel_stats.csv" | head -1)
cp $s gpurun_out/${TAG}_kernel_stats.csv
rm -rf gpurun_out/${TAG}_prof
