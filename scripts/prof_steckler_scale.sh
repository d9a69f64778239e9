# usage: prof_steckler_scale.sh TAG REFINE   -- kernel trace of tests/probe_steckler_scale.py (the real steckler physics through the class layer), breakdown of its last step
TAG=$1; R=${2:-12}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/${TAG}_stprof -o $TAG -- python3 $GRAFT_REPO_ROOT/tests/probe_steckler_scale.py $R 3 $GRAFT_REPO_ROOT/gpurun_out/${TAG}_st_tail.txt > $GRAFT_REPO_ROOT/gpurun_out/${TAG}_steckler_r$R.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(find gpurun_out/${TAG}_stprof -name "*kernel_trace.csv" | head -1)
TOP=60 GAPS=15 python scripts/step_breakdown.py $f --tail-ms $(cat gpurun_out/${TAG}_st_tail.txt) > gpurun_out/${TAG}_steckler_r${R}_last_step.txt
rm -rf gpurun_out/${TAG}_stprof
