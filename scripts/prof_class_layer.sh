# usage: prof_class_layer.sh TAG EDGE   -- kernel trace of the class-layer path (scripts/class_layer_probe.py), breakdown of its last step
TAG=$1; EDGE=${2:-400}
cd /tmp && export TMPDIR=/tmp
PROBE_COMPILED=0 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/${TAG}_clprof -o $TAG -- python3 $GRAFT_REPO_ROOT/scripts/class_layer_probe.py $EDGE 2 $GRAFT_REPO_ROOT/gpurun_out/${TAG}_cl_tail.txt > $GRAFT_REPO_ROOT/gpurun_out/${TAG}_class_layer_$EDGE.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(find gpurun_out/${TAG}_clprof -name "*kernel_trace.csv" | head -1)
TOP=70 python scripts/step_breakdown.py $f --tail-ms $(cat gpurun_out/${TAG}_cl_tail.txt) > gpurun_out/${TAG}_class_layer_${EDGE}_last_step.txt
rm -rf gpurun_out/${TAG}_clprof
