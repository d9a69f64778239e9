# usage: pmc_pass.sh TAG EDGE   -- the two PMC passes over a steady time step (FETCH_SIZE, WRITE_SIZE; separate runs) and the summary
TAG=$1; EDGE=${2:-400}
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/${TAG}_pmc_$c -o pmc -- python3 $GRAFT_REPO_ROOT/scripts/pmc_step.py $EDGE 2 > $GRAFT_REPO_ROOT/gpurun_out/${TAG}_pmc_$c.log 2>&1
done
cd $GRAFT_REPO_ROOT
f=$(find gpurun_out/${TAG}_pmc_FETCH_SIZE -name "*counter_collection.csv" | head -1)
w=$(find gpurun_out/${TAG}_pmc_WRITE_SIZE -name "*counter_collection.csv" | head -1)
python scripts/pmc_summary.py $f $w $((EDGE*EDGE*EDGE)) > gpurun_out/${TAG}_pmc_step${EDGE}_summary.txt
rm -rf gpurun_out/${TAG}_pmc_FETCH_SIZE gpurun_out/${TAG}_pmc_WRITE_SIZE
