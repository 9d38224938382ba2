# rocprofv3 --pmc passes (counters only: no trace domains) of one command; prints per-kernel sums.
#   bash tools/gpu_pmc_job.sh TAG "COUNTERS ..." -- command...
TAG=$1; SETS=$2; shift 3
R=$PWD; mkdir -p $R/gpurun_out; cd /tmp; export TMPDIR=/tmp
i=0
IFS='|' read -ra SETARR <<< "$SETS"
for set in "${SETARR[@]}"; do
  i=$((i+1))
  rm -rf $R/gpurun_out/pmc_${TAG}_$i
  (cd $R && timeout 400 rocprofv3 --pmc $set -d $R/gpurun_out/pmc_${TAG}_$i -o p -- "$@" > $R/gpurun_out/pmc_${TAG}_$i.log 2>&1)
  (cd $R && python tools/rocprof_pmc_sum.py $(find gpurun_out/pmc_${TAG}_$i -name "*.db" | head -1) >> $R/gpurun_out/pmc_${TAG}.txt 2>&1)
  rm -rf $R/gpurun_out/pmc_${TAG}_$i
done
cat $R/gpurun_out/pmc_${TAG}.txt
