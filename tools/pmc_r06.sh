#!/bin/bash
# Runs ON THE GPU BOX (via gpurun) from the repo root: separate PMC passes (memory-side request counters) over
# tools/pmc_workload_r06.py + one kernel-trace pass for the durations. Output: gpurun_out/prof_r06/summary.txt
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_r06
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o w -- python $REPO/tools/pmc_workload_r06.py > $OUT/stats.log 2>&1
for c in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "TCC_HIT_sum TCC_MISS_sum"; do
  tag=$(echo $c | tr ' ' '+')
  rocprofv3 --kernel-trace --output-format csv --pmc $c -d $OUT/pmc_$tag -o pmc -- python $REPO/tools/pmc_workload_r06.py > $OUT/pmc_$tag.log 2>&1 || echo "pmc pass $tag failed"
done
cd $REPO
python tools/summarize_pmc_r06.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
find $OUT -name '*.db' -delete; find $OUT -size +8M -delete
