#!/bin/bash
# Runs ON THE GPU BOX (via gpurun) from the repo root: rocprofv3 kernel-trace stats of bench.py and
# separate PMC passes (HBM fetch / write counters) of a fixed workload. Outputs under gpurun_out/prof/.
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bench -- python $REPO/bench.py --steps 20 --warmup 5 --no-traffic > $OUT/bench_under_rocprof.log 2>&1
for c in FETCH_SIZE WRITE_SIZE "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "TCC_HIT_sum TCC_MISS_sum"; do
  tag=$(echo $c | tr ' ' '+')
  rocprofv3 --kernel-trace --output-format csv --pmc $c -d $OUT/pmc_$tag -o pmc -- python $REPO/tools/pmc_workload.py > $OUT/pmc_$tag.log 2>&1 || echo "pmc pass $tag failed"
done
cd $REPO
python tools/summarize_prof.py $OUT > $OUT/summary.txt 2>&1
tail -40 $OUT/summary.txt
# keep only the small CSVs (the merge-back limit is 64 MiB)
find $OUT -name '*.db' -delete; find $OUT -size +8M -delete
