#!/bin/bash
# Runs ON THE GPU BOX from the repo root: the kron 1024^2 leg plain, then under rocprofv3 --kernel-trace (HIP-event
# number printed by the same process), then the trace summary. Output: gpurun_out/kron_reconcile.txt
REPO=$(pwd)
OUT=$REPO/gpurun_out/kron_prof
mkdir -p $OUT
{
  echo "== plain run (no profiler)"
  python tools/kron_reconcile.py 1024
  echo "== the same process under rocprofv3 --kernel-trace"
  (cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o kron -- python $REPO/tools/kron_reconcile.py 1024 2>/dev/null | grep "kron 1024")
  python tools/kron_reconcile.py --summarise $OUT/trace
} > $REPO/gpurun_out/kron_reconcile.txt 2>&1
find $OUT -name '*.db' -delete; find $OUT -size +4M -delete
cat $REPO/gpurun_out/kron_reconcile.txt
