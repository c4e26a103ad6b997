#!/bin/bash
# rocprofv3 --hip-trace view of the allocation / synchronisation contract: two runs of tools/contract_trace.py that
# differ only in the number of warmed iterations and of push! calls; tools/summarize_hip_trace.py subtracts the per-API
# call counts. Usage (on the GPU box):  bash tools/contract_trace.sh gpurun_out/contract
set -u
OUT=${1:-gpurun_out/contract}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$ROOT/$OUT"
cd /tmp && export TMPDIR=/tmp
run() {  # name iters pushes
  rm -rf "/tmp/ct_$1"
  timeout 300 rocprofv3 --hip-trace --stats -d "/tmp/ct_$1" -o "$1" --output-format csv -- \
      python "$ROOT/tools/contract_trace.py" --iters "$2" --pushes "$3" > "/tmp/ct_$1.log" 2>&1
  f=$(find "/tmp/ct_$1" -name "*hip_api_stats.csv" 2>/dev/null | head -1)
  if [ -n "$f" ]; then cp "$f" "$ROOT/$OUT/$1_hip_api_stats.csv"; else echo "no stats for $1"; tail -5 "/tmp/ct_$1.log"; fi
}
run base 100 10
run iters 1100 10
run pushes 100 1010
python "$ROOT/tools/summarize_hip_trace.py" "$ROOT/$OUT" | tee "$ROOT/$OUT/summary.txt"
