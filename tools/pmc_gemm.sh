#!/bin/bash
# Runs ON THE GPU BOX (via gpurun) from the repo root: rocprofv3 PMC passes (own runs, --kernel-trace only) over the
# kron GEMM kernels at n = 1024 f64 — matrix-pipe busy, issue/wait split, LDS conflicts, L2 hit rate.
#   tools/pmc_gemm.sh <tag> <--only substring> [<--only substring> ...]
REPO=$(pwd)
TAG=$1; shift
OUT=$REPO/gpurun_out/pmc_gemm_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS=""
for o in "$@"; do ARGS="$ARGS --only \"$o\""; done
P1="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE GRBM_COUNT"
P2="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM"
P3="TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum"
P4="SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC SQ_WAIT_INST_ANY SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA"
i=0
for P in "$P1" "$P2" "$P3" "$P4"; do
  i=$((i+1))
  eval rocprofv3 --kernel-trace --output-format csv --pmc $P -d $OUT/p$i -o pmc -- $REPO/tools/tune_gemm --f64 --reps 20 $ARGS 1024 > $OUT/p$i.log 2>&1 || echo "pass $i failed"
done
cd $REPO
python tools/summarize_pmc.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
find $OUT -name '*.db' -delete; find $OUT -size +8M -delete
