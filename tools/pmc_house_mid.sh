#!/bin/bash
# Runs ON THE GPU BOX (via gpurun) from the repo root: memory-side request counters of opHouseholder at n = 2^22 as one launch (two workgroups
# per CU) and as two launches. Output: gpurun_out/prof_house_mid/summary.txt
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_house_mid
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for L in 2 1; do
  for c in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"; do
    tag=$(echo $c | tr ' ' '+')
    MXLO_HOUSE_FUSED_PER_CU=$L rocprofv3 --kernel-trace --output-format csv --pmc $c -d $OUT/per${L}_$tag -o pmc -- python $REPO/tools/pmc_workload_house_mid.py > $OUT/per${L}_$tag.log 2>&1 || echo "pmc pass $L $tag failed"
  done
done
cd $REPO
python - "$OUT" > $OUT/summary.txt <<'PY'
import csv, glob, os, sys
from collections import defaultdict
out = sys.argv[1]
n = 1 << 22
print("# tools/pmc_house_mid.sh: memory-side traffic per APPLY of opHouseholder mul! at n = 2^22 fp64 (one MI355X; rocprofv3 --pmc, separate passes;")
print("# reads = TCC_EA0_RDREQ x 128 B (32-byte ones at 32 B), writes = WRREQ_64B x 64 B + the rest x 32 B), summed over the kernels of one apply")
for L, what in ((2, "one launch, two workgroups per CU (slices of h, v in registers)"), (1, "two launches (dots pass + update pass)")):
    tot = defaultdict(float)
    names = set()
    for d in sorted(glob.glob(os.path.join(out, f"per{L}_*"))):
        if not os.path.isdir(d):
            continue
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                k = r["Kernel_Name"]
                if "householder_fused" in k or "panel_dots" in k or "HouseholderOp" in k or "HouseholderPartialsOp" in k:
                    tot[r["Counter_Name"]] += float(r["Counter_Value"]) / 4.0          # 4 applies
                    names.add(k.split("(")[0][-60:])
    rd = (tot["TCC_EA0_RDREQ_sum"] - tot["TCC_EA0_RDREQ_32B_sum"]) * 128 + tot["TCC_EA0_RDREQ_32B_sum"] * 32
    wr = tot["TCC_EA0_WRREQ_64B_sum"] * 64 + (tot["TCC_EA0_WRREQ_sum"] - tot["TCC_EA0_WRREQ_64B_sum"]) * 32
    print(f"{what:66s}: read {rd / 1e6:6.1f} MB  written {wr / 1e6:5.1f} MB  total {(rd + wr) / 1e6:6.1f} MB = {(rd + wr) / n:4.1f} B/elt   kernels: {sorted(names)}")
PY
cat $OUT/summary.txt
find $OUT -name '*.db' -delete; find $OUT -size +8M -delete
