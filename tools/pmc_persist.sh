#!/bin/bash
# Runs ON THE GPU BOX (via gpurun) from the repo root: memory-side request counters of the persistent quasi-Newton apply with LDS
# parking on / off (separate rocprofv3 --pmc passes). Output: gpurun_out/prof_persist/summary.txt
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_persist
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for L in 1 0; do
  for c in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "TCC_HIT_sum TCC_MISS_sum"; do
    tag=$(echo $c | tr ' ' '+')
    MXLO_QN_PERSIST_LDS=$L rocprofv3 --kernel-trace --output-format csv --pmc $c -d $OUT/lds${L}_$tag -o pmc -- python $REPO/tools/pmc_workload_persist.py > $OUT/lds${L}_$tag.log 2>&1 || echo "pmc pass $L $tag failed"
  done
done
cd $REPO
python - "$OUT" > $OUT/summary.txt <<'PY'
import csv, glob, os, sys
from collections import defaultdict
out = sys.argv[1]
# per setting: kernel dispatches in order (inv m=10 n=2^20 x4, fwd m=5 n=2^20 x4, lsr1 m=20 n=2^19 x4), counters averaged per group of 4
GROUPS = [("InverseLBFGS m=10 n=2^20", (2 * 20 + 3) * 8.0 * (1 << 20)), ("LBFGS m=5 n=2^20", (2 * 10 + 3) * 8.0 * (1 << 20)), ("LSR1 m=20 n=2^19", (2 * 20 + 3) * 8.0 * (1 << 19))]
print("# tools/pmc_persist.sh: memory-side traffic per launch of qn_apply_persist_kernel, LDS parking on / off (one MI355X; rocprofv3 --pmc,")
print("# separate passes; reads = TCC_EA0_RDREQ x 128 B (32-byte ones at 32 B), writes = WRREQ_64B x 64 B + the rest x 32 B); 'two-pass bytes' = (2 ncol + 3) x 8 B x n")
for L in (0, 1):
    ctr = defaultdict(lambda: defaultdict(list))
    for d in sorted(glob.glob(os.path.join(out, f"lds{L}_*"))):
        if not os.path.isdir(d):
            continue
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            rows = [r for r in csv.DictReader(open(f)) if "qn_apply_persist_kernel" in r["Kernel_Name"]]
            rows.sort(key=lambda r: int(r.get("Dispatch_Id", 0)))
            byname = defaultdict(list)
            for r in rows:
                byname[r["Counter_Name"]].append(float(r["Counter_Value"]))
            for name, vals in byname.items():
                for gi in range(len(GROUPS)):
                    ctr[gi][name] += vals[4 * gi:4 * gi + 4]
    for gi, (what, design) in enumerate(GROUPS):
        c = {n: sum(v) / len(v) for n, v in ctr[gi].items() if v}
        if not c:
            print(f"lds {L} {what}: no counters")
            continue
        rd = (c.get("TCC_EA0_RDREQ_sum", 0) - c.get("TCC_EA0_RDREQ_32B_sum", 0)) * 128 + c.get("TCC_EA0_RDREQ_32B_sum", 0) * 32
        wr = c.get("TCC_EA0_WRREQ_64B_sum", 0) * 64 + (c.get("TCC_EA0_WRREQ_sum", 0) - c.get("TCC_EA0_WRREQ_64B_sum", 0)) * 32
        hit, miss = c.get("TCC_HIT_sum", 0), c.get("TCC_MISS_sum", 0)
        print(f"parking {'on ' if L else 'off'} {what:26s}: read {rd / 1e6:7.1f} MB  written {wr / 1e6:6.1f} MB  total {(rd + wr) / 1e6:7.1f} MB = {(rd + wr) / design:5.3f} x the {design / 1e6:.1f} MB of the two-pass bytes; L2 hit rate {hit / max(hit + miss, 1):.3f}")
PY
cat $OUT/summary.txt
find $OUT -name '*.db' -delete; find $OUT -size +8M -delete
