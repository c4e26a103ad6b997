#!/bin/bash
# Runs ON THE GPU BOX from the repo root: PMC passes over the sparse apply (tools/prof_sparse.py: 7-point Laplacian 160^3,
# one workgroup per chunk): HBM bytes, wave-cycle split, vector-memory / LDS activity. Output: gpurun_out/pmc_sparse.txt
REPO=$(pwd)
OUT=$REPO/gpurun_out/pmc_sparse
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
P1="FETCH_SIZE"
P2="WRITE_SIZE"
P3="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"
P4="SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_VALU"
P5="TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum"
P6="TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TA_BUSY_sum"
i=0
for P in "$P1" "$P2" "$P3" "$P4" "$P5" "$P6"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --output-format csv --pmc $P -d $OUT/p$i -o pmc -- python $REPO/tools/prof_sparse.py > $OUT/p$i.log 2>&1 || echo "pass $i failed"
done
cd $REPO
python - <<'PY' > gpurun_out/pmc_sparse.txt
import csv, glob, collections
vals = collections.defaultdict(list)
dur = []
for f in glob.glob("gpurun_out/pmc_sparse/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "csc_mul_kernel" in r["Kernel_Name"]:
            vals[r["Counter_Name"]].append(float(r["Counter_Value"]))
for f in glob.glob("gpurun_out/pmc_sparse/p3/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "csc_mul_kernel" in r["Kernel_Name"]:
            dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print("csc_mul_kernel, 7-point Laplacian 160^3 (28.6 M entries, algorithmic 441.6 MB), per launch (mean of the last 5):")
if dur:
    d = sorted(dur)[len(dur) // 2]
    print(f"  duration (median, under PMC collection) {d:8.1f} us")
for k in sorted(vals):
    v = vals[k][-5:]
    m = sum(v) / len(v)
    extra = ""
    if k == "FETCH_SIZE":
        extra = f"  -> {m * 1024 * 2 / 1e6:8.1f} MB read (KiB, x2: gfx950 counts a 128-B request as 64 B)"
    if k == "WRITE_SIZE":
        extra = f"  -> {m * 1024 / 1e6:8.1f} MB written"
    print(f"  {k:32s} {m:16.1f}{extra}")
PY
cat gpurun_out/pmc_sparse.txt
find $OUT -name '*.db' -delete; find $OUT -size +8M -delete
