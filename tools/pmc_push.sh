#!/bin/bash
# Runs ON THE GPU BOX from the repo root: HBM bytes of the one-pass push! kernels from separate PMC
# passes over tools/prof_push.py (m + 3 pushes of InverseLBFGSOperator m = 10 and LBFGSOperator m = 20 at n = 5e7).
# FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts a 128-byte read request as 64 bytes (x2, see
# profiles/traffic_householder.json). Output: gpurun_out/pmc_push.txt
REPO=$(pwd)
OUT=$REPO/gpurun_out/pmc_push
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for kind in "inv 10" "fwd 20"; do
  tag=$(echo $kind | tr ' ' '_')
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --kernel-trace --output-format csv --pmc $c -d $OUT/${tag}_$c -o pmc -- python $REPO/tools/prof_push.py $kind > $OUT/${tag}_$c.log 2>&1 || echo "pass $tag $c failed"
  done
done
cd $REPO
python - <<'PY' > gpurun_out/pmc_push.txt
import csv, glob, collections
n = 50_000_000
for tag, m in (("inv_10", 10), ("fwd_20", 20)):
    tot = collections.defaultdict(lambda: collections.defaultdict(list))
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        for f in glob.glob(f"gpurun_out/pmc_push/{tag}_{c}/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                if "push_pass_kernel" in r["Kernel_Name"]:
                    store = "true, true>" in r["Kernel_Name"].split("(")[0][-14:] or ", true>(" in r["Kernel_Name"][:120].replace("true, true", "X")
                    key = r["Kernel_Name"].split("(")[0][-60:]
                    tot[key][c].append(float(r["Counter_Value"]))
    print(f"== {tag}: push_pass_kernel variants, per-launch averages over the full-memory pushes (last 3 launches of each)")
    total = 0.0
    for key, d in sorted(tot.items()):
        f = d["FETCH_SIZE"][-3:]; w = d["WRITE_SIZE"][-3:]
        fb = sum(f) / len(f) * 1024 * 2; wb = sum(w) / len(w) * 1024
        total += fb + wb
        print(f"   ...{key:60s} read {fb/1e9:6.3f} GB  write {wb/1e9:6.3f} GB")
    # two passes over m - 1 columns + s, y each; stores: s, y (inverse) / s, y, b (forward)
    nec = (2 * (m - 1) + 4 + (2 if tag.startswith("inv") else 3)) * 8.0 * n
    print(f"   sum over one push: {total/1e9:6.3f} GB moved; necessary (DESIGN §4) {nec/1e9:6.3f} GB -> {total/nec:.4f}")
PY
cat gpurun_out/pmc_push.txt
find $OUT -name '*.db' -delete; find $OUT -size +2M -delete
