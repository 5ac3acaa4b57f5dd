#!/bin/bash
# HBM traffic of the tridiagonalisation kernels (separate --pmc FETCH_SIZE / WRITE_SIZE passes): 256 matrices 512 x 768, the
# triangle-only kernel and (LRF_ANY_TRIDIAG_SQUARE=1, set by the caller) the full-square one.  usage: bash tools/dev_any_traffic.sh <tag>
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-sym}
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 200 rocprofv3 --output-format csv --kernel-trace --pmc $c -d $R/gpurun_out/anytraffic_${TAG}_$c -o p -- python3 $R/tools/dev_anyshape_stages.py child 256 512 768 102 > /dev/null 2>&1 < /dev/null
done
python3 - "$R/gpurun_out" "$TAG" <<'PY'
import csv, glob, sys, collections
root, tag = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(f"{root}/anytraffic_{tag}_{c}/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"].split("(")[0].replace("void ", "")
            if "tridiag" in k or "gram" in k or "k_any_eig" in k:
                acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k in sorted(acc):
    f = acc[k].get("FETCH_SIZE", [0]); w = acc[k].get("WRITE_SIZE", [0])
    fb, wb = 2 * sum(f) / len(f) * 1024, sum(w) / len(w) * 1024   # gfx950: FETCH_SIZE counts half of wide reads (MI355X_MICROARCH.md)
    print(f"{tag}: {k}: read {fb / 1e9:.1f} GB, written {wb / 1e9:.1f} GB per launch ({len(f)} launches)")
PY
