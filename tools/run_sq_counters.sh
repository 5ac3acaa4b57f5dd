#!/bin/bash
# SQ counter passes over bench.py (GPU box): MFMA busy / VALU / LDS / wait counters per kernel -> gpurun_out/<tag>/sq_counters.csv
# usage: bash tools/run_sq_counters.sh <tag>
set -e
TAG=${1:-r01_l}
REPO=$(pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rocprofv3 --output-format csv --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU GRBM_GUI_ACTIVE -d $OUT/sq_a -o run -- python3 $REPO/bench.py --no-cpu-baseline --steps 2 --warmup 1 > /dev/null 2> $OUT/sq_a.err
rocprofv3 --output-format csv --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $OUT/sq_b -o run -- python3 $REPO/bench.py --no-cpu-baseline --steps 2 --warmup 1 > /dev/null 2> $OUT/sq_b.err
cd $REPO
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for d in ("sq_a", "sq_b"):
    for f in glob.glob(f"{out}/{d}/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"].split("(")[0].replace("void ", "")
            if k.startswith("at::") or "elementwise" in k or "rocclr" in k:
                continue
            acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
with open(f"{out}/sq_counters.csv", "w") as f:
    f.write("kernel,counter,launches,avg_per_launch\n")
    for k in sorted(acc):
        for c in sorted(acc[k]):
            v = acc[k][c]
            f.write(f'"{k}",{c},{len(v)},{sum(v)/len(v):.0f}\n')
print(open(f"{out}/sq_counters.csv").read())
PY
