#!/bin/bash
# SQ counter passes (two, counters only) over tools/dev_rank_sweep.py at one rank triple, 256 images.
# usage: bash tools/run_sq_rank.sh <tag> <ranks e.g. 20,10,10>     (outputs under gpurun_out/<tag>/)
set -e
TAG=${1:-r04_sq}
RANKS=${2:-20,10,10}
REPO=$(pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
export LRF_SWEEP_BATCH=256
cd /tmp
rocprofv3 --output-format csv --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU GRBM_GUI_ACTIVE -d $OUT/sq_a -o run -- python3 $REPO/tools/dev_rank_sweep.py $RANKS > /dev/null 2> $OUT/sq_a.err
rocprofv3 --output-format csv --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $OUT/sq_b -o run -- python3 $REPO/tools/dev_rank_sweep.py $RANKS > /dev/null 2> $OUT/sq_b.err
rocprofv3 --output-format csv --kernel-trace --pmc SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS -d $OUT/sq_c -o run -- python3 $REPO/tools/dev_rank_sweep.py $RANKS > /dev/null 2> $OUT/sq_c.err || true
cd $REPO
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for d in ("sq_a", "sq_b", "sq_c"):
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
PY
cat $OUT/sq_counters.csv | grep "w32\|mid" 
