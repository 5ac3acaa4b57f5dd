# round 5: SQ counters of the persistent kernel on pure rank-16 / rank-8 / rank-26 calls (LDS conflicts, waits, issue)
set -e
OUT=$GRAFT_REPO_ROOT/gpurun_out/r05_ad
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export LRF_SWEEP_BATCH=256
for t in 16,16,16 8,8,8 26,26,26; do
  n=$(echo $t | tr , _)
  rocprofv3 --output-format csv --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $OUT/b_$n -o run -- python3 $GRAFT_REPO_ROOT/tools/dev_rank_sweep.py $t > /dev/null 2> $OUT/b_$n.err
  rocprofv3 --output-format csv --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_LDS -d $OUT/a_$n -o run -- python3 $GRAFT_REPO_ROOT/tools/dev_rank_sweep.py $t > /dev/null 2> $OUT/a_$n.err
  python3 - $OUT $n <<'PY'
import csv, glob, sys, collections
out, n = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(list)
for d in ("a_" + n, "b_" + n):
    for f in glob.glob(f"{out}/{d}/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"].split("(")[0].replace("void ", "")
            if k.startswith("k_bcd_p"):
                acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
print("ranks", n, {c: round(sum(v) / len(v) / 1e6, 1) for c, v in sorted(acc.items())}, "(millions per launch)")
PY
  rm -rf $OUT/a_$n $OUT/b_$n
done
