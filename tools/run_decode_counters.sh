set -e
REPO=$(pwd); OUT=$REPO/gpurun_out/decpmc; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
rocprofv3 --output-format csv --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE -d $OUT/a -o run -- python3 $REPO/tools/dev_decode_time.py > /dev/null 2> $OUT/a.err
rocprofv3 --output-format csv --kernel-trace --pmc SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAIT_ANY SQ_INST_CYCLES_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $OUT/b -o run -- python3 $REPO/tools/dev_decode_time.py > /dev/null 2> $OUT/b.err
rocprofv3 --output-format csv --kernel-trace --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WR_UNCACHED_32B_sum TCC_EA0_WRREQ_STALL_sum WRITE_SIZE -d $OUT/c -o run -- python3 $REPO/tools/dev_decode_time.py > /dev/null 2> $OUT/c.err
cd $REPO
python3 - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("gpurun_out/decpmc/*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "k_decode16" in row["Kernel_Name"]:
            acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
for k in sorted(acc): print(k, len(acc[k]), sum(acc[k])/len(acc[k]))
PY
