set -e
REPO=$(pwd); OUT=$REPO/gpurun_out/single; mkdir -p $OUT; export TMPDIR=/tmp
python tools/dev_single_image.py > $OUT/single.txt 2>&1
cd /tmp
rocprofv3 --output-format csv --kernel-trace --memory-copy-trace -d $OUT/trace -o run -- python3 $REPO/tools/dev_single_image.py > /dev/null 2> $OUT/trace.err
cd $REPO
cat $OUT/single.txt
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/single/trace/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# find a late window: last 400 kernels; print a sequence of one encode (from k_planes16 to next k_planes16)
idx = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("k_planes16")]
i0, i1 = idx[5], idx[6]
t0 = int(rows[i0]["Start_Timestamp"])
prev_end = t0
for r in rows[i0:i1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f'{(s - t0)/1e3:8.1f} us  gap {(s - prev_end)/1e3:6.1f}  dur {(e - s)/1e3:6.1f}  {r["Kernel_Name"][:40]}')
    prev_end = e
PY
