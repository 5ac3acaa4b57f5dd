# round 5: per-kernel times of the any-shape branches, one patch size per rocprofv3 run (256 images, quality 20)
set -e
OUT=$GRAFT_REPO_ROOT/gpurun_out/r05_n
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for ps in none 16 32 4; do
  rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/st_$ps -o run -- python3 $GRAFT_REPO_ROOT/tools/bench_anyshape.py 256 20 $ps > $OUT/any_$ps.txt 2> $OUT/any_$ps.err
  python3 - $OUT/st_$ps/run_kernel_stats.csv <<'PY' > $OUT/kernels_$ps.txt
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:14]:
    nm = r["Name"].split("(")[0].replace("void ", "")[:60]
    print(f"{nm:60s} calls {r['Calls']:>5s} total_us {float(r['TotalDurationNs'])/1e3:10.1f} avg_us {float(r['AverageNs'])/1e3:9.1f} pct {r['Percentage']}")
PY
  rm -rf $OUT/st_$ps
  echo "== patch $ps"; cat $OUT/any_$ps.txt; cat $OUT/kernels_$ps.txt
done
