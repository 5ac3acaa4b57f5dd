# round 5: k_any_eig's stages (LRF_DEBUG_INIT_SWEEPS = stop_after: 1 tridiagonal form read, 2 eigenvalues, 3 twisted vectors, 4
# Gram-Schmidt, 0 everything) at patch=False and 16x16 patches, 256 images, quality 20: kernel durations from a trace
set -e
OUT=$GRAFT_REPO_ROOT/gpurun_out/r05_ab
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for ps in none 16; do
  for s in 1 2 3 4 0; do
    export LRF_DEBUG_INIT_SWEEPS=$s
    rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/st -o run -- python3 $GRAFT_REPO_ROOT/tools/bench_anyshape.py 256 20 $ps > $OUT/any.txt 2> $OUT/any.err || true
    python3 - $OUT/st/run_kernel_stats.csv "$ps stop_after=$s" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    nm = r["Name"].split("(")[0].replace("void ", "")
    if nm.startswith("k_any_eig"):
        print(f"{sys.argv[2]:24s} {nm:16s} calls {r['Calls']:>3s} avg_us {float(r['AverageNs'])/1e3:9.1f}")
PY
    rm -rf $OUT/st
  done
done
