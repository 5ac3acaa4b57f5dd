# round 5: k_planes16_gram with and without the compiler's packed fp32 math (variant noslp: lrf_encode8 built with -fno-slp-vectorize)
set -e
OUT=$GRAFT_REPO_ROOT/gpurun_out/r05_t
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for lib in liblrf_hip.so liblrf_hip_noslp.so; do
  rocprofv3 --kernel-trace --output-format csv -d $OUT/tr -o run -- python3 $GRAFT_REPO_ROOT/tools/dev_lib_rank.py $lib 7,3,3 256 > $OUT/tr_$lib.log 2>&1
  f=$(find $OUT/tr -name 'run_kernel_trace.csv' | head -1)
  echo "== $lib"; tail -1 $OUT/tr_$lib.log
  python3 - $f <<'PY'
import csv, sys, re, collections
rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.defaultdict(list)
for r in rows:
    nm = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")
    if nm.startswith("at::") or "rocclr" in nm: continue
    acc[nm].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in acc.items():
    v = sorted(v)
    print(f"{k:40s} n {len(v):3d} median {v[len(v)//2]:8.1f} us  min {v[0]:8.1f}")
PY
  rm -rf $OUT/tr
done
