set -e
OUT=$GRAFT_REPO_ROOT/gpurun_out/r05_e
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
python -m pytest tests -x -q -m gpu > $OUT/t_all.log 2>&1 || { tail -40 $OUT/t_all.log; exit 1; }
tail -3 $OUT/t_all.log
python bench.py > $OUT/bench.json 2> $OUT/bench.err || { tail -20 $OUT/bench.err; exit 1; }
python - <<'PY'
import json
d = json.load(open("gpurun_out/r05_e/bench.json"))
print({k: d[k] for k in ("value", "ms_per_step", "roofline")})
for e in d.get("sweep", []): print(e)
print(d.get("kernels"))
PY
