#!/bin/bash
# GPU box: the persistent iteration kernel's equality test, then the bench line at the headline and the CLIC-sized workload.
# usage: bash tools/run_persist_check.sh <tag>   (gpurun_out/<tag>_bench.json, <tag>_bench_clic.json)
set -e
TAG=${1:-p}
timeout -k 10 600 python -m pytest tests/test_configs_at_size.py -x -q -m gpu -k persistent 2>&1 | tail -2
python bench.py --no-cpu-baseline > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
python bench.py --config clic --steps 5 --warmup 1 --no-extras > gpurun_out/${TAG}_bench_clic.json 2> gpurun_out/${TAG}_bench_clic.err
python - "$TAG" <<'PY'
import json, sys
for f in ("_bench.json", "_bench_clic.json"):
    d = json.load(open("gpurun_out/" + sys.argv[1] + f))
    print(f, d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["avg_launch_ms"])
PY
