#!/bin/bash
# Copies the judged summaries of a profile run (tools/run_profiles_r05.sh <tag>) from gpurun_out/<tag>/ into profiles/<tag>_*.
# usage: bash tools/collect_profiles.sh r05_a
set -e
T=$1
S=gpurun_out/$T
D=profiles
cp $S/bench.json $D/${T}_bench.json
cp $S/bench_clic.json $D/${T}_bench_clic.json
cp $S/bench_svd.json $D/${T}_bench_svd.json
cp $S/config3_fused.json $D/${T}_config3_fused.json
cp $S/rank_sweep256.txt $D/${T}_rank_sweep_256_images.txt
cp $S/rank_sweep64.txt $D/${T}_rank_sweep.txt
cp $S/anyshape.txt $D/${T}_anyshape.txt
cp $S/stats_plain/run_kernel_stats.csv $D/${T}_kernel_stats.csv
cp $S/stats_plain_bench.json $D/${T}_kernel_stats_bench_line.json
cp $S/stats/run_kernel_stats.csv $D/${T}_kernel_stats_with_extras.csv
cp $S/stats_clic/run_kernel_stats.csv $D/${T}_clic_kernel_stats.csv
cp $S/stats_svd/run_kernel_stats.csv $D/${T}_svd_kernel_stats.csv
cp $S/sq_counters.csv $D/${T}_sq_counters_per_kernel.csv
cp $S/traffic.json $D/${T}_traffic.json
cp $S/traffic.json $D/traffic_latest.json
for n in 16_8_8 26_13_13; do
  cp $S/stats_r$n/run_kernel_stats.csv $D/${T}_rank${n}_kernel_stats.csv
  cp $S/sq_counters_r$n.csv $D/${T}_rank${n}_sq_counters_per_kernel.csv
  cp $S/traffic_r$n.json $D/${T}_rank${n}_traffic.json
  cp $S/timeline_r$n.txt $D/${T}_rank${n}_timeline.txt
done
# rocprofv3 prints kernel names with full torch template arguments: keep the library's rows and the totals short
for f in $D/${T}_*kernel_stats*.csv; do
  python3 - "$f" <<'PY'
import csv, sys
p = sys.argv[1]
rows = list(csv.reader(open(p)))
out = [rows[0]] + [[(c if i else c[:160]) for i, c in enumerate(r)] for r in rows[1:]]
csv.writer(open(p, "w"), quoting=csv.QUOTE_ALL).writerows(out)
PY
done
ls -la $D | grep $T | wc -l
