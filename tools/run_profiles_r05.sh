#!/bin/bash
# Round-5 profile set (GPU box).  usage: bash tools/run_profiles_r05.sh <tag>   (outputs under gpurun_out/<tag>/)
#   bench lines of the three configs (the default one carries the `sweep` object), rocprofv3 kernel stats of the default bench
#   command (with and without the extra legs), two PMC passes for HBM traffic, two SQ counter passes, the rank sweep at 64 and
#   256 images, kernel stats + time lines + HBM traffic at the ranks of the new persistent instantiations ((16,8,8), (26,13,13)),
#   the CLIC-sized and svd stats, config 3 as one fused sweep call, the any-shape branches.
# rocprofv3 gets `python3 <script>` directly after `--` (no env / shell hop), counters in passes of their own.
set -e
TAG=${1:-r05_a}
REPO=$(pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
python bench.py > $OUT/bench.json 2> $OUT/bench.err
python bench.py --config clic --steps 5 --warmup 1 > $OUT/bench_clic.json 2> $OUT/bench_clic.err
python bench.py --config svd --steps 5 --warmup 1 > $OUT/bench_svd.json 2> $OUT/bench_svd.err
python tools/run_config3_fused.py $OUT/config3_fused.json > $OUT/config3_fused.txt 2> $OUT/config3_fused.err
LRF_SWEEP_BATCH=256 python3 $REPO/tools/dev_rank_sweep.py > $OUT/rank_sweep256.txt 2> $OUT/rank_sweep256.err
python3 $REPO/tools/dev_rank_sweep.py > $OUT/rank_sweep64.txt 2> $OUT/rank_sweep64.err
cd /tmp
rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/stats -o run -- python3 $REPO/bench.py --no-cpu-baseline > $OUT/stats_bench.json 2> $OUT/stats.err
rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/stats_plain -o run -- python3 $REPO/bench.py --no-extras > $OUT/stats_plain_bench.json 2> $OUT/stats_plain.err
rocprofv3 --output-format csv --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc_fetch -o run -- python3 $REPO/bench.py --no-extras --steps 2 --warmup 1 > /dev/null 2> $OUT/pmc_fetch.err
rocprofv3 --output-format csv --kernel-trace --pmc WRITE_SIZE -d $OUT/pmc_write -o run -- python3 $REPO/bench.py --no-extras --steps 2 --warmup 1 > /dev/null 2> $OUT/pmc_write.err
rocprofv3 --output-format csv --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU GRBM_GUI_ACTIVE -d $OUT/sq_a -o run -- python3 $REPO/bench.py --no-extras --steps 2 --warmup 1 > /dev/null 2> $OUT/sq_a.err
rocprofv3 --output-format csv --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $OUT/sq_b -o run -- python3 $REPO/bench.py --no-extras --steps 2 --warmup 1 > /dev/null 2> $OUT/sq_b.err
export LRF_SWEEP_BATCH=256
for t in 16,8,8 26,13,13; do
  n=$(echo $t | tr , _)
  rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/stats_r$n -o run -- python3 $REPO/tools/dev_rank_sweep.py $t > $OUT/r$n.txt 2> $OUT/stats_r$n.err
  rocprofv3 --output-format csv --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc_fetch_r$n -o run -- python3 $REPO/tools/dev_rank_sweep.py $t > /dev/null 2> $OUT/pmc_fetch_r$n.err
  rocprofv3 --output-format csv --kernel-trace --pmc WRITE_SIZE -d $OUT/pmc_write_r$n -o run -- python3 $REPO/tools/dev_rank_sweep.py $t > /dev/null 2> $OUT/pmc_write_r$n.err
  rocprofv3 --output-format csv --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU GRBM_GUI_ACTIVE -d $OUT/sq_a_r$n -o run -- python3 $REPO/tools/dev_rank_sweep.py $t > /dev/null 2> $OUT/sq_a_r$n.err
done
unset LRF_SWEEP_BATCH
rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/stats_svd -o run -- python3 $REPO/bench.py --config svd --steps 3 --warmup 1 > /dev/null 2> $OUT/stats_svd.err
rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/stats_clic -o run -- python3 $REPO/bench.py --config clic --steps 3 --warmup 1 --no-extras > /dev/null 2> $OUT/stats_clic.err
cd $REPO
python tools/bench_anyshape.py 256 20 > $OUT/anyshape.txt 2> $OUT/anyshape.err
python tools/make_traffic.py $OUT/pmc_fetch $OUT/pmc_write > $OUT/traffic.json
for n in 16_8_8 26_13_13; do
  python tools/make_traffic.py $OUT/pmc_fetch_r$n $OUT/pmc_write_r$n > $OUT/traffic_r$n.json
  python tools/dev_trace_timeline.py $OUT/stats_r$n/run_kernel_trace.csv 4 > $OUT/timeline_r$n.txt
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for tag, dirs in (("sq_counters", ("sq_a", "sq_b")), ("sq_counters_r16_8_8", ("sq_a_r16_8_8",)), ("sq_counters_r26_13_13", ("sq_a_r26_13_13",))):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for d in dirs:
        for f in glob.glob(f"{out}/{d}/**/*counter_collection.csv", recursive=True):
            for row in csv.DictReader(open(f)):
                k = row["Kernel_Name"].split("(")[0].replace("void ", "")
                if k.startswith("at::") or "elementwise" in k or "rocclr" in k:
                    continue
                acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
    with open(f"{out}/{tag}.csv", "w") as f:
        f.write("kernel,counter,launches,avg_per_launch\n")
        for k in sorted(acc):
            for c in sorted(acc[k]):
                v = acc[k][c]
                f.write(f'"{k}",{c},{len(v)},{sum(v)/len(v):.0f}\n')
PY
# keep the merge-back small: the raw traces stay on the box
find $OUT -name "*kernel_trace.csv" -size +2M -delete
find $OUT -name "*counter_collection.csv" -delete
find $OUT -name "*kernel_stats.csv"
