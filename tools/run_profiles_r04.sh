#!/bin/bash
# Round-4 profile set (GPU box).  usage: bash tools/run_profiles_r04.sh <tag>   (outputs under gpurun_out/<tag>/)
#   bench lines of the three configs, rocprofv3 kernel stats of the default bench command (with and without the extra legs),
#   two PMC passes for HBM traffic, two SQ counter passes, the rank sweep at 64 and 256 images with kernel stats at the
#   ranks of the new kernels ((20,10,10), (26,13,13)) and their SQ counters / HBM traffic, the CLIC-sized and svd stats,
#   the config-3 R-D table (per image and batched), the any-shape branches.
# rocprofv3 gets `python3 <script>` directly after `--` (no env / shell hop), counters in passes of their own.
set -e
TAG=${1:-r04_a}
REPO=$(pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
python bench.py > $OUT/bench.json 2> $OUT/bench.err
python bench.py --config clic --steps 5 --warmup 1 > $OUT/bench_clic.json 2> $OUT/bench_clic.err
python bench.py --config svd --steps 5 --warmup 1 > $OUT/bench_svd.json 2> $OUT/bench_svd.err
python tools/run_config3.py $OUT/config3.json > $OUT/config3.txt 2> $OUT/config3.err
python tools/run_config3.py $OUT/config3_batched.json batched > $OUT/config3_batched.txt 2> $OUT/config3_batched.err
LRF_SWEEP_BATCH=256 python3 $REPO/tools/dev_rank_sweep.py > $OUT/rank_sweep256.txt 2> $OUT/rank_sweep256.err
python3 $REPO/tools/dev_rank_sweep.py > $OUT/rank_sweep64.txt 2> $OUT/rank_sweep64.err
LRF_NO_BCDW32=1 LRF_SWEEP_BATCH=256 python3 $REPO/tools/dev_rank_sweep.py 20,10,10 > $OUT/rank_sweep256_mid_20.txt 2>/dev/null
LRF_NO_BCDW32=1 LRF_SWEEP_BATCH=256 python3 $REPO/tools/dev_rank_sweep.py 26,13,13 > $OUT/rank_sweep256_mid_26.txt 2>/dev/null
cd /tmp
rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/stats -o run -- python3 $REPO/bench.py --no-cpu-baseline > $OUT/stats_bench.json 2> $OUT/stats.err
rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/stats_plain -o run -- python3 $REPO/bench.py --no-extras > $OUT/stats_plain_bench.json 2> $OUT/stats_plain.err
rocprofv3 --output-format csv --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc_fetch -o run -- python3 $REPO/bench.py --no-extras --steps 2 --warmup 1 > /dev/null 2> $OUT/pmc_fetch.err
rocprofv3 --output-format csv --kernel-trace --pmc WRITE_SIZE -d $OUT/pmc_write -o run -- python3 $REPO/bench.py --no-extras --steps 2 --warmup 1 > /dev/null 2> $OUT/pmc_write.err
rocprofv3 --output-format csv --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU GRBM_GUI_ACTIVE -d $OUT/sq_a -o run -- python3 $REPO/bench.py --no-extras --steps 2 --warmup 1 > /dev/null 2> $OUT/sq_a.err
rocprofv3 --output-format csv --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $OUT/sq_b -o run -- python3 $REPO/bench.py --no-extras --steps 2 --warmup 1 > /dev/null 2> $OUT/sq_b.err
export LRF_SWEEP_BATCH=256
rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/stats_r20 -o run -- python3 $REPO/tools/dev_rank_sweep.py 20,10,10 > $OUT/r20.txt 2> $OUT/stats_r20.err
rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/stats_r26 -o run -- python3 $REPO/tools/dev_rank_sweep.py 26,13,13 > $OUT/r26.txt 2> $OUT/stats_r26.err
rocprofv3 --output-format csv --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc_fetch_r20 -o run -- python3 $REPO/tools/dev_rank_sweep.py 20,10,10 > /dev/null 2> $OUT/pmc_fetch_r20.err
rocprofv3 --output-format csv --kernel-trace --pmc WRITE_SIZE -d $OUT/pmc_write_r20 -o run -- python3 $REPO/tools/dev_rank_sweep.py 20,10,10 > /dev/null 2> $OUT/pmc_write_r20.err
rocprofv3 --output-format csv --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU GRBM_GUI_ACTIVE -d $OUT/sq_a_r20 -o run -- python3 $REPO/tools/dev_rank_sweep.py 20,10,10 > /dev/null 2> $OUT/sq_a_r20.err
rocprofv3 --output-format csv --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $OUT/sq_b_r20 -o run -- python3 $REPO/tools/dev_rank_sweep.py 20,10,10 > /dev/null 2> $OUT/sq_b_r20.err
unset LRF_SWEEP_BATCH
rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/stats_svd -o run -- python3 $REPO/bench.py --config svd --steps 3 --warmup 1 > /dev/null 2> $OUT/stats_svd.err
rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/stats_clic -o run -- python3 $REPO/bench.py --config clic --steps 3 --warmup 1 --no-extras > /dev/null 2> $OUT/stats_clic.err
cd $REPO
python tools/bench_anyshape.py 256 20 > $OUT/anyshape.txt 2> $OUT/anyshape.err
python tools/make_traffic.py $OUT/pmc_fetch $OUT/pmc_write > $OUT/traffic.json
python tools/make_traffic.py $OUT/pmc_fetch_r20 $OUT/pmc_write_r20 > $OUT/traffic_r20.json
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for tag, dirs in (("sq_counters", ("sq_a", "sq_b")), ("sq_counters_r20", ("sq_a_r20", "sq_b_r20"))):
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
find $OUT -name "*kernel_stats.csv"
