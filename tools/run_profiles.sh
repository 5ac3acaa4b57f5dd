#!/bin/bash
# Round profile set (GPU box): bench line, rocprofv3 kernel stats of the same command, two PMC passes for HBM traffic,
# and kernel stats of the any-shape / big-rank paths.  usage: bash tools/run_profiles.sh <tag>   (outputs under gpurun_out/)
set -e
TAG=${1:-r01_k}
REPO=$(pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
python bench.py > $OUT/bench.json 2> $OUT/bench.err
cd /tmp
rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/stats -o run -- python3 $REPO/bench.py --no-cpu-baseline > $OUT/stats_bench.json 2> $OUT/stats.err
rocprofv3 --output-format csv --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc_fetch -o run -- python3 $REPO/bench.py --no-cpu-baseline --steps 2 --warmup 1 > /dev/null 2> $OUT/pmc_fetch.err
rocprofv3 --output-format csv --kernel-trace --pmc WRITE_SIZE -d $OUT/pmc_write -o run -- python3 $REPO/bench.py --no-cpu-baseline --steps 2 --warmup 1 > /dev/null 2> $OUT/pmc_write.err
rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/stats_any -o run -- python3 $REPO/tools/bench_anyshape.py 32 20 > $OUT/anyshape.txt 2> $OUT/stats_any.err
rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/stats_rank -o run -- python3 $REPO/tools/dev_rank_sweep.py > $OUT/rank_sweep.txt 2> $OUT/stats_rank.err
cd $REPO
python tools/make_traffic.py $OUT/pmc_fetch $OUT/pmc_write > $OUT/traffic.json
find $OUT -name "*kernel_stats.csv" | head
