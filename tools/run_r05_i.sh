# round 5: A/B of two library builds on one box (argv: the two file names under lrf_amd/), alternating
set -e
OUT=$GRAFT_REPO_ROOT/gpurun_out/r05_i
mkdir -p $OUT
rm -f $OUT/ab.txt
cd $GRAFT_REPO_ROOT
for r in 7,3,3 16,8,8 8,8,8 16,16,16; do
  for rep in 1 2; do
    for l in $1 $2; do
      python tools/dev_lib_rank.py $l $r 256 >> $OUT/ab.txt 2>&1
    done
  done
done
cat $OUT/ab.txt
