# round 5: k_any_eig's Gram-Schmidt with batched dot products: parity of the any-shape paths, stage times, branch times
set -e
OUT=$GRAFT_REPO_ROOT/gpurun_out/r05_ac
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_anyshape.py tests/test_svd_any.py tests/test_rgbspace.py tests/test_qmf_kwargs.py -x -q -m gpu > $OUT/t.log 2>&1 || { tail -30 $OUT/t.log; exit 1; }
tail -2 $OUT/t.log
python tools/bench_anyshape.py 256 20 > $OUT/any.txt 2>&1; cat $OUT/any.txt
python bench.py --config svd --steps 5 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('svd', d['ms_per_step'])"
