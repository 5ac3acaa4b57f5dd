set -e
OUT=$GRAFT_REPO_ROOT/gpurun_out/r05_d
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
{
for cb in 0 2112 1600 2600 3100; do
  echo "LRF_PERSIST_CHUNK_BLOCKS=$cb"
  for r in 7,3,3 16,8,8 26,13,13; do LRF_PERSIST_CHUNK_BLOCKS=$cb python tools/dev_lib_persist.py liblrf_hip.so $r; done
done
} 2>&1 | grep -v amdgpu.ids > $OUT/chunks.txt
cat $OUT/chunks.txt
python -m pytest tests/test_configs_at_size.py -x -q -m gpu -k "persistent or full_size" > $OUT/t_persist.log 2>&1 || { tail -30 $OUT/t_persist.log; exit 1; }
tail -2 $OUT/t_persist.log
