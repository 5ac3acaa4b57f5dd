#!/bin/bash
# ablation timing of k_any_tridiag_blk: variants built with -DLRF_BLK_ABL=1 (no loads in the product / update passes),
# 3 (no panel update), 6 (panel update without its stores); eigen-solver stopped after stage 1
export LRF_DEBUG_INIT_SWEEPS=1
echo base; timeout -k 10 100 python tools/dev_any_init_times.py < /dev/null || exit 1
for v in abl1 abl3 abl6; do
  cp tools/probe/_variants/lib_$v.so lrf_amd/liblrf_hip.so && echo $v && timeout -k 10 100 python tools/dev_any_init_times.py < /dev/null || exit 1
done
