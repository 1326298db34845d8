#!/bin/bash
# Same-box A/B of the pre-training step's switches (tools/pretrain_bench.py 10): LayerNorm register kernels, the
# per-image attention forward, the 64x64 tile for the 2560-row GEMMs.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
run() { echo "$1: $(env $2 timeout -k 10 100 python tools/pretrain_bench.py 10 2>/dev/null | tail -1)"; }
for rep in 1 2; do
  run "shipped" "X=1"
  run "generic LayerNorm kernels" "VQA_LN_FAST=0"
  run "per-query attention forward" "VQA_ATTN_FAST=3"
  run "128x64 tiles for the 2560-row GEMMs" "VQA_HOT_TALL_SMALL_CFG=-1"
done
