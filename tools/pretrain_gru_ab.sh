#!/bin/bash
# In-situ sweep of the recurrence's tile choice by live-row count inside the pre-training step (tools/pretrain_bench.py):
# mid = 512 < rows < 2048, tall = rows >= 2048, narrow = the H-wide candidate kernel of tall batches.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
run() { echo "$1: $(env $1 timeout -k 10 100 python tools/pretrain_bench.py 10 2>/dev/null | tail -1 | cut -c1-24)"; }
run "X=shipped"
for c in 13 12 7 17 18; do run "VQA_HOT_GRU_MID_FWD=$c"; done
for c in 13 7 11 9 18; do run "VQA_HOT_GRU_MID_BWD=$c"; done
for c in 10 13 9; do run "VQA_HOT_GRU_TALL_FWD=$c"; done
for c in 7 9 11 17; do run "VQA_HOT_GRU_TALL_BWD=$c"; done
for c in -1 13 7; do run "VQA_HOT_GRU_NARROW_CFG=$c"; done
run "X=shipped"
