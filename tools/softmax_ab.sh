#!/bin/bash
# Same-box A/B of the register-resident n-way softmax cross-entropy rows in the cfg-5 pre-training step.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
run() { echo "$1: $(env $2 timeout -k 10 100 python tools/pretrain_bench.py 10 2>/dev/null | tail -1)"; }
for rep in 1 2 3; do
  run "shipped (rows in registers)" "X=1"
  run "three-pass kernel          " "VQA_SOFTMAX_FAST=0"
done
