#!/bin/bash
# Same-box A/B: blank-fill captions of both categories encoded as ONE recurrence over 2 x 2560 rows (this tree) against
# the previous commit's two recurrences over 2560 (a copy of that tree with its own library under build_ab/old_tree).
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
run() { echo "$1: $(cd $3 && env $2 timeout -k 10 100 python tools/pretrain_bench.py 10 2>&1 | tail -1)"; }
for rep in 1 2 3; do
  run "one recurrence, 5120 rows   " "X=1" $R
  run "two recurrences, 2560 rows  " "X=1" $R/build_ab/old_tree
  run "one; candidate on gate tiles" "VQA_HOT_GRU_NARROW_CFG=-1" $R
  run "one; backward tall cfg 12   " "VQA_HOT_GRU_TALL_BWD=12" $R
done
