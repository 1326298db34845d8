#!/bin/bash
# Is one recurrence over the 2 x 2560 captions of both categories faster than two over 2560?  (tools/gru_tune.py, full-length rows)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for b in 2560 5120; do
  echo "== rows $b, T 10, library defaults (-1) and forced tile configs"
  GRU_T=10 GRU_B=$b timeout -k 10 120 python tools/gru_tune.py -1,12,13,20,21,9,17 2>&1 | tail -12
done
