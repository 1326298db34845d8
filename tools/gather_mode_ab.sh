#!/bin/bash
# GPU box: the train step under the three ways of producing V_ft (VQA_HOT_GATHER = side | fused | inline), two rounds each.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for round in 1 2; do
  for mode in side fused inline; do
    VQA_HOT_GATHER=$mode timeout -k 10 120 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-vfeat --no-e2e 2>/dev/null \
      | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$mode', round(d['ms_per_step'],4), 'ms/step  fwd gemm', round(d['roofline']['kernel_ms']*1e3,1), 'us', flush=True)"
  done
done
