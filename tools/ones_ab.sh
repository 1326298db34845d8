#!/bin/bash
# Same-box A/B of the constant-1 input column (GRU bias gradients out of the x-part weight-gradient GEMM) against a
# library built with the two colsum passes: build_ab/libvqahot_noones.so (old fusion_model / pretrain_model objects).
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
V=$R/build_ab/libvqahot_noones.so
line() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.1f samples/s  %.3f ms/step' % (d['value'], d['ms_per_step']))"; }
for rep in 1 2 3; do
  echo "step  shipped : $(timeout -k 10 200 python bench.py --steps 200 --warmup 20 2>/dev/null | line)"
  echo "step  colsums : $(timeout -k 10 200 python tools/bench_with_lib.py $V --steps 200 --warmup 20 2>/dev/null | line)"
  echo "cfg5  shipped : $(timeout -k 10 100 python tools/pretrain_bench.py 10 2>/dev/null | tail -1)"
  echo "cfg5  colsums : $(VQA_HOT_LIB=$V timeout -k 10 100 python tools/pretrain_bench.py 10 2>/dev/null | tail -1)"
done
