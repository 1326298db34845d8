#!/bin/bash
# In-situ A/B of the convolution tile rules: the whole extractor (tools/vfeat_bench.py 128 10) with the shipped
# per-shape choice, with every implicit-GEMM convolution forced to one tile config (1x1 pinned to 64x64), and with
# every 1x1 forced to one config.  Isolated per-layer timings (tools/conv_tune.py, uniform random inputs) run at a
# different power point than the network's post-ReLU activations and do not transfer one to one.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for rep in 1 2; do
  echo "shipped rules: $(timeout -k 10 120 python tools/vfeat_bench.py 128 10 2>/dev/null | tail -1)"
  for c in 0 1 2; do
    echo "3x3 cfg $c (1x1 on 64x64): $(VQA_CONV_CFG=$c timeout -k 10 120 python tools/vfeat_bench.py 128 10 2>/dev/null | tail -1)"
  done
  for c in 3 16 20 21; do
    echo "1x1 cfg $c (3x3 by rule): $(VQA_GEMM_CFG=$c timeout -k 10 120 python tools/vfeat_bench.py 128 10 2>/dev/null | tail -1)"
  done
done
