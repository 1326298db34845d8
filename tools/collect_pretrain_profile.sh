#!/bin/bash
# GPU box: per-kernel summary of the cfg-5 pre-training step (tools/pretrain_bench.py) under rocprofv3.
set -eo pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/prof_pretrain
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python3 $R/tools/pretrain_bench.py 10 > $O/pretrain_bench.txt 2>&1
rocprofv3 --kernel-trace --stats -d $O/trace -o t --output-format csv -- python3 $R/tools/pretrain_bench.py 10 > $O/trace.log 2>&1
python3 $R/tools/trace_summary.py $O/trace/t_kernel_trace.csv 13 > $O/trace_summary.txt 2>&1 || true
rm -f $O/trace/*.db
tail -1 $O/pretrain_bench.txt; head -45 $O/trace_summary.txt; tail -1 $O/trace_summary.txt
