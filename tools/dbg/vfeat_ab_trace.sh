#!/bin/bash
# which kernels does the extractor launch under VQA_CONV_CFG=0 and under VQA_GEMM_CFG=3 (expected: the same)?
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/vfeat_ab
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export VQA_CONV_CFG=0
rocprofv3 --kernel-trace --stats -d $O/a -o t --output-format csv -- python3 $R/tools/vfeat_bench.py 128 3 > $O/a.log 2>&1
unset VQA_CONV_CFG
export VQA_GEMM_CFG=3
rocprofv3 --kernel-trace --stats -d $O/b -o t --output-format csv -- python3 $R/tools/vfeat_bench.py 128 3 > $O/b.log 2>&1
python3 $R/tools/trace_summary.py $O/a/t_kernel_trace.csv 4 > $O/a_summary.txt
python3 $R/tools/trace_summary.py $O/b/t_kernel_trace.csv 4 > $O/b_summary.txt
rm -rf $O/a/*.db $O/b/*.db $O/a/t_kernel_trace.csv $O/b/t_kernel_trace.csv
head -16 $O/a_summary.txt; echo ----; head -16 $O/b_summary.txt
