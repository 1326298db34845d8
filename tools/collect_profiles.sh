#!/bin/bash
# Runs on the GPU box (gpurun): bench line, kernel trace + stats, and the separate PMC passes the
# roofline object quotes.  Outputs under gpurun_out/prof_final/; copy the summaries into profiles/.
set -eo pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${PROF_TAG:-prof_final}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-vfeat --no-e2e --no-pretrain --no-groups --no-bf16x3"
rocprofv3 --kernel-trace --stats -d $O/trace -o t --output-format csv -- $B > $O/trace.log 2>&1
python3 $R/tools/trace_summary.py $O/trace/t_kernel_trace.csv 13 > $O/trace_summary.txt 2>&1 || true
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/fetch -o f --output-format csv -- $B > $O/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/write -o w --output-format csv -- $B > $O/write.log 2>&1
python3 $R/tools/pmc_summary.py $O/fetch/f_counter_collection.csv $O/write/w_counter_collection.csv $O/pmc_traffic.json > $O/pmc_traffic.txt 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_VALU_MFMA_MOPS_F32 --kernel-trace -d $O/mfma -o m --output-format csv -- $B > $O/mfma.log 2>&1
python3 $R/tools/pmc_simple.py $O/mfma/m_counter_collection.csv gemm_f32 > $O/pmc_mfma.txt 2>&1
echo "profiles collected"
