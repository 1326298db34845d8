#!/bin/bash
# GPU box: PMC passes over the persistent forward recurrence in its XCD-local form (VQA_GRU_PERSIST_XCD=2) next to the
# per-step kernels of the same process (PERSIST=1 tools/gru_tune.py runs both).
set -eo pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/pmc_gru_persistent
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export VQA_GRU_PERSIST_XCD=2 PERSIST=1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_VALU_MFMA_MOPS_F32 --kernel-trace -d $O/mfma -o m --output-format csv -- python3 $R/tools/gru_tune.py > $O/mfma.log 2>&1
python3 $R/tools/pmc_simple.py $O/mfma/m_counter_collection.csv > $O/pmc_mfma.txt 2>&1 || true
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES --kernel-trace -d $O/lds -o l --output-format csv -- python3 $R/tools/gru_tune.py > $O/lds.log 2>&1
python3 $R/tools/pmc_simple.py $O/lds/l_counter_collection.csv > $O/pmc_lds.txt 2>&1 || true
rm -f $O/*/*.db
head -40 $O/pmc_mfma.txt; head -60 $O/pmc_lds.txt
