set -eo pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4ws_lds
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-vfeat --no-e2e --no-pretrain --no-groups --no-bf16x3"
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS --kernel-trace -d $O/lds -o l --output-format csv -- $B > $O/lds.log 2>&1
python3 $R/tools/pmc_simple.py $O/lds/l_counter_collection.csv gru_ws > $O/pmc_lds.txt 2>&1
cat $O/pmc_lds.txt
