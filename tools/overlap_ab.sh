F="--steps 30 --warmup 5 --no-cpu-baseline --no-vfeat --no-e2e --no-pretrain --no-groups --no-bf16x3"
for cfg in "0 0 2" "1 0 2" "1 256 2" "1 512 2" "1 256 1" "1 128 2" "1 384 2"; do
  set -- $cfg
  echo "OVERLAP=$1 SIDE_BLOCKS=$2 CHAINS=$3" >> gpurun_out/r4_overlap.txt
  VQA_HOT_OVERLAP=$1 VQA_HOT_SIDE_BLOCKS=$2 VQA_HOT_GRU_CHAINS=$3 timeout -k 10 120 python bench.py $F 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(' ms_per_step', round(d['ms_per_step'],4), 'value', round(d['value'],1), 'kernel_ms', d['roofline'].get('kernel_ms'))" >> gpurun_out/r4_overlap.txt 2>&1
done
cat gpurun_out/r4_overlap.txt
