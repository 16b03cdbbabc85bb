#!/bin/bash
# PMC passes over the split-bf16 MLP kernel (scratch/split_modes.py)
export TMPDIR=/tmp
mkdir -p gpurun_out/pmc3
i=0
for pass in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_VALU_MFMA_BUSY_CYCLES" "SQ_INSTS_MFMA SQ_WAVES SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES"; do
  i=$((i+1))
  rm -rf gpurun_out/pmc3/p$i
  rocprofv3 --pmc $pass --kernel-trace --output-format csv -d gpurun_out/pmc3/p$i -o p -- python scratch/split_modes.py > gpurun_out/pmc3/p$i.log 2>&1
done
python - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('gpurun_out/pmc3/*/p_counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        n = r['Kernel_Name']
        for key in ('split_kernelILi3', 'split_kernelILi2', 'mlp_fwd_pipe', 'mlp_fwd_bf16_kernel'):
            if key in n:
                acc[key][r['Counter_Name']].append(float(r['Counter_Value']))
for k, cs in acc.items():
    print("==", k)
    for c, v in sorted(cs.items()):
        print(f"  {c:32s} {sum(v)/len(v):16.1f}")
PY
