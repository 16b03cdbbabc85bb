#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/pmc2
i=0
for pass in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_VALU_MFMA_BUSY_CYCLES" "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_MFMA SQ_WAVES SQ_INSTS_SMEM SQ_ACTIVE_INST_EXP_GDS SQ_WAIT_INST_ANY SQ_INSTS_FLAT SQ_LDS_ADDR_CONFLICT"; do
  i=$((i+1))
  rm -rf gpurun_out/pmc2/p$i
  rocprofv3 --pmc $pass --kernel-trace --output-format csv -d gpurun_out/pmc2/p$i -o p -- python scratch/mlp_stagger.py > gpurun_out/pmc2/p$i.log 2>&1
done
python - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob('gpurun_out/pmc2/*/p_counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if 'mlp_fwd_pipe' in r['Kernel_Name']:
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
for c, v in sorted(acc.items()):
    print(f"{c:32s} mean {sum(v)/len(v):16.1f}  n={len(v)}")
PY
