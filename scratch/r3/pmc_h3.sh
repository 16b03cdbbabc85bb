#!/bin/bash
# PMC passes over the fp16x3 / bf16x6 / bf16 MLP kernels alone (scratch/r3/h3_ab.py), kernel-trace only next to --pmc -> gpurun_out/r3b/pmc_h3.txt
export TMPDIR=/tmp
O=gpurun_out/r3b/pmc_h3
rm -rf $O; mkdir -p $O
for pass in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE" "SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_SALU SQ_INSTS_VMEM" "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum"; do
  tag=$(echo $pass | tr ' ' '_' | cut -c1-30)
  rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $O/$tag -o p -- python scratch/r3/h3_ab.py fp16x3 bf16x6 > $O/$tag.log 2>&1
done
python - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('gpurun_out/r3b/pmc_h3/*/p_counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].replace('(anonymous namespace)::', '').split('(')[0].replace('void ', '')[:40]
        acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
with open('gpurun_out/r3b/pmc_h3.txt', 'w') as o:
    for k, cs in acc.items():
        if 'mlp_fwd' not in k: continue
        o.write(k + '\n')
        for c, v in sorted(cs.items()):
            o.write(f"   {c:32s} {sum(v)/len(v):16.1f}  (n={len(v)})\n")
print(open('gpurun_out/r3b/pmc_h3.txt').read())
PY
find $O -name "*.csv" -size +1M -delete
