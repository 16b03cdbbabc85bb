#!/bin/bash
# PMC passes over the encoder kernels (scratch/enc_only.py); summary -> gpurun_out/pmc_enc/summary.txt
export TMPDIR=/tmp
mkdir -p gpurun_out/pmc_enc
i=0
for pass in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INST_CYCLES_VMEM SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM GRBM_GUI_ACTIVE" "SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_SMEM" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  rm -rf gpurun_out/pmc_enc/p$i
  rocprofv3 --pmc $pass --kernel-trace --output-format csv -d gpurun_out/pmc_enc/p$i -o p -- python scratch/enc_only.py > gpurun_out/pmc_enc/p$i.log 2>&1
  tail -1 gpurun_out/pmc_enc/p$i.log | cut -c1-200
done
python - <<'PY' > gpurun_out/pmc_enc/summary.txt
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('gpurun_out/pmc_enc/*/p_counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        n = r['Kernel_Name']
        for key in ('conv3d_k3s1_c8_mfma_kernel', 'planesweep_kernel', 'conv3d_k3s1_tiled_kernel<16, 16', 'convT3d_k3s2_kernel<16, 8', 'conv_mfma_kernel'):
            if key in n:
                acc[key][r['Counter_Name']].append(float(r['Counter_Value']))
for k, cs in acc.items():
    print("==", k)
    for c, v in sorted(cs.items()):
        print(f"  {c:28s} {sum(v)/len(v):16.1f}  n={len(v)}")
PY
cat gpurun_out/pmc_enc/summary.txt
