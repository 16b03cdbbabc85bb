#!/bin/bash
# round 4, call 23: PMC counters of the fp32 tiled kernels
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out/c23
export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"
for pass in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL" "FETCH_SIZE" "WRITE_SIZE"; do
  tag=$(echo $pass | tr ' ' '_' | cut -c1-30)
  (cd /tmp && rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $R/gpurun_out/c23/$tag -o p -- python $R/scratch/r3/enc_only.py > $R/gpurun_out/c23/$tag.log 2>&1)
done
python - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('gpurun_out/c23/*/p_counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if 'tiled' in k or 'mfma32_kernel<32, 32' in k:
            acc[k[:70]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, cs in acc.items():
    print(k)
    for c, v in sorted(cs.items()): print(f"   {c:32s} {sum(v)/len(v):16.1f}")
PY
find gpurun_out/c23 -name "*.csv" -delete
