#!/bin/bash
# two rocprofv3 --pmc passes over the prototype at the two encode shapes (grid 512)
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; R="$GRAFT_REPO_ROOT"; O=gpurun_out/r5_prep; mkdir -p $O
i=0
for pass in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F32" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM"; do
  i=$((i+1))
  (cd /tmp && R5_GRIDS=512 timeout 60 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $R/$O/pmc$i -o p -- python $R/scratch/r5_prep/check.py > $R/$O/pmc$i.log 2>&1)
done
python - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('gpurun_out/r5_prep/pmc*/**/p_counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'tiled' in r['Kernel_Name']:
            acc[r['Kernel_Name'][:70]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, cs in acc.items():
    print(k)
    for c, v in sorted(cs.items()):
        print(f"    {c:32s} {sum(v)/len(v):16.0f}  (n={len(v)})")
PY
rm -rf $O/pmc1 $O/pmc2
