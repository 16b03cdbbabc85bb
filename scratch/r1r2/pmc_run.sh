#!/bin/bash
# separate rocprofv3 PMC passes over the bench workload (kernel-trace only, as the guide prescribes)
export TMPDIR=/tmp
mkdir -p gpurun_out/pmc
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM"; do
  tag=$(echo $pass | tr ' ' '_' | cut -c1-40)
  rm -rf gpurun_out/pmc/$tag
  rocprofv3 --pmc $pass --kernel-trace --output-format csv -d gpurun_out/pmc/$tag -o p -- python bench.py --steps 20 --warmup 3 --cpu-batches 0 --no-extras > gpurun_out/pmc/$tag.log 2>&1
  ls gpurun_out/pmc/$tag | head -3
done
python - <<'PY'
import csv, glob, collections, json
out = {}
for f in glob.glob('gpurun_out/pmc/*/p_counter_collection.csv'):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].replace('(anonymous namespace)::', '').split('(')[0].replace('void ', '')[:48]
        acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
    for k, cs in acc.items():
        for c, v in cs.items():
            out.setdefault(k, {})[c] = {"mean": sum(v) / len(v), "n": len(v)}
keep = {k: v for k, v in out.items() if any(s in k for s in ('mlp_fwd', 'volume_sample', 'color_sample', 'composite', 'planesweep', 'conv3d', 'convT', 'abn', 'dir_feature', 'gather_fused', 'conv2d'))}
json.dump(keep, open('gpurun_out/pmc/summary.json', 'w'), indent=1)
for k, v in keep.items():
    print(k, {c: round(x['mean'], 1) for c, x in v.items()})
PY
