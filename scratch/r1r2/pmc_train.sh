#!/bin/bash
# separate rocprofv3 PMC passes over the training step (kernel-trace only) -> gpurun_out/pmc_train/summary.json
export TMPDIR=/tmp
mkdir -p gpurun_out/pmc_train
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_ANY"; do
  tag=$(echo $pass | tr ' ' '_' | cut -c1-40)
  rm -rf gpurun_out/pmc_train/$tag
  rocprofv3 --pmc $pass --kernel-trace --output-format csv -d gpurun_out/pmc_train/$tag -o p -- python scratch/train_prof.py > gpurun_out/pmc_train/$tag.log 2>&1
done
python - <<'PY'
import csv, glob, collections, json
out = {}
for f in glob.glob('gpurun_out/pmc_train/*/p_counter_collection.csv'):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '')
        k = k.split('(')[0][:64]
        acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
    for k, cs in acc.items():
        for c, v in cs.items():
            out.setdefault(k, {})[c] = {"mean": sum(v) / len(v), "n": len(v)}
keep = {k: v for k, v in out.items() if any(s in k for s in ('wgrad', 'planesweep', 'conv3d_k3s1_c8', 'convT3d', 'conv3d_k3s1_tiled', 'mlp_', 'abn_bwd', 'partial_sum_multi', 'pack_weights'))}
json.dump(keep, open('gpurun_out/pmc_train/summary.json', 'w'), indent=1)
for k, v in sorted(keep.items()):
    print(k, {c: round(x['mean'], 1) for c, x in v.items()})
PY
