#!/bin/bash
# separate rocprofv3 PMC passes over three scene encodes (kernel-trace only next to --pmc, as the guide prescribes) -> gpurun_out/r3_pmc_enc/summary.json
export TMPDIR=/tmp
mkdir -p gpurun_out/r3_pmc_enc
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY"; do
  tag=$(echo $pass | tr ' ' '_' | cut -c1-40)
  rm -rf gpurun_out/r3_pmc_enc/$tag
  rocprofv3 --pmc $pass --kernel-trace --output-format csv -d gpurun_out/r3_pmc_enc/$tag -o p -- python scratch/r3/enc_only.py > gpurun_out/r3_pmc_enc/$tag.log 2>&1
done
python - <<'PY'
import csv, glob, collections, json
out = {}
for f in glob.glob('gpurun_out/r3_pmc_enc/*/p_counter_collection.csv'):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].replace('(anonymous namespace)::', '').split('(')[0].replace('void ', '')[:56]
        acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
    for k, cs in acc.items():
        for c, v in cs.items():
            out.setdefault(k, {})[c] = {"mean": sum(v) / len(v), "n": len(v)}
keep = {k: v for k, v in out.items() if any(s in k for s in ('planesweep', 'conv3d', 'convT', 'conv2d', 'abn_'))}
json.dump(keep, open('gpurun_out/r3_pmc_enc/summary.json', 'w'), indent=1)
for k, v in keep.items():
    if 'SQ_LDS_IDX_ACTIVE' in v and v['SQ_LDS_IDX_ACTIVE']['mean'] > 0:
        print(f"{k:58s} lds conflict rate {v['SQ_LDS_BANK_CONFLICT']['mean'] / v['SQ_LDS_IDX_ACTIVE']['mean']:.3f}  fetch {2*v.get('FETCH_SIZE',{}).get('mean',0)/1024:.1f} MB write {v.get('WRITE_SIZE',{}).get('mean',0)/1024:.1f} MB")
PY
find gpurun_out/r3_pmc_enc -name "*.csv" -size +2M -delete
