#!/bin/bash
# Everything under profiles/r03_* that comes from the GPU, in one gpurun call (about 6 GPU-minutes):
#   1. rocprofv3 --kernel-trace --stats of the DEFAULT bench command  -> r03_bench_kernel_stats.csv, r03_bench_under_rocprof.json
#   2. separate --pmc passes over the bench workload                   -> r03_pmc_summary.json (+ the hash of the kernel sources it was measured on)
#   3. the same two for three scene encodes and for the use_amp / fp32 training step
# Results land in gpurun_out/r3_profiles/; copy what is to be judged into profiles/.
export TMPDIR=/tmp
O=gpurun_out/r3_profiles
rm -rf $O; mkdir -p $O
# ---- 1. kernel trace of the default bench command
rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench_trace -o b -- python bench.py > $O/bench_under_rocprof.log 2>&1
grep "^{\"metric\"" $O/bench_under_rocprof.log | tail -1 > $O/r03_bench_under_rocprof.json
find $O/bench_trace -name "*kernel_stats.csv" -exec cp {} $O/r03_bench_kernel_stats.csv \;
rm -rf $O/bench_trace
# ---- 2. PMC passes over the bench workload (kernel-trace only next to --pmc)
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM"; do
  tag=$(echo $pass | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $O/pmc/$tag -o p -- python bench.py --steps 20 --warmup 3 --cpu-batches 0 --no-extras > $O/pmc_$tag.log 2>&1
done
python - <<'PY'
import csv, glob, collections, json, sys
sys.path.insert(0, '.')
import bench
out = {}
for f in glob.glob('gpurun_out/r3_profiles/pmc/*/p_counter_collection.csv'):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].replace('(anonymous namespace)::', '').split('(')[0].replace('void ', '')[:48]
        acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
    for k, cs in acc.items():
        for c, v in cs.items():
            out.setdefault(k, {})[c] = {"mean": sum(v) / len(v), "n": len(v)}
keep = {k: v for k, v in out.items() if any(s in k for s in ('mlp_fwd', 'volume_sample', 'color_sample', 'composite', 'planesweep', 'conv3d', 'convT', 'abn', 'dir_feature', 'gather_fused', 'conv2d'))}
keep["_csrc_sha16"] = bench.csrc_sha16()
keep["_command"] = "rocprofv3 --pmc <pass> --kernel-trace -- python bench.py --steps 20 --warmup 3 --cpu-batches 0 --no-extras (scratch/r3/profiles_r3.sh); FETCH_SIZE / WRITE_SIZE in KiB, raw"
json.dump(keep, open('gpurun_out/r3_profiles/r03_pmc_summary.json', 'w'), indent=1)
for k, v in keep.items():
    if isinstance(v, dict): print(k, {c: round(x['mean'], 1) for c, x in v.items()})
PY
rm -rf $O/pmc
# ---- 3. encoder and training step
bash scratch/r3/prof_enc.sh > $O/prof_enc.out 2>&1;  cp gpurun_out/r3_enc_kernel_summary.txt $O/r03_encode_kernel_summary.txt; cp gpurun_out/r3_enc_kernel_stats.csv $O/r03_encode_kernel_stats.csv
bash scratch/r3/pmc_enc.sh > $O/r03_pmc_enc_table.txt 2>&1; cp gpurun_out/r3_pmc_enc/summary.json $O/r03_pmc_enc_summary.json
bash scratch/r3/prof_train.sh > $O/prof_train.out 2>&1
cp gpurun_out/r3_train_kernel_summary_fp32.txt $O/r03_train_kernel_summary_fp32.txt; cp gpurun_out/r3_train_kernel_summary_amp.txt $O/r03_train_kernel_summary_use_amp.txt
cp gpurun_out/r3_train_kernel_stats_fp32.csv $O/r03_train_kernel_stats_fp32.csv; cp gpurun_out/r3_train_kernel_stats_amp.csv $O/r03_train_kernel_stats_use_amp.csv
bash scratch/r3/pmc_train.sh amp > /dev/null 2>&1; cp gpurun_out/r3_pmc_train/table.txt $O/r03_pmc_train_use_amp_table.txt
# ---- 4. the un-profiled bench line of the same tree
python bench.py > $O/r03_bench.json 2> $O/r03_bench.err
ls -la $O
