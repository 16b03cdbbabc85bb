#!/bin/bash
# Second half of round 3 (after the fp16x3 MLP kernel): the bench-side profiles of the FINAL tree in one gpurun call (about 4 GPU-minutes):
#   1. rocprofv3 --kernel-trace --stats of the DEFAULT bench command   -> r03_bench_kernel_stats.csv, r03_bench_under_rocprof.json
#   2. separate --pmc passes over the bench workload                    -> r03_pmc_summary.json (+ hash of the kernel sources it was measured on)
#   3. the un-profiled bench line                                       -> r03_bench.json ; bench.py --mlp-precision fp16x3 -> r03b_bench_fp16x3.json
# The encode / training-step profiles of scratch/r3/profiles_r3.sh (sections 3) are unchanged by the MLP work: no kernel of theirs was touched.
export TMPDIR=/tmp
O=gpurun_out/r3b_profiles
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench_trace -o b -- python bench.py > $O/bench_under_rocprof.log 2>&1
grep "^{\"metric\"" $O/bench_under_rocprof.log | tail -1 > $O/r03_bench_under_rocprof.json
find $O/bench_trace -name "*kernel_stats.csv" -exec cp {} $O/r03_bench_kernel_stats.csv \;
rm -rf $O/bench_trace
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM"; do
  tag=$(echo $pass | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $O/pmc/$tag -o p -- python bench.py --steps 20 --warmup 3 --cpu-batches 0 --no-extras > $O/pmc_$tag.log 2>&1
  rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $O/pmc/h3_$tag -o p -- python bench.py --steps 20 --warmup 3 --cpu-batches 0 --no-extras --mlp-precision fp16x3 > $O/pmc_h3_$tag.log 2>&1
done
python - <<'PY'
import csv, glob, collections, json, sys
sys.path.insert(0, '.')
import bench
out = {}
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('gpurun_out/r3b_profiles/pmc/*/p_counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('_ZN12_GLOBAL__N_120mlp_fwd_f16x3_kernelI', 'mlp_fwd_f16x3_kernel<').split('(')[0].replace('void ', '')[:48]
        acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
for k, cs in acc.items():
    for c, v in cs.items():
        out.setdefault(k, {})[c] = {"mean": sum(v) / len(v), "n": len(v)}
keep = {k: v for k, v in out.items() if any(s in k for s in ('mlp_fwd', 'volume_sample', 'color_sample', 'composite', 'planesweep', 'conv3d', 'convT', 'abn', 'dir_feature', 'gather_fused', 'conv2d'))}
keep["_csrc_sha16"] = bench.csrc_sha16()
keep["_command"] = "rocprofv3 --pmc <pass> --kernel-trace -- python bench.py --steps 20 --warmup 3 --cpu-batches 0 --no-extras [--mlp-precision fp16x3] (scratch/r3/profiles_r3b.sh); FETCH_SIZE / WRITE_SIZE in KiB, raw"
json.dump(keep, open('gpurun_out/r3b_profiles/r03_pmc_summary.json', 'w'), indent=1)
for k, v in keep.items():
    if isinstance(v, dict): print(k, {c: round(x['mean'], 1) for c, x in v.items()})
PY
rm -rf $O/pmc
cp $O/r03_pmc_summary.json profiles/r03_pmc_summary.json      # (on the box) so that the bench line below reports the traffic measured on THESE kernel sources
python bench.py > $O/r03_bench.json 2> $O/r03_bench.err
python bench.py --mlp-precision fp16x3 --no-extras > $O/r03b_bench_fp16x3.json 2> $O/r03b_bench_fp16x3.err
ls -la $O
