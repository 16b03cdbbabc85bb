#!/bin/bash
# Round 6: the evidence of the FINAL tree in one gpurun call (about 7 GPU-minutes).  Everything lands in gpurun_out/r6_final/; the files named r06_*
# are copied to profiles/ afterwards.
#   1. rocprofv3 --kernel-trace --stats of the DEFAULT bench command        -> r06_bench_kernel_stats.csv, r06_bench_under_rocprof.json
#   2. separate --pmc passes over the bench workload (fp32 headline, fp16x3) -> r06_pmc_summary.json (+ hash of the kernel sources it was measured on)
#   3. kernel statistics of the encode and of the training step (fp32, use_amp)  -> r06_{encode,train_fp32,train_use_amp}_kernel_stats.csv
#   4. --pmc passes over the encode and the use_amp training step            -> r06_pmc_enc_table.txt, r06_pmc_train_use_amp_table.txt
#   5. the un-profiled bench line                                            -> r06_bench.json
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r6_final
rm -rf $O; mkdir -p $O
R="$GRAFT_REPO_ROOT"
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/bench_trace -o b -- python $R/bench.py > $R/$O/bench_under_rocprof.log 2>&1)
grep "^{\"metric\"" $O/bench_under_rocprof.log | tail -1 > $O/r06_bench_under_rocprof.json
find $O/bench_trace -name "*kernel_stats.csv" -exec cp {} $O/r06_bench_kernel_stats.csv \;
rm -rf $O/bench_trace
PASSES=("FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F32")
for pass in "${PASSES[@]}"; do
  tag=$(echo $pass | tr ' ' '_' | cut -c1-40)
  (cd /tmp && rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $R/$O/pmc/$tag -o p -- python $R/bench.py --steps 20 --warmup 3 --cpu-batches 0 --no-extras > $R/$O/pmc_$tag.log 2>&1)
  (cd /tmp && rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $R/$O/pmc/h3_$tag -o p -- python $R/bench.py --steps 20 --warmup 3 --cpu-batches 0 --no-extras --mlp-precision fp16x3 > $R/$O/pmc_h3_$tag.log 2>&1)
  (cd /tmp && rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $R/$O/pmc_enc/$tag -o p -- python $R/scratch/r6/enc_only.py > $R/$O/pmc_enc_$tag.log 2>&1)
  (cd /tmp && rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $R/$O/pmc_train/$tag -o p -- python $R/scratch/r6/train_prof.py amp 3 > $R/$O/pmc_train_$tag.log 2>&1)
done
python - <<'PY'
import csv, glob, collections, json, sys
sys.path.insert(0, '.')
import bench
O = 'gpurun_out/r6_final'
def collect(pattern, maxlen=64):
    out = {}
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(pattern):
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('_ZN12_GLOBAL__N_120mlp_fwd_f16x3_kernelI', 'mlp_fwd_f16x3_kernel<').replace('_ZN12_GLOBAL__N_1', '').split('(')[0].replace('void ', '')[:maxlen]
            acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
    for k, cs in acc.items():
        for c, v in cs.items():
            out.setdefault(k, {})[c] = {"mean": sum(v) / len(v), "n": len(v)}
    return out
b = collect(O + '/pmc/*/p_counter_collection.csv', 48)
keep = {k: v for k, v in b.items() if any(s in k for s in ('mlp_fwd', 'volume_sample', 'color_sample', 'composite', 'planesweep', 'conv3d', 'convT', 'abn', 'dir_feature', 'gather_fused', 'conv2d', 'guard', 'conv_f16x3'))}
keep["_csrc_sha16"] = bench.csrc_sha16()
keep["_command"] = "rocprofv3 --pmc <pass> --kernel-trace -- python bench.py --steps 20 --warmup 3 --cpu-batches 0 --no-extras [--mlp-precision fp16x3] (scratch/r6/final_profiles.sh); FETCH_SIZE / WRITE_SIZE in KiB, raw"
json.dump(keep, open(O + '/r06_pmc_summary.json', 'w'), indent=1)
def table(out, path, title, min_us=15.0):
    rows = []
    for k, v in out.items():
        g = v.get('GRBM_GUI_ACTIVE', {}).get('mean', 0)
        us = g / 8 / 2100.0
        if us < min_us: continue
        busy = (v.get('SQ_VALU_MFMA_BUSY_CYCLES', {}).get('mean', 0) / 1024.0) / (g / 8.0) if g else 0
        lds = v.get('SQ_LDS_BANK_CONFLICT', {}).get('mean', 0) / max(v.get('SQ_LDS_IDX_ACTIVE', {}).get('mean', 0), 1)
        rows.append((us, k, busy, 2 * v.get('FETCH_SIZE', {}).get('mean', 0) / 1024, v.get('WRITE_SIZE', {}).get('mean', 0) / 1024, lds,
                     v.get('SQ_INSTS_VALU', {}).get('mean', 0), v.get('SQ_INSTS_LDS', {}).get('mean', 0), v.get('SQ_INSTS_VMEM', {}).get('mean', 0)))
    with open(path, 'w') as f:
        f.write(title + "\nmeans per launch; us = GRBM_GUI_ACTIVE / 8 at 2.1 GHz; matrix pipes busy = (SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs) / (GRBM_GUI_ACTIVE / 8 XCDs);\n"
                "fetch = 2 x FETCH_SIZE (the guide's gfx950 correction), write = WRITE_SIZE, MB; instruction counts per launch; kernels above %g us\n" % min_us)
        f.write(f"{'kernel':64s} {'us':>8s} {'mfma busy':>10s} {'fetch MB':>10s} {'write MB':>10s} {'LDS confl/act':>14s} {'VALU':>12s} {'LDS':>10s} {'VMEM':>10s}\n")
        for us, k, busy, fe, wr, lds, iv, il, im in sorted(rows, reverse=True):
            f.write(f"{k:64s} {us:8.1f} {100*busy:9.1f}% {fe:10.1f} {wr:10.1f} {lds:14.3f} {iv:12.0f} {il:10.0f} {im:10.0f}\n")
    print(open(path).read())
table(collect(O + '/pmc_enc/*/p_counter_collection.csv'), O + '/r06_pmc_enc_table.txt', "rocprofv3 --pmc passes over three scene encodes (scratch/r6/enc_only.py; library defaults: guarded fp16x3 conv0)", 10.0)
table(collect(O + '/pmc_train/*/p_counter_collection.csv'), O + '/r06_pmc_train_use_amp_table.txt', "rocprofv3 --pmc passes over the use_amp training step (scratch/r6/train_prof.py amp 3)")
table(b, O + '/r06_pmc_bench_table.txt', "rocprofv3 --pmc passes over the bench workload (fp32 headline and --mlp-precision fp16x3)", 2.0)
PY
rm -rf $O/pmc $O/pmc_enc $O/pmc_train
for m in "fp32 4 train_fp32" "amp 4 train_use_amp"; do set -- $m
  (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/tr_$3 -o t -- python $R/scratch/r6/train_prof.py $1 $2 > $R/$O/$3.log 2>&1)
  find $O/tr_$3 -name "*kernel_stats.csv" -exec cp {} $O/r06_$3_kernel_stats.csv \;
  grep "train step\|again" $O/$3.log > $O/r06_$3_step_ms_under_rocprof.txt
  rm -rf $O/tr_$3
done
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/tr_enc -o t -- python $R/scratch/r6/enc_only.py > $R/$O/enc.log 2>&1)
find $O/tr_enc -name "*kernel_stats.csv" -exec cp {} $O/r06_encode_kernel_stats.csv \;
rm -rf $O/tr_enc
for m in fp32 amp; do python scratch/r6/train_prof.py $m 10 2>&1 | grep -v amdgpu.ids >> $O/r06_train_step_ms.txt; done
cp $O/r06_pmc_summary.json profiles/r06_pmc_summary.json      # (on the box) so that the bench line below reports the traffic measured on THESE kernel sources
python bench.py > $O/r06_bench.json 2> $O/r06_bench.err
tail -c 600 $O/r06_bench.json; cat $O/r06_train_step_ms.txt
ls -la $O
