#!/bin/bash
# Round 4, after the plane sweep moved to csrc/planesweep.hip (no packed fp32 instructions): the evidence bench.py depends on, re-measured on the new
# kernel sources in one gpurun call (about 5 GPU-minutes).  Everything lands in gpurun_out/r4_refresh/; the files named r04_* are copied to profiles/.
#   1. --pmc passes over the bench workload (fp32 headline, fp16x3)        -> r04_pmc_summary.json (hash of THESE kernel sources), r04_pmc_bench_table.txt
#   2. the un-profiled bench line                                          -> r04_bench.json
#   3. kernel statistics of three scene encodes                            -> r04_encode_kernel_stats.csv
#   4. rocprofv3 --kernel-trace --stats of the DEFAULT bench command       -> r04_bench_kernel_stats.csv, r04_bench_under_rocprof.json   (if time is left)
# Not re-measured (the only kernel that changed is planesweep_kernel): r04_pmc_enc_table.txt, r04_pmc_train_use_amp_table.txt, r04_train_*.
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r4_refresh
rm -rf $O; mkdir -p $O
R="$GRAFT_REPO_ROOT"
T0=$(date +%s)
# (the LDS / instruction-count pass of final_profiles.sh is left out: bench.py reads FETCH_SIZE, WRITE_SIZE and the matrix-pipe busy cycles only)
PASSES=("FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY")
for pass in "${PASSES[@]}"; do
  tag=$(echo $pass | tr ' ' '_' | cut -c1-40)
  (cd /tmp && timeout 60 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $R/$O/pmc/$tag -o p -- python $R/bench.py --steps 20 --warmup 3 --cpu-batches 0 --no-extras > $R/$O/pmc_$tag.log 2>&1)
  (cd /tmp && timeout 60 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $R/$O/pmc/h3_$tag -o p -- python $R/bench.py --steps 20 --warmup 3 --cpu-batches 0 --no-extras --mlp-precision fp16x3 > $R/$O/pmc_h3_$tag.log 2>&1)
  echo "pass $tag done at $(( $(date +%s) - T0 )) s" >> $O/timeline.txt
done
python - <<'PY'
import csv, glob, collections, json, sys
sys.path.insert(0, '.')
import bench
O = 'gpurun_out/r4_refresh'
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
keep = {k: v for k, v in b.items() if any(s in k for s in ('mlp_fwd', 'volume_sample', 'color_sample', 'composite', 'planesweep', 'conv3d', 'convT', 'abn', 'dir_feature', 'gather_fused', 'conv2d', 'guard'))}
keep["_csrc_sha16"] = bench.csrc_sha16()
keep["_command"] = "rocprofv3 --pmc <pass> --kernel-trace -- python bench.py --steps 20 --warmup 3 --cpu-batches 0 --no-extras [--mlp-precision fp16x3] (scratch/r4/refresh_evidence.sh); FETCH_SIZE / WRITE_SIZE in KiB, raw"
json.dump(keep, open(O + '/r04_pmc_summary.json', 'w'), indent=1)
rows = []
for k, v in b.items():
    g = v.get('GRBM_GUI_ACTIVE', {}).get('mean', 0)
    us = g / 8 / 2100.0
    if us < 2.0: continue
    busy = (v.get('SQ_VALU_MFMA_BUSY_CYCLES', {}).get('mean', 0) / 1024.0) / (g / 8.0) if g else 0
    lds = v.get('SQ_LDS_BANK_CONFLICT', {}).get('mean', 0) / max(v.get('SQ_LDS_IDX_ACTIVE', {}).get('mean', 0), 1)
    rows.append((us, k, busy, 2 * v.get('FETCH_SIZE', {}).get('mean', 0) / 1024, v.get('WRITE_SIZE', {}).get('mean', 0) / 1024, lds,
                 v.get('SQ_INSTS_VALU', {}).get('mean', 0), v.get('SQ_INSTS_LDS', {}).get('mean', 0), v.get('SQ_INSTS_VMEM', {}).get('mean', 0)))
with open(O + '/r04_pmc_bench_table.txt', 'w') as f:
    f.write("rocprofv3 --pmc passes over the bench workload (fp32 headline and --mlp-precision fp16x3), tree with csrc/planesweep.hip (scratch/r4/refresh_evidence.sh)\n"
            "means per launch; us = GRBM_GUI_ACTIVE / 8 at 2.1 GHz; matrix pipes busy = (SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs) / (GRBM_GUI_ACTIVE / 8 XCDs);\n"
            "fetch = 2 x FETCH_SIZE (the guide's gfx950 correction), write = WRITE_SIZE, MB; instruction counts per launch; kernels above 2 us\n")
    f.write(f"{'kernel':64s} {'us':>8s} {'mfma busy':>10s} {'fetch MB':>10s} {'write MB':>10s} {'LDS confl/act':>14s} {'VALU':>12s} {'LDS':>10s} {'VMEM':>10s}\n")
    for us, k, busy, fe, wr, lds, iv, il, im in sorted(rows, reverse=True):
        f.write(f"{k:64s} {us:8.1f} {100*busy:9.1f}% {fe:10.1f} {wr:10.1f} {lds:14.3f} {iv:12.0f} {il:10.0f} {im:10.0f}\n")
print(open(O + '/r04_pmc_bench_table.txt').read())
PY
rm -rf $O/pmc
echo "summary at $(( $(date +%s) - T0 )) s" >> $O/timeline.txt
cp $O/r04_pmc_summary.json profiles/r04_pmc_summary.json      # (on the box) so that the bench line below reports the traffic measured on THESE kernel sources
timeout 150 python bench.py > $O/r04_bench.json 2> $O/r04_bench.err
echo "bench at $(( $(date +%s) - T0 )) s" >> $O/timeline.txt
(cd /tmp && timeout 60 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/tr_enc -o t -- python $R/scratch/r3/enc_only.py > $R/$O/enc.log 2>&1)
find $O/tr_enc -name "*kernel_stats.csv" -exec cp {} $O/r04_encode_kernel_stats.csv \;
rm -rf $O/tr_enc
echo "encode trace at $(( $(date +%s) - T0 )) s" >> $O/timeline.txt
if [ $(( $(date +%s) - T0 )) -lt 215 ]; then
  (cd /tmp && timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/bench_trace -o b -- python $R/bench.py > $R/$O/bench_under_rocprof.log 2>&1)
  grep "^{\"metric\"" $O/bench_under_rocprof.log | tail -1 > $O/r04_bench_under_rocprof.json
  find $O/bench_trace -name "*kernel_stats.csv" -exec cp {} $O/r04_bench_kernel_stats.csv \;
  rm -rf $O/bench_trace
  echo "bench trace at $(( $(date +%s) - T0 )) s" >> $O/timeline.txt
fi
cat $O/timeline.txt; ls -la $O
