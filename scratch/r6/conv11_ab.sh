#!/bin/bash
# conv11 epilogue through LDS (1 KB row stores): same-box A/B of the encode and the training steps against scratch/lib/libmvsnerf_hip_encbase.so, bit-identity of the volume
out=gpurun_out/$1; mkdir -p $out
for rep in 1 2; do
  python scratch/r6/enc_ab.py
  MVS_LIB=scratch/lib/libmvsnerf_hip_encbase.so python scratch/r6/enc_ab.py
done 2>&1 | grep -v amdgpu.ids > $out/ab.txt
cat $out/ab.txt
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/tr -o t -- python $GRAFT_REPO_ROOT/scratch/r6/enc_only.py > /dev/null 2>&1)
find $out/tr -name "*kernel_stats.csv" -exec cp {} $out/enc_kernel_stats.csv \; ; rm -rf $out/tr
cut -d, -f1-4 $out/enc_kernel_stats.csv | cut -c1-140 | head -24
timeout 1500 python -m pytest tests/test_gpu_encoder.py tests/test_gpu_fp16x3_encoder.py tests/test_gpu_train.py tests/test_gpu_backward.py tests/test_gpu_headline_parity.py tests/test_gpu_configs45.py tests/test_gpu_layout.py tests/test_gpu_bf16_encoder.py tests/test_gpu_bf16_layers.py tests/test_gpu_featnet.py -x -q 2>&1 | tail -4 > $out/tests.txt
cat $out/tests.txt
