#!/bin/bash
# use_amp training step: kernel statistics + step time + the bf16 / training tests (after a change to a bf16 training kernel)
out=gpurun_out/$1; mkdir -p $out
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/tr -o t -- python $GRAFT_REPO_ROOT/scratch/r6/train_prof.py fp32 4 > /dev/null 2>&1)
find $out/tr -name "*kernel_stats.csv" -exec cp {} $out/fp32_kernel_stats.csv \; ; rm -rf $out/tr
python scratch/r6/train_prof.py fp32 10 2>&1 | grep -v amdgpu.ids > $out/step.txt; cat $out/step.txt
timeout 1500 python -m pytest tests/test_gpu_encoder.py tests/test_gpu_headline_parity.py tests/test_gpu_guard.py tests/test_gpu_fp16x3_encoder.py tests/test_gpu_train.py tests/test_gpu_backward.py tests/test_gpu_shared.py tests/test_gpu_costream.py -x -q 2>&1 | tail -4 > $out/tests.txt
cat $out/tests.txt
