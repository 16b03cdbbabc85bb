#!/bin/bash
# three builds of the same sources on one box: tests that changed with the flags + the co-stream bisect
o=gpurun_out/r5e; mkdir -p $o
cp mvsnerf_amd/lib/libmvsnerf_hip.so /tmp/lib_nopk.so
for v in nopk r4flags noslp; do
  if [ $v = nopk ]; then cp /tmp/lib_nopk.so mvsnerf_amd/lib/libmvsnerf_hip.so; else cp scratch/lib/libmvsnerf_hip_$v.so mvsnerf_amd/lib/libmvsnerf_hip.so; fi
  python -m pytest "tests/test_gpu_signatures.py" tests/test_gpu_train.py::test_training_step_matches_oracle_autograd tests/test_gpu_guard.py::test_two_streams_do_not_share_a_guard_buffer -m gpu -q 2>&1 | grep -v "^$" | grep "^E  .*Error\|passed\|failed\|^FAILED" | cut -c1-300 > $o/tests_$v.log
  python scratch/r5/costream_bisect.py amp > $o/bisect_amp_$v.log 2>&1
  python scratch/r5/costream_bisect.py fp32 > $o/bisect_fp32_$v.log 2>&1
  echo "== $v"; cat $o/tests_$v.log; grep "<<<\|worst\|use_amp" $o/bisect_amp_$v.log | cut -c1-200 | head -40; grep -c "<<<" $o/bisect_fp32_$v.log
done
cp /tmp/lib_nopk.so mvsnerf_amd/lib/libmvsnerf_hip.so
