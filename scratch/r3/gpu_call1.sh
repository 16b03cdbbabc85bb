#!/bin/bash
# round 3, re-entry, GPU call 1: LDS-DMA probe, the new fp16x3 kernel's tests + the ray-march suite (refactor check), host-time diagnosis of
# the split modes, one bench line.  Everything lands under gpurun_out/r3b/.
mkdir -p gpurun_out/r3b
O=gpurun_out/r3b
./scratch/r3/lds_dma_hi_probe > $O/lds_dma_hi_probe.txt 2>&1; echo "probe rc $?" >> $O/lds_dma_hi_probe.txt
timeout 600 python -m pytest tests/test_gpu_fp16x3.py -q -s > $O/test_fp16x3.log 2>&1; echo "rc $?" >> $O/test_fp16x3.log
timeout 600 python -m pytest tests/test_gpu_raymarch.py -x -q > $O/test_raymarch.log 2>&1; echo "rc $?" >> $O/test_raymarch.log
timeout 300 python scratch/r3/split_step_diag.py > $O/split_step_diag.txt 2>&1; echo "rc $?" >> $O/split_step_diag.txt
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; echo "rc $?" >> $O/bench.err
tail -5 $O/lds_dma_hi_probe.txt; tail -15 $O/test_fp16x3.log; tail -3 $O/test_raymarch.log; head -12 $O/split_step_diag.txt | cut -c1-400; tail -c 1500 $O/bench.json
