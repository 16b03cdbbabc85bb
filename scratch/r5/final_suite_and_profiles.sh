#!/bin/bash
# after the per-kernel no-packed-fp32 attribute on the fallback wgrad kernel + check_isa.sh: full GPU suite, smoke, then the evidence of the final tree
o=gpurun_out/r5m; mkdir -p $o
python -m pytest tests -m gpu -q 2>&1 | grep -v "^$" > $o/pytest_gpu.log
grep -n "^E  .*Error\|passed\|failed\|^FAILED" $o/pytest_gpu.log | cut -c1-300 | tail -12
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
bash scratch/r5/final_profiles.sh 2>&1 | tail -15
