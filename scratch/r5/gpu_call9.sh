#!/bin/bash
# round-4 prototypes that were never measured: hoisted prefetch addressing, 16-wide tiles (scratch/r5_prep), + the bench legs added in round 5
o=gpurun_out/r5i; mkdir -p $o
for lib in libr5_pipelined.so libr5_hoist.so; do
  echo "== $lib" >> $o/r5prep.txt
  R5_LIB=$lib R5_GRIDS=512,-512,256,-256,1024,-1024 python scratch/r5_prep/check.py 2>&1 | grep -v "^$" | cut -c1-60,200-400 >> $o/r5prep.txt
done
R5_LIB=libr5_hoist.so python scratch/r5_prep/check.py 2>&1 | cut -c1-250 > $o/r5prep_check_hoist.txt
cat $o/r5prep.txt
tail -4 $o/r5prep_check_hoist.txt
( time python bench.py > $o/bench.json 2> $o/bench.err ) 2> $o/bench_time.txt
tail -3 $o/bench_time.txt
python - <<PY
import json
d=json.load(open("$o/bench.json"))
print(json.dumps(d["extras"].get("guard_tripped"))[:1500])
print("cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"])
PY
