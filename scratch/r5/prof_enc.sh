#!/bin/bash
# rocprofv3 kernel trace of the scene encodes of scratch/r3/enc_only.py (config 2) -> gpurun_out/$1/enc_kernel_stats.csv + a summary on stdout
out=gpurun_out/${1:-r5}
export TMPDIR=/tmp
mkdir -p $out
rm -rf $out/prof_enc
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_enc -o e -- python scratch/r3/enc_only.py > $out/prof_enc.log 2>&1
tail -1 $out/prof_enc.log | cut -c1-300
find $out/prof_enc -name "*kernel_stats.csv" -exec cp {} $out/enc_kernel_stats.csv \;
rm -rf $out/prof_enc
python - <<PY
import csv
rows=list(csv.DictReader(open("$out/enc_kernel_stats.csv")))
for r in rows[:14]:
    print(r["Name"][:70].ljust(70), r["Calls"].rjust(4), "%9.1f us" % (float(r["AverageNs"])/1e3), r["Percentage"])
PY
