#!/bin/bash
for v in "$@"; do
  python bench.py --steps 200 --warmup 20 --cpu-batches 0 --mlp-variant $v 2>/dev/null | tail -1 > /tmp/b.json
  python - <<PY
import json
d=json.load(open('/tmp/b.json'))
print("variant $v rays/s", d["value"], "ms/step", d["ms_per_step"], "mlp TF", d["roofline"]["achieved"], "mlp ms", d["roofline"]["avg_launch_ms"])
PY
done
