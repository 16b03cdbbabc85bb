#!/bin/bash
# round 4, call 24: build() + smoke() as the driver runs them, and bench.py --gpus 2 --shared-gpu-dry-run
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
timeout 900 python -c "import __graft_entry__ as g; g.build(); g.smoke(); print('smoke ok')" > gpurun_out/c24_smoke.log 2>&1; echo "smoke rc $?"; tail -5 gpurun_out/c24_smoke.log
timeout 900 python bench.py --gpus 2 --shared-gpu-dry-run > gpurun_out/c24_dry.json 2> gpurun_out/c24_dry.err; echo "dry rc $?"; tail -c 400 gpurun_out/c24_dry.json
