#!/bin/bash
# usage: bash scratch/run_one.sh <pytest args...>   -> gpurun_out/one.log
mkdir -p gpurun_out
timeout 2400 python -m pytest "$@" -x -q -s 2>&1 | tail -60 > gpurun_out/one.log
