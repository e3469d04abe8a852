#!/bin/bash
# first-contact script for a gpurun box
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
make -s -C oracle oracle 2>&1 | tail -3
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -s 2>&1 | tail -60 > gpurun_out/first_check.log
cat gpurun_out/first_check.log
