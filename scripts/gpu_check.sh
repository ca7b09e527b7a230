#!/bin/bash
# On the GPU box (via gpurun): the GPU test suite, then optional extra test files named on the command line.
set -u
mkdir -p gpurun_out
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/pytest_gpu.log 2>&1 || { tail -30 gpurun_out/pytest_gpu.log; exit 1; }
tail -3 gpurun_out/pytest_gpu.log
for f in "$@"; do
  timeout -k 10 600 python -u -m pytest "$f" -m gpu -x -v --timeout 300 --timeout-method thread > gpurun_out/pytest_extra.log 2>&1 || { tail -40 gpurun_out/pytest_extra.log; exit 1; }
  tail -8 gpurun_out/pytest_extra.log
done
