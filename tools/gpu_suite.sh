#!/bin/bash
# the whole GPU test suite, as the driver runs it at round end; output under gpurun_out/
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests/ -q -m gpu --durations=8 > gpurun_out/gpu_suite.txt 2>&1
rc=$?
tail -25 gpurun_out/gpu_suite.txt
exit $rc
