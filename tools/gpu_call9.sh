#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -k "single_iteration or graph_chains" > gpurun_out/r3_c9_tests0.txt 2>&1
rc=$?
tail -5 gpurun_out/r3_c9_tests0.txt
if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -k "icp and not benchmark_shapes" > gpurun_out/r3_c9_tests.txt 2>&1
rc=$?
tail -5 gpurun_out/r3_c9_tests.txt
if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 300 python tools/single_pair_probe.py > gpurun_out/r3_c9_single.txt 2>&1
tail -12 gpurun_out/r3_c9_single.txt
