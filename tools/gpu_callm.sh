#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -k "icp and not benchmark_shapes" > gpurun_out/r3_m_tests.txt 2>&1
rc=$?
tail -15 gpurun_out/r3_m_tests.txt
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "tests timed out: stopping"; exit $rc; fi
timeout -k 10 600 python tools/step_ab.py 64 65536 50 GPSCAL_ICP_MULTI_BELOW=0 default GPSCAL_ICP_MULTI_BELOW=0.01 GPSCAL_ICP_MULTI_BELOW=0.1 2>&1 | grep -v amdgpu.ids | cut -c1-330 > gpurun_out/r3_m_ab.txt
cat gpurun_out/r3_m_ab.txt
