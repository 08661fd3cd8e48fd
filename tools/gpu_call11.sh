#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -k "icp and not benchmark_shapes" > gpurun_out/r3_c11_tests.txt 2>&1
rc=$?
tail -5 gpurun_out/r3_c11_tests.txt
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "tests timed out: stopping"; exit $rc; fi
for lib in variants/libgpscal_base.so gpscalibration_amd/libgpscal_hip.so; do
  echo "== $lib"
  GPSCAL_LIB=$lib timeout -k 10 600 python tools/step_ab.py 64 65536 50 default 2>&1 | grep -v amdgpu.ids | cut -c1-330
  GPSCAL_LIB=$lib timeout -k 10 600 python tools/step_ab.py 16 262144 20 default 2>&1 | grep -v amdgpu.ids | cut -c1-330
  GPSCAL_LIB=$lib GPSCAL_BUILD_TIMING=1 timeout -k 10 300 python tools/build_probe.py 2>&1 | tail -4
done > gpurun_out/r3_c11_ab.txt 2>&1
cat gpurun_out/r3_c11_ab.txt
