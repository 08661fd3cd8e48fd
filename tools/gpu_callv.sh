#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 600 python tools/step_ab.py 64 65536 50 default GPSCAL_ICP_MULTI_BELOW=0.2 GPSCAL_ICP_MULTI_BELOW=0.35 GPSCAL_ICP_MULTI_BELOW=0.5 2>&1 | grep -v amdgpu.ids | cut -c1-330 > gpurun_out/r3_v_ab.txt
cat gpurun_out/r3_v_ab.txt
