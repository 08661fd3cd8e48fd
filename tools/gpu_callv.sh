#!/bin/bash
mkdir -p gpurun_out
( timeout -k 10 600 python tools/step_ab.py 64 65536 50 default GPSCAL_ICP_MULTI_BELOW=0 ; timeout -k 10 600 python tools/step_ab.py 8 1048576 50 default GPSCAL_ICP_MULTI_BELOW=0 GPSCAL_ICP_MULTI_BELOW=0.03 ) 2>&1 | grep -v amdgpu.ids | cut -c1-330 > gpurun_out/r3_v_ab.txt
cat gpurun_out/r3_v_ab.txt
