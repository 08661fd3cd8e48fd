#!/bin/bash
# tools/build_variant.sh NAME "-DMACRO=..." : libgpscal_hip.so with knn_icp.hip compiled under extra flags, as
# variants/libgpscal_NAME.so (git-ignored, travels to the GPU box); select it with GPSCAL_LIB=variants/libgpscal_NAME.so
set -e
cd "$(dirname "$0")/../gpscalibration_amd/csrc"
make -s -j8
name=$1; shift
mkdir -p ../../variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-result -ffp-contract=off "$@" -c knn_icp.hip -o ../../variants/knn_icp_$name.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../variants/libgpscal_$name.so ../../variants/knn_icp_$name.o loam.o sr.o loam_pipeline.o track.o geo.o api.o -ldl
echo variants/libgpscal_$name.so
