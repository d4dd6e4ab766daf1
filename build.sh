#!/bin/bash
# Build libdvsg_amd.so (gfx950 only) in-tree.  Usage: ./build.sh [extra hipcc flags]
set -euo pipefail
cd "$(dirname "$0")"
SRC=coupe/dvsg_amd/csrc
OUT=coupe/dvsg_amd/libdvsg_amd.so
OBJ=build/obj
mkdir -p "$OBJ"
COMMON="-O3 --offload-arch=gfx950 -fPIC -std=c++17 -Wall -Wno-unused-function $*"
pids=()
# warps and conv1's fused scale_RGB: separately rounded float32 ops like the reference graph
hipcc $COMMON -ffp-contract=off -c $SRC/warp_kernels.hip -o $OBJ/warp_kernels.o & pids+=($!)
hipcc $COMMON -ffp-contract=off -c $SRC/conv1_pool.hip -o $OBJ/conv1_pool.o & pids+=($!)
# frame formats: separately rounded float64 ops like NumPy / OpenCV on the host
hipcc $COMMON -ffp-contract=off -c $SRC/frames.hip -o $OBJ/frames.o & pids+=($!)
hipcc $COMMON -c $SRC/conv_gemm.hip -o $OBJ/conv_gemm.o & pids+=($!)
hipcc $COMMON -c $SRC/conv_gemm_x3.hip -o $OBJ/conv_gemm_x3.o & pids+=($!)
hipcc $COMMON -c $SRC/conv_fused.hip -o $OBJ/conv_fused.o & pids+=($!)
hipcc $COMMON -c $SRC/conv_fused_x3.hip -o $OBJ/conv_fused_x3.o & pids+=($!)
hipcc $COMMON -c $SRC/conv_gemm_wide16.hip -o $OBJ/conv_gemm_wide16.o & pids+=($!)
hipcc $COMMON -c $SRC/head.hip -o $OBJ/head.o & pids+=($!)
hipcc $COMMON -c $SRC/locnet.hip -o $OBJ/locnet.o & pids+=($!)
hipcc $COMMON -x hip -c $SRC/api_common.cpp -o $OBJ/api_common.o & pids+=($!)
for p in "${pids[@]}"; do wait "$p"; done
hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT $OBJ/warp_kernels.o $OBJ/conv1_pool.o $OBJ/conv_gemm.o $OBJ/conv_gemm_x3.o $OBJ/conv_gemm_wide16.o $OBJ/conv_fused.o $OBJ/conv_fused_x3.o \
  $OBJ/head.o $OBJ/locnet.o $OBJ/frames.o $OBJ/api_common.o
echo "built $OUT"
