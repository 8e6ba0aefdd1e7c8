#!/bin/bash
# Diagnostics: librqamd_trace.so = the library with gemm.hip compiled under -DRQ_GEMM_TRACE=<block> (shader-clock stamps of that
# workgroup's epilogue phases in the 256x256 eight-phase kernel), for scripts/gemm_trace.py.  bash scripts/gemm_trace.sh [block]
set -e
cd "$(dirname "$0")/../rq-vae-transformer_amd"
python build.py > /dev/null
BLK=${1:-300}
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -I csrc -DRQ_GEMM_TRACE=$BLK -c csrc/gemm.hip -o build/gemm_trace.o
OBJS=$(ls build/*.hip.o | grep -v "build/gemm.hip.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o librqamd_trace.so $OBJS build/gemm_trace.o
echo built librqamd_trace.so
