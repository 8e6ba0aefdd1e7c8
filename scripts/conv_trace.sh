#!/bin/bash
# Diagnostics: build librqamd_trace.so = the library with conv_halo.hip compiled under -DRQ_CONV_TRACE=<block> (shader-clock
# stamps of that workgroup's barriers), for scripts/conv_trace.py.  Usage (in the build container): bash scripts/conv_trace.sh [block]
set -e
cd "$(dirname "$0")/../rq-vae-transformer_amd"
python build.py > /dev/null
BLK=${1:-1000}
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -I csrc -DRQ_CONV_TRACE=$BLK $RQ_TRACE_DEFS -c csrc/conv_halo.hip -o build/conv_halo_trace.o
OBJS=$(ls build/*.hip.o | grep -v conv_halo.hip.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ${RQ_TRACE_OUT:-librqamd_trace.so} $OBJS build/conv_halo_trace.o
echo built ${RQ_TRACE_OUT:-librqamd_trace.so}
