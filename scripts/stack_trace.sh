#!/bin/bash
# Diagnostics: librqamd_trace.so = the library with rqt_kernels.hip compiled under -DRQ_STACK_TRACE=<workgroup> (constant-clock stamps
# of every phase of the persistent stack kernel), for scripts/stack_trace.py.  bash scripts/stack_trace.sh [workgroup]
set -e
cd "$(dirname "$0")/../rq-vae-transformer_amd"
python build.py > /dev/null
WG=${1:-200}
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -I csrc -DRQ_STACK_TRACE=$WG -c csrc/rqt_kernels.hip -o build/rqt_kernels_trace.o
OBJS=$(ls build/*.hip.o | grep -v "build/rqt_kernels.hip.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o librqamd_trace.so $OBJS build/rqt_kernels_trace.o
echo built librqamd_trace.so
