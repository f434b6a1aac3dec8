#!/bin/bash
# diagnostic build of the library with the 256 x 256 gemm_nt kernel's phase stamps (-DCMPC_V5_TRACE) -> build/libcmpc_trace.so
# (run here, before gpurun: the .so travels to the GPU box; scripts/v5_trace.py loads it through CMPC_LIB_PATH)
set -e
cd "$(dirname "$0")/../cmpc-refseg_amd/csrc"
make
mkdir -p ../../build
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DCMPC_V5_TRACE -c gemm.hip -o /tmp/gemm_trace.o
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../../build/libcmpc_trace.so /tmp/gemm_trace.o ops_norm.o ops_fusion.o ops_convlstm.o ops_score.o ops_lang.o ops_crf.o engine.o
ls -la ../../build/libcmpc_trace.so
