#!/bin/bash
# Ablation builds of the fused GEMM + LayerNorm kernel (csrc/gemm_ln.hip, CARE_LN_DBG): which part of
# a K step costs what.  Run here (CPU, hipcc cross-compiles), then time on the GPU box with
#   CARE_HIP_LIB=care_amd/dbg/libcare_hip_dbgN.so python tools/gemm_bench.py ln
set -e
cd "$(dirname "$0")/.."
mkdir -p care_amd/dbg
SRC="care_amd/csrc/gemm.hip care_amd/csrc/gemm_as.hip care_amd/csrc/gemm_vocab.hip care_amd/csrc/gemm_store32.hip care_amd/csrc/gemm_ln.hip care_amd/csrc/rowops.hip care_amd/csrc/attention.hip care_amd/csrc/attention_latent.hip care_amd/csrc/heads.hip care_amd/csrc/beam.hip care_amd/csrc/beam_sparse.hip care_amd/csrc/compact.hip"
for d in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -fno-gpu-rdc -DCARE_LN_DBG=$d -o care_amd/dbg/libcare_hip_dbg$d.so $SRC &
done
wait
ls -la care_amd/dbg
