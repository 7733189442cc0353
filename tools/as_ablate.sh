#!/bin/bash
# Ablation builds of the A-stationary GEMM (csrc/gemm_as.hip, CARE_AS_DBG bits): what a W tile costs.
#   CARE_HIP_LIB=care_amd/dbg/libcare_hip_as<N>.so python tools/argmax_bench.py
set -e
cd "$(dirname "$0")/.."
mkdir -p care_amd/dbg
SRC="care_amd/csrc/gemm.hip care_amd/csrc/gemm_as.hip care_amd/csrc/gemm_vocab.hip care_amd/csrc/gemm_store32.hip care_amd/csrc/gemm_ln.hip care_amd/csrc/rowops.hip care_amd/csrc/attention.hip care_amd/csrc/attention_latent.hip care_amd/csrc/heads.hip care_amd/csrc/beam.hip care_amd/csrc/beam_sparse.hip care_amd/csrc/compact.hip"
for d in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -fno-gpu-rdc -DCARE_AS_DBG=$d -o care_amd/dbg/libcare_hip_as$d.so $SRC &
done
wait
