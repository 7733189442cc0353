#!/bin/bash
# Ablation builds of the library: tools/lat_ablate.sh NAME "-DCARE_LAT_DBG=4 ..." [NAME FLAGS ...] -> care_amd/dbg/libcare_hip_NAME.so
# (CARE_LAT_DBG bits: 1 no ct stores, 2 no qt loads, 4 no arithmetic, 8 direct 8-byte stores; CARE_LAT_LD_AUX / CARE_LAT_ST_NT: cache policy)
set -e
cd "$(dirname "$0")/.."
mkdir -p care_amd/dbg
SRC="care_amd/csrc/gemm.hip care_amd/csrc/gemm_as.hip care_amd/csrc/gemm_vocab.hip care_amd/csrc/gemm_store32.hip care_amd/csrc/gemm_ln.hip care_amd/csrc/rowops.hip care_amd/csrc/attention.hip care_amd/csrc/attention_latent.hip care_amd/csrc/heads.hip care_amd/csrc/beam.hip care_amd/csrc/beam_sparse.hip care_amd/csrc/compact.hip"
while [ $# -ge 2 ]; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -fno-gpu-rdc $2 -o care_amd/dbg/libcare_hip_$1.so $SRC &
  shift 2
done
wait
