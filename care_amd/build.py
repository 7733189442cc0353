"""Build libcare_hip.so (hand-written gfx950 kernels + C ABI) in-tree with hipcc.

    python -m care_amd.build [--force]

hipcc cross-compiles for gfx950 without a GPU.  The .so is git-ignored but travels to
the GPU box with the repo snapshot.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libcare_hip.so")
SOURCES = ("gemm.hip", "gemm_as.hip", "gemm_vocab.hip", "gemm_store32.hip", "gemm_tile.hip", "gemm_ln.hip", "rowops.hip", "attention.hip", "attention_seq.hip", "attention_latent.hip", "heads.hip", "beam.hip", "beam_sparse.hip", "compact.hip", "backward.hip", "decode_resident.hip")
ARCH = "gfx950"


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    return "hipcc"


def needs_build() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    deps += [os.path.join(CSRC, "care_common.h"), os.path.join(os.path.dirname(HERE), "include", "care_hip.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = True) -> str:
    if not force and not needs_build():
        return LIB
    srcs = [os.path.join(CSRC, s) for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    cmd = [_hipcc(), "--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-shared",
           "-fgpu-rdc" if False else "-fno-gpu-rdc", "-o", LIB] + srcs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
