"""Build libcare_hip.so (hand-written gfx950 kernels + C ABI) in-tree with hipcc.

    python -m care_amd.build [--force]

hipcc cross-compiles for gfx950 without a GPU.  Every source is compiled to an object of its own
(`care_amd/csrc/.obj/`, in parallel, only when it or a header it includes is newer) and the objects are
linked into the .so.  The .so is git-ignored but travels to the GPU box with the repo snapshot; the
objects do not (`.gpurunignore`).
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, ".obj")
LIB = os.path.join(HERE, "libcare_hip.so")
SOURCES = ("gemm.hip", "gemm_as.hip", "gemm_vocab.hip", "gemm_store32.hip", "gemm_tile.hip", "gemm_ln.hip", "rowops.hip",
           "attention.hip", "attention_seq.hip", "attention_latent.hip", "heads.hip", "beam.hip", "beam_sparse.hip",
           "compact.hip", "backward.hip", "decode_resident.hip", "decode_resident_beam.hip")
ARCH = "gfx950"


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    return "hipcc"


def _headers():
    hs = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith(".h")]
    return hs + [os.path.join(os.path.dirname(HERE), "include", "care_hip.h")]


def _sources():
    return [os.path.join(CSRC, s) for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]


def needs_build() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(d) > t for d in _sources() + _headers())


def build(force: bool = False, verbose: bool = True, jobs: int = 0, out: str = None, flags_extra=(), sources=None) -> str:
    """out / flags_extra / sources: a VARIANT build (tools: e.g. the fenced hand-offs of the resident decodes,
    -DRES_FENCED) into a library of its own, objects under .obj/<name>/; the default build is untouched."""
    if out:
        return _build_variant(out, list(flags_extra), sources, verbose)
    if not force and not needs_build():
        return LIB
    os.makedirs(OBJ, exist_ok=True)
    hdr_t = max(os.path.getmtime(h) for h in _headers())
    flags = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc"]
    extra = os.environ.get("CARE_HIPCC_FLAGS", "").split()
    todo, objs = [], []
    for src in _sources():
        obj = os.path.join(OBJ, os.path.basename(src)[:-4] + ".o")
        objs.append(obj)
        if force or extra or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), hdr_t):
            todo.append((src, obj))

    def compile_one(so):
        cmd = [_hipcc()] + flags + extra + ["-c", so[0], "-o", so[1]]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)

    jobs = jobs or int(os.environ.get("CARE_BUILD_JOBS", "0")) or min(8, os.cpu_count() or 1)
    with ThreadPoolExecutor(max_workers=max(1, jobs)) as pool:
        list(pool.map(compile_one, todo))
    cmd = [_hipcc(), "--offload-arch=" + ARCH, "-shared", "-fPIC", "-fno-gpu-rdc", "-o", LIB] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return LIB


def _build_variant(out: str, flags_extra, sources, verbose: bool) -> str:
    name = os.path.splitext(os.path.basename(out))[0]
    odir = os.path.join(OBJ, name)
    os.makedirs(odir, exist_ok=True)
    flags = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc"] + list(flags_extra)
    variant = set(sources or SOURCES)
    objs, todo = [], []
    for src in _sources():
        base = os.path.basename(src)
        if base in variant:
            obj = os.path.join(odir, base[:-4] + ".o")
            todo.append((src, obj))
        else:  # untouched sources: the default build's objects
            obj = os.path.join(OBJ, base[:-4] + ".o")
            if not os.path.exists(obj):
                raise RuntimeError("run the default build first ({} is missing)".format(obj))
        objs.append(obj)

    def compile_one(so):
        cmd = [_hipcc()] + flags + ["-c", so[0], "-o", so[1]]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)

    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as pool:
        list(pool.map(compile_one, todo))
    subprocess.run([_hipcc(), "--offload-arch=" + ARCH, "-shared", "-fPIC", "-fno-gpu-rdc", "-o", out] + objs, check=True)
    return out


if __name__ == "__main__":
    if "--fenced" in sys.argv:  # the resident decodes with agent-scope release / acquire fences at every hand-off
        print(build(out=os.path.join(HERE, "libcare_hip_fenced.so"), flags_extra=["-DRES_FENCED"],
                    sources=("decode_resident.hip", "decode_resident_beam.hip")))
        sys.exit(0)
    build(force="--force" in sys.argv)
    print(LIB)
