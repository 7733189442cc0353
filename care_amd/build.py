"""Build libcare_hip*.so (hand-written gfx950 kernels + C ABI) in-tree with hipcc.

    python -m care_amd.build [--force] [--variant f16] [--all]

hipcc cross-compiles for gfx950 without a GPU.  Every source is compiled to an object of its own (in parallel, only
when it or a header is newer) and the objects are linked into the .so.  The .so is git-ignored but travels to the GPU
box with the repo snapshot; the objects do not (`.gpurunignore`).

**The binary is bound to its sources.**  `source_hash(variant)` = SHA-256 over the bytes of every `csrc/*.hip`,
`csrc/*.h`, `include/care_hip.h` and the compile flags; it is compiled into the library (`csrc/version.hip`,
`care_source_hash()`), and `needs_build()` / `care_amd._lib.load()` compare the library's hash with the tree's - not
modification times.  A stale library is rebuilt (or, without hipcc, refused), never silently used.  Objects live under
`.obj/<variant>-<hash of the flags>/`, so a build with other flags (CARE_HIPCC_FLAGS, a tool's -D switches) never mixes
its objects with the default build's.

Variants (same sources, same ABI, a library each):
  ""    libcare_hip.so      16-bit storage / MFMA operands are bf16 (the throughput mode `bf16`; fp32 / fp16x3 modes too)
  "f16" libcare_hip_f16.so  the SAME kernels with IEEE fp16 as the 16-bit type (-DCARE_H16_FP16): compute mode `fp16`
"""
import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, ".obj")
SOURCES = ("gemm.hip", "gemm_as.hip", "gemm_vocab.hip", "gemm_store32.hip", "gemm_tile.hip", "gemm_ln.hip", "rowops.hip",
           "attention.hip", "attention_seq.hip", "attention_latent.hip", "heads.hip", "beam.hip", "beam_sparse.hip", "beam_pick.hip",
           "compact.hip", "backward.hip", "decode_resident.hip", "decode_resident_beam.hip", "decode_resident_beam_wide.hip", "decode_chain.hip")
VERSION_SRC = "version.hip"  # compiled on every link with the hash / flags of the build
ARCH = "gfx950"
BASE_FLAGS = ("--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc")
VARIANTS = {"": (), "f16": ("-DCARE_H16_FP16",)}
HASH_MARK = b"CARE_SRC_HASH="
LAST_BUILD = {}  # variant -> "compiled" | "reused" (what the last build() call of this process did)


def lib_path(variant: str = "") -> str:
    return os.path.join(HERE, "libcare_hip{}.so".format("_" + variant if variant else ""))


LIB = lib_path("")


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    return "hipcc"


def have_hipcc() -> bool:
    c = _hipcc()
    if os.path.isabs(c):
        return os.path.exists(c)
    from shutil import which
    return which(c) is not None


def _headers():
    hs = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith(".h")]
    return hs + [os.path.join(os.path.dirname(HERE), "include", "care_hip.h")]


def _sources():
    return [os.path.join(CSRC, s) for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]


def variant_flags(variant: str = "", flags_extra=()):
    if variant not in VARIANTS:
        raise ValueError("unknown library variant {!r} (have {})".format(variant, sorted(VARIANTS)))
    return list(VARIANTS[variant]) + os.environ.get("CARE_HIPCC_FLAGS", "").split() + list(flags_extra)


def have_sources() -> bool:
    return os.path.isdir(CSRC) and os.path.exists(os.path.join(CSRC, VERSION_SRC))


def source_hash(variant: str = "", flags_extra=()) -> str:
    """SHA-256 (hex, 32 chars) of the kernel sources, headers and compile flags of a variant."""
    h = hashlib.sha256()
    files = sorted(set(_sources() + _headers() + [os.path.join(CSRC, VERSION_SRC)]))
    for f in files:
        h.update(os.path.basename(f).encode() + b"\0")
        with open(f, "rb") as fh:
            h.update(fh.read())
        h.update(b"\0")
    h.update(" ".join(list(BASE_FLAGS) + variant_flags(variant, flags_extra)).encode())
    return h.hexdigest()[:32]


def embedded_hash(path: str):
    """The source hash a built library carries (None: not a library of this build system)."""
    try:
        with open(path, "rb") as fh:
            blob = fh.read()
    except OSError:
        return None
    i = blob.find(HASH_MARK)
    if i < 0:
        return None
    return blob[i + len(HASH_MARK): i + len(HASH_MARK) + 32].decode("ascii", "replace")


def needs_build(variant: str = "") -> bool:
    p = lib_path(variant)
    if not os.path.exists(p):
        return True
    if not have_sources():
        return False
    return embedded_hash(p) != source_hash(variant)


def _compile_link(out: str, variant: str, flags_extra, force: bool, verbose: bool, jobs: int) -> None:
    vflags = variant_flags(variant, flags_extra)
    flags = list(BASE_FLAGS) + vflags
    key = hashlib.sha256(" ".join(flags).encode()).hexdigest()[:8]
    odir = os.path.join(OBJ, "{}-{}".format(variant or "default", key))
    os.makedirs(odir, exist_ok=True)
    hdr_t = max(os.path.getmtime(h) for h in _headers())
    todo, objs = [], []
    for src in _sources():
        obj = os.path.join(odir, os.path.basename(src)[:-4] + ".o")
        objs.append(obj)
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), hdr_t):
            todo.append((src, obj, ()))
    vobj = os.path.join(odir, "version.o")
    objs.append(vobj)
    todo.append((os.path.join(CSRC, VERSION_SRC), vobj,
                 ('-DCARE_SRC_HASH_STR="{}"'.format(source_hash(variant, flags_extra)),
                  '-DCARE_BUILD_FLAGS_STR="{}"'.format(" ".join(vflags).replace('"', "'")))))

    def compile_one(job):
        cmd = [_hipcc()] + flags + list(job[2]) + ["-c", job[0], "-o", job[1]]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)

    jobs = jobs or int(os.environ.get("CARE_BUILD_JOBS", "0")) or min(8, os.cpu_count() or 1)
    # the slowest translation units first (the resident decodes take minutes, most others seconds)
    slow = ("decode_resident_beam", "decode_resident", "decode_chain", "gemm_tile", "attention.")
    todo.sort(key=lambda j: next((i for i, s in enumerate(slow) if s in os.path.basename(j[0])), len(slow)))
    with ThreadPoolExecutor(max_workers=max(1, jobs)) as pool:
        list(pool.map(compile_one, todo))
    tmp = "{}.{}.tmp".format(out, os.getpid())   # (a name of this process' own: nobody else's link or replace touches it)
    cmd = [_hipcc(), "--offload-arch=" + ARCH, "-shared", "-fPIC", "-fno-gpu-rdc", "-o", tmp] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    try:
        subprocess.run(cmd, check=True)
        os.replace(tmp, out)
    finally:
        if os.path.exists(tmp):
            os.remove(tmp)


class _BuildLock:
    """Exclusive advisory lock (fcntl.flock on .obj/build.lock) around a compile + link.  Under torchrun every rank imports
    care_amd at once and sees the same stale hash: without the lock they would all run hipcc into the same objects and
    replace each other's half-linked library.  With it ONE builds; the others block here, then find the library current
    (the caller re-checks needs_build once it holds the lock) and load it."""

    def __enter__(self):
        import fcntl
        os.makedirs(OBJ, exist_ok=True)
        self.fh = open(os.path.join(OBJ, "build.lock"), "w")
        fcntl.flock(self.fh, fcntl.LOCK_EX)
        return self

    def __exit__(self, *exc):
        import fcntl
        fcntl.flock(self.fh, fcntl.LOCK_UN)
        self.fh.close()


def build(force: bool = False, verbose: bool = True, jobs: int = 0, out: str = None, flags_extra=(), variant: str = "") -> str:
    """Build (or reuse) the library of `variant`; LAST_BUILD[variant] says which.  out / flags_extra: a TOOL build
    (ablation switches, e.g. -DRES_NOINLINE) into a library of its own - objects keyed by the flags, the default
    library untouched."""
    if out:
        with _BuildLock():
            _compile_link(out, variant, flags_extra, force, verbose, jobs)
        return out
    p = lib_path(variant)
    if not force and not needs_build(variant):
        LAST_BUILD[variant] = "reused"
        return p
    with _BuildLock():
        if not force and not needs_build(variant):   # another process built it while this one waited for the lock
            LAST_BUILD[variant] = "reused"
            return p
        _compile_link(p, variant, (), force, verbose, jobs)
    LAST_BUILD[variant] = "compiled"
    return p


def build_all(force: bool = False, verbose: bool = True) -> dict:
    return {v: build(force=force, verbose=verbose, variant=v) for v in VARIANTS}


if __name__ == "__main__":
    force = "--force" in sys.argv
    if "--all" in sys.argv:
        for v, p in build_all(force=force).items():
            print("{} [{}] {}".format(p, LAST_BUILD[v], embedded_hash(p)))
        sys.exit(0)
    v = sys.argv[sys.argv.index("--variant") + 1] if "--variant" in sys.argv else ""
    p = build(force=force, variant=v)
    print("{} [{}] {}".format(p, LAST_BUILD[v], embedded_hash(p)))
