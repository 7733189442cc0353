"""Tool builds of ONE translation unit with extra -D switches, linked with the default build's other objects:

    python tools/variant_lib.py gemm_ln.hip out.so -DCARE_LN_DBG=6 [...]

(ablation / experiment libraries for CARE_HIP_LIB=...; care_amd.build's `out=` form recompiles every source with the
flags, minutes for the resident decodes).  The default library must be built first."""
import hashlib
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from care_amd import build


def main():
    src, out, extra = sys.argv[1], sys.argv[2], sys.argv[3:]
    build.build(verbose=False)
    flags = list(build.BASE_FLAGS) + build.variant_flags("")
    odir = os.path.join(build.OBJ, "default-" + hashlib.sha256(" ".join(flags).encode()).hexdigest()[:8])
    objs = [os.path.join(odir, f) for f in sorted(os.listdir(odir)) if f.endswith(".o") and f != src[:-4] + ".o"]
    tdir = os.path.join(build.OBJ, "tool")
    os.makedirs(tdir, exist_ok=True)
    obj = os.path.join(tdir, os.path.basename(out) + "." + src[:-4] + ".o")
    subprocess.run([build._hipcc()] + flags + extra + ["-c", os.path.join(build.CSRC, src), "-o", obj], check=True)
    subprocess.run([build._hipcc(), "--offload-arch=" + build.ARCH, "-shared", "-fPIC", "-fno-gpu-rdc", "-o", out, obj] + objs, check=True)
    print(out)


if __name__ == "__main__":
    main()
