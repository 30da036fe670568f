"""Builds libmacx.so (hand-written HIP for gfx950) in-tree with hipcc.

No torch.utils.cpp_extension (it hipifies), no Triton, no BLAS: one translation unit, one
`hipcc --offload-arch=gfx950 -shared -fPIC` command.  The .so is git-ignored but travels to the GPU
box with the repo snapshot.
"""
import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
ROOT = os.path.dirname(HERE)
LIB_DIR = os.path.join(HERE, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libmacx.so")
STAMP = LIB_PATH + ".stamp"
SOURCES = ["macx_api.hip"]


def _digest():
    """sha256 over every file the translation unit can include: all of csrc/ and include/ (a fixed header list once let
    edits to a newly added header go unbuilt)."""
    h = hashlib.sha256()
    files = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC))]
    inc = os.path.join(ROOT, "include")
    files += [os.path.join(inc, f) for f in sorted(os.listdir(inc))]
    for p in files:
        if os.path.isfile(p):
            h.update(os.path.basename(p).encode())
            with open(p, "rb") as fh:
                h.update(fh.read())
    return h.hexdigest()


def hipcc_path():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def build(force=False, verbose=False):
    """Compile libmacx.so for gfx950 if sources changed.  Returns the library path."""
    os.makedirs(LIB_DIR, exist_ok=True)
    dig = _digest()
    if not force and os.path.exists(LIB_PATH) and os.path.exists(STAMP):
        with open(STAMP) as fh:
            if fh.read().strip() == dig:
                return LIB_PATH
    cmd = [hipcc_path(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC",
           "-Wno-pass-failed", "-Wno-inline-asm", "-I", os.path.join(ROOT, "include")]
    cmd += [os.path.join(CSRC, s) for s in SOURCES]
    cmd += ["-o", LIB_PATH]
    # the register / spill / occupancy remarks of every kernel ride along (tools/kernel_resources.py --from-build reads them)
    cmd += ["-Rpass-analysis=kernel-resource-usage"]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    res = subprocess.run(cmd, stderr=subprocess.PIPE, text=True)
    with open(os.path.join(LIB_DIR, "kernel_resources.raw"), "w") as fh:
        fh.write(res.stderr)
    if res.returncode != 0:
        sys.stderr.write("\n".join(l for l in res.stderr.splitlines() if "remark:" not in l) + "\n")
        raise subprocess.CalledProcessError(res.returncode, cmd)
    with open(STAMP, "w") as fh:
        fh.write(dig)
    return LIB_PATH


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
