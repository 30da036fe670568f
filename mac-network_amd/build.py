"""Builds libmacx.so (hand-written HIP for gfx950) in-tree with hipcc.

No torch.utils.cpp_extension (it hipifies), no Triton, no BLAS: three translation units compiled in parallel with
`hipcc --offload-arch=gfx950 -c -fPIC`, one `hipcc -shared` link.  The .so is git-ignored but travels to the GPU
box with the repo snapshot.
"""
import concurrent.futures
import hashlib
import os
import re
import subprocess
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
ROOT = os.path.dirname(HERE)
LIB_DIR = os.path.join(HERE, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libmacx.so")
STAMP = LIB_PATH + ".stamp"
# translation units: the API + every kernel but the chain kernels | the forward chain kernels | the backward chain kernels
SOURCES = ["macx_api.hip", "macx_chain_fwd.hip", "macx_chain_bwd.hip"]


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
    h.update(os.environ.get("MACX_BUILD_DEFINES", "").encode())
    return h.hexdigest()


def hipcc_path():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


_DIAG = re.compile(r"^\S.*?:\d+:\d+: (warning|error|remark|note|fatal error):")


def _without_remarks(stderr_text):
    """clang's stderr minus its -Rpass remark blocks (the remark line, its `In file included from` preamble, the quoted source line
    and the caret line): what is left are warnings, errors and their notes."""
    out, pre, dropping = [], [], False
    for line in stderr_text.splitlines():
        if line.startswith("In file included from"):
            pre.append(line)
            continue
        m = _DIAG.match(line)
        if m:
            dropping = m.group(1) == "remark"
            if not dropping:
                out += pre
            pre = []
        if not dropping and not re.match(r"^\d+ (remark|warning)s? generated", line) and line.strip():
            out.append(line)
    return "\n".join(out)


def _compile_one(src, obj, verbose):
    cmd = [hipcc_path(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-c", "-fPIC", "-Wno-pass-failed", "-Wno-inline-asm",
           "-I", os.path.join(ROOT, "include"), os.path.join(CSRC, src), "-o", obj]
    for name in os.environ.get("MACX_BUILD_DEFINES", "").split():     # e.g. MACX_FILL_PROF, MACX_PROFILE_VARIANTS (measurement builds)
        cmd += ["-D" + name]
    if os.environ.get("MACX_BUILD_REMARKS", "1") != "0":
        # the register / spill / occupancy remarks of every kernel ride along (tools/kernel_resources.py --from-build reads them)
        cmd += ["-Rpass-analysis=kernel-resource-usage"]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    t0 = time.time()
    res = subprocess.run(cmd, stderr=subprocess.PIPE, text=True)
    return src, res.returncode, res.stderr, time.time() - t0, cmd


def build(force=False, verbose=False):
    """Compile libmacx.so for gfx950 if sources changed: the translation units in parallel, then one link.  Returns the library path."""
    os.makedirs(LIB_DIR, exist_ok=True)
    dig = _digest()
    if not force and os.path.exists(LIB_PATH) and os.path.exists(STAMP):
        with open(STAMP) as fh:
            if fh.read().strip() == dig:
                return LIB_PATH
    obj_dir = os.path.join(LIB_DIR, "obj")
    os.makedirs(obj_dir, exist_ok=True)
    objs = [os.path.join(obj_dir, os.path.splitext(s)[0] + ".o") for s in SOURCES]
    with concurrent.futures.ThreadPoolExecutor(max_workers=min(len(SOURCES), os.cpu_count() or 1)) as ex:
        results = list(ex.map(lambda so: _compile_one(so[0], so[1], verbose), zip(SOURCES, objs)))
    with open(os.path.join(LIB_DIR, "kernel_resources.raw"), "w") as fh:
        for _, _, err, _, _ in results:
            fh.write(err)
    for src, rc, err, dt, cmd in results:
        # warnings and errors are shown whether or not the compile succeeded; the resource remarks only live in the .raw file
        text = _without_remarks(err)
        if text:
            sys.stderr.write("[%s]\n%s\n" % (src, text))
        if verbose:
            print("compiled %s in %.0f s" % (src, dt), file=sys.stderr)
        if rc != 0:
            raise subprocess.CalledProcessError(rc, cmd)
    link = [hipcc_path(), "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", LIB_PATH]
    if verbose:
        print(" ".join(link), file=sys.stderr)
    subprocess.run(link, check=True)
    with open(STAMP, "w") as fh:
        fh.write(dig)
    return LIB_PATH


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
