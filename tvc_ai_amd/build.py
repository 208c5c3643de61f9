"""Builds libtvc_hip.so (hand-written HIP kernels for gfx950 + the C ABI of include/tvc_native.h).

hipcc cross-compiles without a GPU; the .so is kept in-tree (git-ignored) so it travels to the GPU box.
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIB = os.environ.get("TVC_HIP_LIB") or os.path.join(CSRC, "libtvc_hip.so")
SOURCES = ["tvc_env.hip", "tvc_sac.hip", "tvc_replay.hip"]
ARCH = "gfx950"


def _hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", shutil.which("hipcc")):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (needed to build libtvc_hip.so)")


def sources():
    return [os.path.join(CSRC, s) for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]


def sources_sha256():
    """sha256 over the library's sources (csrc/*.hip, csrc/*.h, include/tvc_native.h, by name): stored in the committed profile
    summaries so that bench.py can tell when a figure it quotes from them was taken at other kernels than the ones it runs"""
    import hashlib
    h = hashlib.sha256()
    files = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".h")))
    files.append(os.path.join(ROOT, "include", "tvc_native.h"))
    for f in files:
        h.update(os.path.basename(f).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def needs_build():
    if os.environ.get("TVC_HIP_LIB"):
        return False  # an explicitly chosen prebuilt variant (kernel A/B experiments)
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = sources() + [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    deps.append(os.path.join(ROOT, "include", "tvc_native.h"))
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False, extra_flags=()):
    if not force and not needs_build():
        return LIB
    cmd = [_hipcc(), "-O3", "-std=c++17", f"--offload-arch={ARCH}", "-fPIC", "-shared",
           "-fno-gpu-rdc", "-ffp-contract=fast", "-fno-slp-vectorize",  # SLP-packed v_pk_* f32 cost 15 % here (A/B in profiles/)
           "-Wall", "-Wno-unused-function",
           "-I", os.path.join(ROOT, "include"), "-I", CSRC, *extra_flags, "-o", LIB, *sources()]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
