"""Build libimcoal_fwd.so (the C-ABI HIP library) in-tree for gfx950.

hipcc cross-compiles without a GPU, so this runs in the build container and the resulting .so
travels to the GPU box with the repository snapshot.
"""
import os
import shutil
import subprocess

PKG = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(PKG, "csrc", "imcoal_fwd.hip")
HDR = os.path.join(os.path.dirname(PKG), "include", "imcoal_fwd.h")
DEPS = [SRC, HDR, os.path.join(os.path.dirname(PKG), "include", "imcoal_model.h")] + [os.path.join(PKG, "csrc", f) for f in
                     ("kernels_plain.hpp", "kernels_zip.hpp", "kernels_stitch.hpp", "kernels_big.hpp", "kernels_zip2.hpp", "kernels_zip3.hpp", "kernels_zip4.hpp", "pair_dict.hpp", "obs_io.hpp",
                      "model_host.hpp")]
LIB = os.path.join(PKG, "libimcoal_fwd.so")


def hipcc_path():
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: cannot build libimcoal_fwd.so")


STAMP = os.path.join(PKG, ".libimcoal_fwd.sources.sha256")
last_build = {"compiled": None, "seconds": 0.0, "reason": ""}   # what the last build_library() call did


def sources_digest():
    import hashlib
    h = hashlib.sha256()
    for f in DEPS:
        with open(f, "rb") as fh:
            h.update(os.path.basename(f).encode() + b"\0" + fh.read())
    return h.hexdigest()


def needs_build():
    """The library is rebuilt unless it exists AND was built from exactly these sources (content hash, not mtime: a
    fresh checkout or a copied tree gives every file the same time)."""
    if not os.path.exists(LIB):
        return "library missing"
    try:
        with open(STAMP) as fh:
            if fh.read().strip() == sources_digest():
                return ""
    except OSError:
        return "no source stamp next to the library"
    return "sources changed since the library was built"


def build_library(force=False, verbose=False):
    import time
    reason = "forced" if force or os.environ.get("IMC_FORCE_BUILD") else needs_build()
    if not reason:
        last_build.update(compiled=False, seconds=0.0, reason="prebuilt library matches the sources (sha256)")
        return LIB
    cmd = [hipcc_path(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC",
           "-o", LIB, SRC]
    if verbose:
        cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
    t0 = time.time()
    subprocess.run(cmd, check=True)
    with open(STAMP, "w") as fh:
        fh.write(sources_digest() + "\n")
    last_build.update(compiled=True, seconds=time.time() - t0, reason=reason)
    return LIB


if __name__ == "__main__":
    import sys
    print(build_library(force="--force" in sys.argv, verbose="--verbose" in sys.argv))
    print("hipcc ran: %(compiled)s (%(reason)s, %(seconds).1f s)" % last_build)
