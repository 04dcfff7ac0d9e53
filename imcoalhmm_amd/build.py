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
DEPS = [SRC, HDR] + [os.path.join(PKG, "csrc", f) for f in
                     ("kernels_plain.hpp", "kernels_zip.hpp", "kernels_stitch.hpp", "kernels_big.hpp", "kernels_zip2.hpp", "kernels_zip3.hpp", "pair_dict.hpp", "obs_io.hpp")]
LIB = os.path.join(PKG, "libimcoal_fwd.so")


def hipcc_path():
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: cannot build libimcoal_fwd.so")


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(f) > t for f in DEPS)


def build_library(force=False, verbose=False):
    if not force and not needs_build():
        return LIB
    cmd = [hipcc_path(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC",
           "-o", LIB, SRC]
    if verbose:
        cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
    subprocess.run(cmd, check=True)
    return LIB


if __name__ == "__main__":
    import sys
    print(build_library(force="--force" in sys.argv, verbose="--verbose" in sys.argv))
