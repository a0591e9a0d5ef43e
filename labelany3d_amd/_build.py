"""Build recipe of libla3d.so — importable by file path without importing the package (which loads
the library and fails loudly when it is missing)."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SRC = os.path.join(HERE, "csrc", "la3d.hip")
SOURCES = [SRC, os.path.join(HERE, "csrc", "la3d_split.hip")]
HEADERS = [os.path.join(HERE, "csrc", "la3d_device.hpp"), os.path.join(ROOT, "include", "la3d.h")]
LIB = os.environ.get("LA3D_LIB") or os.path.join(HERE, "lib", "libla3d.so")  # LA3D_LIB: experiment builds only
INCLUDE = os.path.join(ROOT, "include")

HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared"]


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile csrc/la3d.hip for gfx950 into lib/libla3d.so (in-tree, so it travels with the repo)."""
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    newest = max(os.path.getmtime(f) for f in SOURCES + HEADERS)
    if not force and os.path.exists(LIB) and os.path.getmtime(LIB) >= newest:
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, *HIPCC_FLAGS, "-I", INCLUDE, *SOURCES, "-o", LIB]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    return LIB


