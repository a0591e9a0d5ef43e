"""Build recipe of libla3d.so — importable by file path without importing the package (which loads
the library and fails loudly when it is missing)."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SRC = os.path.join(HERE, "csrc", "la3d.hip")
LIB = os.environ.get("LA3D_LIB") or os.path.join(HERE, "lib", "libla3d.so")  # LA3D_LIB: experiment builds only
INCLUDE = os.path.join(ROOT, "include")

HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared"]


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile csrc/la3d.hip for gfx950 into lib/libla3d.so (in-tree, so it travels with the repo)."""
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    hdr = os.path.join(INCLUDE, "la3d.h")
    if not force and os.path.exists(LIB) and os.path.getmtime(LIB) >= max(os.path.getmtime(SRC), os.path.getmtime(hdr)):
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, *HIPCC_FLAGS, "-I", INCLUDE, SRC, "-o", LIB]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    return LIB


