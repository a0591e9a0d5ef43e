"""Build recipe of libla3d.so — importable by file path without importing the package (which loads
the library and fails loudly when it is missing)."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SRC = os.path.join(HERE, "csrc", "la3d.hip")
SOURCES = [SRC, os.path.join(HERE, "csrc", "la3d_split.hip"), os.path.join(HERE, "csrc", "la3d_aux.hip"), os.path.join(HERE, "csrc", "la3d_json.cpp")]
HEADERS = [os.path.join(HERE, "csrc", "la3d_device.hpp"), os.path.join(HERE, "csrc", "la3d_poly.hpp"), os.path.join(ROOT, "include", "la3d.h")]
LIB = os.environ.get("LA3D_LIB") or os.path.join(HERE, "lib", "libla3d.so")  # LA3D_LIB: experiment builds only
INCLUDE = os.path.join(ROOT, "include")

HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared"]


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile the sources of csrc/ for gfx950 into lib/libla3d.so (in-tree, so it travels with the repo): one object per
    translation unit, compiled side by side, then one link."""
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    newest = max(os.path.getmtime(f) for f in SOURCES + HEADERS)
    if not force and os.path.exists(LIB) and os.path.getmtime(LIB) >= newest:
        return LIB
    import tempfile
    from concurrent.futures import ThreadPoolExecutor

    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    flags = [f for f in HIPCC_FLAGS if f != "-shared"]
    with tempfile.TemporaryDirectory(prefix="la3d_build_") as tmp:
        objs = [os.path.join(tmp, os.path.splitext(os.path.basename(src))[0] + ".o") for src in SOURCES]
        cmds = [[hipcc, *flags, "-I", INCLUDE, "-c", src, "-o", obj] for src, obj in zip(SOURCES, objs)]
        if verbose:
            for c in cmds:
                print(" ".join(c))
        with ThreadPoolExecutor(max_workers=len(cmds)) as ex:
            for r in ex.map(lambda c: subprocess.run(c, capture_output=True, text=True), cmds):
                if r.returncode != 0:
                    raise RuntimeError("hipcc failed:\n" + r.stderr[-4000:])
        link = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", LIB]
        if verbose:
            print(" ".join(link))
        subprocess.run(link, check=True)
    return LIB


EXAMPLE_SRC = os.path.join(ROOT, "examples", "fit_from_c.cpp")
EXAMPLE_BIN = os.path.join(HERE, "lib", "fit_from_c")


def build_example(force: bool = False, verbose: bool = False) -> str:
    """The plain host program that drives the C-ABI without Python (examples/fit_from_c.cpp), linked against the in-tree
    library; tests/test_gpu_cabi.py runs it on the GPU box."""
    lib = build(force=False, verbose=verbose)
    if not force and os.path.exists(EXAMPLE_BIN) and os.path.getmtime(EXAMPLE_BIN) >= max(os.path.getmtime(EXAMPLE_SRC),
                                                                                         os.path.getmtime(lib)):
        return EXAMPLE_BIN
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "-O2", "-std=c++17", "-I", INCLUDE, EXAMPLE_SRC, "-L", os.path.dirname(lib), "-lla3d", "-Wl,-rpath,$ORIGIN",
           "-o", EXAMPLE_BIN]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    return EXAMPLE_BIN


