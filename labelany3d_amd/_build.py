"""Build recipe of libla3d.so — importable by file path without importing the package (which loads
the library and fails loudly when it is missing)."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
# one translation unit per engine (compiled side by side), the C-ABI + dispatcher, the other kernels, the host-side JSON writer (last:
# it carries the build identity)
SOURCES = [os.path.join(CSRC, f) for f in ("la3d_instance.hip", "la3d_band.hip", "la3d_rows.hip", "la3d_split.hip", "la3d_points.hip", "la3d_masks.hip", "la3d_consumers.hip", "la3d.hip",
                                           "la3d_json.cpp")]
HEADERS = [os.path.join(CSRC, f) for f in ("la3d_device.hpp", "la3d_walks.hpp", "la3d_stages.hpp", "la3d_engines.hpp", "la3d_poly.hpp")] + \
          [os.path.join(ROOT, "include", "la3d.h")]
LIB = os.environ.get("LA3D_LIB") or os.path.join(HERE, "lib", "libla3d.so")  # LA3D_LIB: experiment builds only
INCLUDE = os.path.join(ROOT, "include")

HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared"]


def source_sha256() -> str:
    """Hash of everything libla3d.so is compiled from (sources and headers, fixed order, names included)."""
    import hashlib
    h = hashlib.sha256()
    for f in SOURCES + HEADERS:
        h.update(os.path.basename(f).encode() + b"\0")
        h.update(open(f, "rb").read())
    return h.hexdigest()


def compile_cmd_sha256() -> str:
    """Hash of the compile command: the flags and the compiler's own version line."""
    import hashlib
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    try:
        ver = subprocess.run([hipcc, "--version"], capture_output=True, text=True).stdout.splitlines()[0]
    except Exception:  # noqa: BLE001
        ver = "unknown"
    return hashlib.sha256((" ".join(HIPCC_FLAGS) + "|" + ver).encode()).hexdigest()


def embedded_info(lib_path: str = LIB):
    """(sources hash, compile-command hash) the library at lib_path was built from - read from the file's bytes (the string
    la3d_build_info() returns), without loading it - or None for a library that carries none."""
    import re
    try:
        data = open(lib_path, "rb").read()
    except OSError:
        return None
    m = re.search(rb"LA3D_BUILD_INFO:([0-9a-f]{64}):([0-9a-f]{64})", data)
    return (m.group(1).decode(), m.group(2).decode()) if m else None


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile the sources of csrc/ for gfx950 into lib/libla3d.so (in-tree, so it travels with the repo): one object per
    translation unit, compiled side by side, then one link.  The hash of the sources and of the compile command is compiled INTO
    the library (la3d_build_info); an existing library is reused only when the hash it carries is the hash of the tree's sources
    (not by file time: a library that travelled with a snapshot proves what it was built from)."""
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    src_hash = source_sha256()
    if not force and os.path.exists(LIB):
        have = embedded_info(LIB)
        if have is not None and have[0] == src_hash:
            if verbose:
                print(f"libla3d.so is current: built from sources sha256 {src_hash[:16]}.. (compile command {have[1][:16]}..)")
            return LIB
    import tempfile
    from concurrent.futures import ThreadPoolExecutor

    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd_hash = compile_cmd_sha256()
    flags = [f for f in HIPCC_FLAGS if f != "-shared"]
    with tempfile.TemporaryDirectory(prefix="la3d_build_") as tmp:
        objs = [os.path.join(tmp, os.path.splitext(os.path.basename(src))[0] + ".o") for src in SOURCES]
        cmds = [[hipcc, *flags, "-I", INCLUDE, "-c", src, "-o", obj] for src, obj in zip(SOURCES, objs)]
        # the host-only translation unit carries the identity string
        cmds[-1][1:1] = [f'-DLA3D_BUILD_SOURCES="{src_hash}"', f'-DLA3D_BUILD_CMD="{cmd_hash}"']
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
    if verbose:
        print(f"libla3d.so built from sources sha256 {src_hash[:16]}.. with compile command {cmd_hash[:16]}..")
    assert embedded_info(LIB) == (src_hash, cmd_hash), "the library does not carry the identity it was built with"
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


