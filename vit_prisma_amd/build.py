"""Builds libpvnative.so (hand-written HIP for gfx950) in-tree with hipcc.

    python -m vit_prisma_amd.build            # incremental
    python -m vit_prisma_amd.build --force

hipcc cross-compiles without a GPU, so this runs in the build container; the resulting .so sits
next to this file (git-ignored, but it travels to the GPU box with the tree).
"""
from __future__ import annotations

import concurrent.futures as cf
import os
import shutil
import subprocess
import sys
from typing import List

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "_obj")
LIB = os.path.join(HERE, "libpvnative.so")
ARCH = "gfx950"
FLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-Wall", "-Wno-unused-function",
         "-ffp-contract=off"]


def _hipcc() -> str:
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found (ROCm toolchain required to build the MI355X kernels)")


def sources() -> List[str]:
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))


def source_id() -> str:
    """sha256 over the kernel sources (csrc/*.hip, csrc/*.hpp, include/pv_native.h: names and contents, sorted) -- what
    ``pv_build_id()`` of a library built from exactly these files returns."""
    import hashlib
    h = hashlib.sha256()
    files = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith((".hip", ".hpp"))]
    files.append(os.path.join(os.path.dirname(HERE), "include", "pv_native.h"))
    for path in files:
        h.update(os.path.basename(path).encode() + b"\0")
        with open(path, "rb") as f:
            h.update(f.read())
        h.update(b"\0")
    return h.hexdigest()[:32]


def _write_build_id() -> None:
    inc = os.path.join(OBJ, "build_id.inc")
    text = f'#define PV_BUILD_ID "{source_id()}"\n'
    old = None
    if os.path.exists(inc):
        with open(inc) as f:
            old = f.read()
    if old != text:
        with open(inc, "w") as f:
            f.write(text)


def _deps_mtime() -> float:
    paths = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hpp", ".h"))]
    paths.append(os.path.join(os.path.dirname(HERE), "include", "pv_native.h"))
    return max(os.path.getmtime(p) for p in paths)


def _extra_dep_mtime(src: str) -> float:
    if src == "build_id.hip":
        return os.path.getmtime(os.path.join(OBJ, "build_id.inc"))
    return 0.0


def _compile(src: str, force: bool) -> str:
    obj = os.path.join(OBJ, src.replace(".hip", ".o"))
    spath = os.path.join(CSRC, src)
    if (not force and os.path.exists(obj) and os.path.getmtime(obj) > os.path.getmtime(spath)
            and os.path.getmtime(obj) > _deps_mtime() and os.path.getmtime(obj) > _extra_dep_mtime(src)):
        return obj
    cmd = [_hipcc(), *FLAGS, "-c", spath, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}\n{r.stderr}")
    if r.stderr.strip():
        sys.stderr.write(r.stderr)
    return obj


def build(force: bool = False, verbose: bool = True) -> str:
    os.makedirs(OBJ, exist_ok=True)
    _write_build_id()
    srcs = sources()
    with cf.ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
        objs = list(ex.map(lambda s: _compile(s, force), srcs))
    if force or not os.path.exists(LIB) or any(os.path.getmtime(o) > os.path.getmtime(LIB) for o in objs):
        cmd = [_hipcc(), "-shared", "-fPIC", f"--offload-arch={ARCH}", *objs, "-o", LIB]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
        # a kernel template whose host-side instantiation failed silently leaves an undefined stub behind: load the library
        # with immediate binding (in a child process: this one may hold an older copy) before calling it built
        r = subprocess.run([sys.executable, "-c", f"import ctypes, os; ctypes.CDLL({LIB!r}, mode=os.RTLD_NOW)"],
                           capture_output=True, text=True)
        if r.returncode != 0:
            os.remove(LIB)
            raise RuntimeError(f"the linked library does not load:\n{r.stderr}")
    if verbose:
        print(f"[vit_prisma_amd.build] {LIB} ({os.path.getsize(LIB) // 1024} kB) from {len(srcs)} HIP sources")
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
