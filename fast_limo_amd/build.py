"""Build driver: compiles the in-tree native libraries with hipcc / g++ through csrc/Makefile.

``build_all()`` is what ``__graft_entry__.build()`` calls.  hipcc cross-compiles gfx950 code objects
without a GPU, so this works in the CPU-only container; the resulting ``.so`` files stay in-tree
(``fast_limo_amd/lib*.so``) and travel to the GPU box with the repository snapshot.
"""
from __future__ import annotations

import os
import subprocess

_PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(_PKG)


_STAMP = os.path.join(_PKG, ".build_stamp")


def sources_hash() -> str:
    """Hash of every source the two libraries are built from (content, not mtime: a repository snapshot keeps neither order
    nor times)."""
    import hashlib
    h = hashlib.sha256()
    roots = [os.path.join(_PKG, "csrc"), os.path.join(ROOT, "include")]
    files = []
    for r in roots:
        for d, dn, fn in os.walk(r):
            dn[:] = [x for x in dn if x != "build"]
            files += [os.path.join(d, f) for f in fn if f.endswith((".hip", ".h", ".hpp", ".cpp")) or f == "Makefile"]
    for p in sorted(files):
        h.update(os.path.relpath(p, ROOT).encode())
        with open(p, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def is_stale() -> bool:
    """True when a library is missing or was built from other sources than the ones in the tree."""
    for so in ("libflimo_hip.so", "libfast_limo.so"):
        if not os.path.exists(os.path.join(_PKG, so)):
            return True
    try:
        return open(_STAMP).read().strip() != sources_hash()
    except OSError:
        return True


def build_native(jobs: int = 4, force: bool = False) -> None:
    cmd = ["make", "-C", os.path.join(_PKG, "csrc"), f"-j{jobs}"]
    if force:
        cmd.append("-B")
    subprocess.check_call(cmd)
    for so in ("libflimo_hip.so", "libfast_limo.so"):
        if not os.path.exists(os.path.join(_PKG, so)):
            raise RuntimeError(f"{so} was not produced")
    with open(_STAMP, "w") as fh:
        fh.write(sources_hash() + "\n")


def build_tools() -> None:
    """GPU-side checker binaries (device float math vs host IEEE)."""
    src = os.path.join(ROOT, "tools", "devmath_check.hip")
    out = os.path.join(ROOT, "tools", "devmath_check")
    if (not os.path.exists(out)) or os.path.getmtime(out) < max(
            os.path.getmtime(src), os.path.getmtime(os.path.join(_PKG, "csrc", "hip", "flimo_math.h"))):
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off",
                               "-I", os.path.join(_PKG, "csrc", "hip"), src, "-o", out])


def build_all(jobs: int = 4) -> None:
    build_native(jobs)
    build_tools()
