"""Build libp2w_gfx950.so in-tree with hipcc (cross-compiles without a GPU).

    python -m pointstowood_amd.build [--force]

Flags that matter: ``--offload-arch=gfx950`` (MI355X only, no other targets),
``-ffp-contract=off`` (the geometry kernels need individually rounded fp32 ops for
bit-exact neighbour sets; kernels that want an FMA call fmaf explicitly).
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
INCLUDE = os.path.join(os.path.dirname(HERE), "include")
LIB = os.path.join(HERE, "libp2w_gfx950.so")
SOURCES = ["p2w_geom.hip", "p2w_feat.hip", "p2w_feat_h1.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
         *os.environ.get("P2W_EXTRA_CFLAGS", "").split(),      # diagnostic builds, e.g. -DP2W_SLAB_PROFILE
         "-I", INCLUDE]


def _hipcc() -> str:
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def source_hash() -> str:
    """sha256 over the kernel sources, the ABI header and the compile flags."""
    import hashlib
    h = hashlib.sha256(" ".join(FLAGS[:-1]).encode())
    for path in sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC)) + [os.path.join(INCLUDE, "p2w.h")]:
        h.update(open(path, "rb").read())
    return h.hexdigest()


def _stale() -> bool:
    if not os.path.exists(LIB) or not os.path.exists(LIB + ".srchash"):
        return True
    return open(LIB + ".srchash").read().strip() != source_hash()


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile and link in-tree.  Safe under several processes (every rank of a ``torch.distributed.run`` launch calls
    this through ``_lib.lib()`` on a fresh checkout): one process builds under an exclusive file lock, the others wait
    for the lock and then find the library fresh; objects and the link output use per-process names and are moved into
    place atomically, so nobody ever loads a half-written file."""
    import fcntl
    if not force and not _stale():
        return LIB
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    with open(os.path.join(objdir, ".lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not force and not _stale():      # somebody else built it while we waited
                return LIB
            return _build_locked(objdir, verbose)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


def _build_locked(objdir: str, verbose: bool) -> str:
    hipcc = _hipcc()
    tag = f".{os.getpid()}"

    def cc(src):
        obj = os.path.join(objdir, src.replace(".hip", tag + ".o"))
        cmd = [hipcc, *FLAGS, "-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{r.stderr}")
        return obj

    objs = []
    try:
        with ThreadPoolExecutor(max_workers=len(SOURCES)) as ex:
            objs = list(ex.map(cc, SOURCES))
        tmp = LIB + tag + ".tmp"
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", f"-Wl,--version-script={os.path.join(CSRC, 'p2w_exports.map')}",
               "-o", tmp, *objs]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stderr}")
        digest = source_hash()
        os.replace(tmp, LIB)
        with open(LIB + ".srchash" + tag, "w") as f:
            f.write(digest)
        os.replace(LIB + ".srchash" + tag, LIB + ".srchash")
    finally:
        for o in objs:
            if os.path.exists(o):
                os.remove(o)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
