"""Build liblram_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

Staleness is decided by CONTENT, not by mtimes: `source_hash()` is a sha256 over every source, header and compiler flag;
it is compiled into the library (csrc/build_id.cpp -> `lram_build_id()`), `needs_build()` compares the marker found in the
.so file with the checked-out tree, and tests/conftest.py asserts the loaded library reports the same id -- a stale library
on a GPU box (the .so is git-ignored and travels with the gpurun snapshot) cannot run old kernels silently."""
from __future__ import annotations

import glob
import hashlib
import os
import re
import shutil
import subprocess
import sys

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
SOURCES = ["engine.hip", "gemm_f32.hip", "gemm_bf16x3.hip", "gemm_f16x2.hip", "gemm_f16x2p.hip", "gemm_f16x2_8p.hip", "gemm_narrow.hip", "xlstm_kernels.hip",
           "mlstm_chunk.hip", "mlstm_lazy.hip", "mlstm_front.hip", "slstm_seq.hip", "impala_cnn.hip", "misc_kernels.hip",
           "mamba_kernels.hip", "selftest.hip"]
BUILD_ID_SOURCE = "build_id.cpp"   # carries the hash of everything else; compiled on every build (a second)

# every header / include fragment of csrc counts as a dependency of every object: *.h and *.inl are globbed, so a new
# fragment cannot be forgotten here
HEADERS = sorted(os.path.basename(f) for f in glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(CSRC, "*.inl"))) \
    + [os.path.join("..", "..", "include", "lram_hip.h")]
LIB = os.path.join(CSRC, "liblram_hip.so")

# -packed-fp32-ops: no v_pk_*_f32 VALU instructions in device code.  On MI355X they return wrong results for a
# quarter wave when a co-resident wave of another dispatch issues bf16 MFMAs (csrc/selftest.hip); the flag is a
# device target feature, the host pass prints a harmless "not a recognized feature" note that is filtered below.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-comment",
         "-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops"]
OBJ_DIR = os.path.join(CSRC, "_obj")
_NOISE = "'-packed-fp32-ops' is not a recognized feature for this target"
_MARK = re.compile(rb"LRAM_BUILD_ID=([0-9a-f]{64})")


def _hipcc() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (ROCm toolchain required to build the HIP engine)")


def _digest(names) -> str:
    h = hashlib.sha256()
    for name in names:
        h.update(name.encode() + b"\0")
        with open(os.path.join(CSRC, name), "rb") as f:
            h.update(hashlib.sha256(f.read()).digest())
    h.update("\0".join(FLAGS).encode())
    return h.hexdigest()


def source_hash() -> str:
    """sha256 over SOURCES + build_id.cpp + HEADERS + FLAGS: the identity `lram_build_id()` must report."""
    return _digest(SOURCES + [BUILD_ID_SOURCE] + HEADERS)


def library_build_id(path: str = LIB):
    """The id compiled into a library file (None when the file is absent or carries no marker).  Read from the bytes of the
    file, not through dlopen: a library loaded here would stay mapped under its path and shadow the rebuilt one."""
    if not os.path.exists(path):
        return None
    with open(path, "rb") as f:
        m = _MARK.search(f.read())
    return m.group(1).decode() if m else None


def needs_build() -> bool:
    return library_build_id() != source_hash()


def _run(cmd, verbose):
    if verbose:
        print("[lram_amd.build]", " ".join(cmd), file=sys.stderr)
    proc = subprocess.run(cmd, cwd=CSRC, stderr=subprocess.PIPE, text=True)
    err = "\n".join(l for l in proc.stderr.splitlines() if _NOISE not in l)
    if err.strip():
        print(err, file=sys.stderr)
    if proc.returncode != 0:
        raise subprocess.CalledProcessError(proc.returncode, cmd)


def build(force: bool = False, verbose: bool = True) -> str:
    """One translation unit per source, compiled in parallel (objects under csrc/_obj, re-used when neither the source
    nor a header nor a flag changed: a `.hash` file beside each object holds what it was compiled from), then one link step."""
    if not force and not needs_build():
        return LIB
    from concurrent.futures import ThreadPoolExecutor
    hipcc = _hipcc()
    os.makedirs(OBJ_DIR, exist_ok=True)
    jobs, objs, stamps = [], [], []
    for src in SOURCES:
        obj = os.path.join(OBJ_DIR, os.path.splitext(src)[0] + ".o")
        objs.append(obj)
        want = _digest([src] + HEADERS)
        stamp = obj + ".hash"
        have = open(stamp).read().strip() if os.path.exists(stamp) and os.path.exists(obj) else None
        if force or have != want:
            if os.path.exists(stamp):
                os.remove(stamp)
            jobs.append([hipcc] + FLAGS + ["-c", os.path.join(CSRC, src), "-o", obj])
            stamps.append((stamp, want))
    with ThreadPoolExecutor(max_workers=max(1, min(len(jobs) or 1, os.cpu_count() or 1))) as pool:
        list(pool.map(lambda c: _run(c, verbose), jobs))
    for stamp, want in stamps:
        with open(stamp, "w") as f:
            f.write(want)
    id_obj = os.path.join(OBJ_DIR, "build_id.o")
    _run([hipcc, "-O2", "-std=c++17", "-fPIC", f'-DLRAM_BUILD_ID_HEX="{source_hash()}"', "-x", "c++", "-c",
          os.path.join(CSRC, BUILD_ID_SOURCE), "-o", id_obj], verbose)
    _run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + [id_obj], verbose)
    if library_build_id() != source_hash():
        raise RuntimeError("liblram_hip.so does not carry the id of the sources it was just built from")
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
