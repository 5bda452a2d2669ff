"""Build liblram_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
SOURCES = ["engine.hip", "gemm_f32.hip", "gemm_bf16x3.hip", "gemm_f16x2.hip", "gemm_f16x2p.hip", "xlstm_kernels.hip", "mlstm_chunk.hip", "mlstm_lazy.hip", "mlstm_front.hip", "slstm_seq.hip", "impala_cnn.hip", "misc_kernels.hip", "mamba_kernels.hip",
           "selftest.hip"]
import glob

# every header / include fragment of csrc counts as a dependency of every object (a stale .so on the GPU box would run
# old kernels without a word): *.h and *.inl are globbed, so a new fragment cannot be forgotten here
HEADERS = sorted(os.path.basename(f) for f in glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(CSRC, "*.inl"))) \
    + [os.path.join("..", "..", "include", "lram_hip.h")]
LIB = os.path.join(CSRC, "liblram_hip.so")


def _hipcc() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (ROCm toolchain required to build the HIP engine)")


def needs_build() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS]
    return any(os.path.getmtime(d) > t for d in deps)


# -packed-fp32-ops: no v_pk_*_f32 VALU instructions in device code.  On MI355X they return wrong results for a
# quarter wave when a co-resident wave of another dispatch issues bf16 MFMAs (csrc/selftest.hip); the flag is a
# device target feature, the host pass prints a harmless "not a recognized feature" note that is filtered below.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-comment",
         "-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops"]
OBJ_DIR = os.path.join(CSRC, "_obj")
_NOISE = "'-packed-fp32-ops' is not a recognized feature for this target"


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
    nor a header changed), then one link step."""
    if not force and not needs_build():
        return LIB
    from concurrent.futures import ThreadPoolExecutor
    hipcc = _hipcc()
    os.makedirs(OBJ_DIR, exist_ok=True)
    hdr_time = max(os.path.getmtime(os.path.join(CSRC, h)) for h in HEADERS)
    jobs, objs = [], []
    for src in SOURCES:
        obj = os.path.join(OBJ_DIR, os.path.splitext(src)[0] + ".o")
        objs.append(obj)
        src_path = os.path.join(CSRC, src)
        stale = force or not os.path.exists(obj) or os.path.getmtime(obj) < max(hdr_time, os.path.getmtime(src_path))
        if stale:
            jobs.append([hipcc] + FLAGS + ["-c", src_path, "-o", obj])
    with ThreadPoolExecutor(max_workers=max(1, min(len(jobs) or 1, os.cpu_count() or 1))) as pool:
        list(pool.map(lambda c: _run(c, verbose), jobs))
    _run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs, verbose)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
