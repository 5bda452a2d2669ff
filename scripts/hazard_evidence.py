#!/usr/bin/env python3
"""Evidence for the -packed-fp32-ops build flag (csrc/selftest.hip), kept per round under profiles/.

Two builds of the SAME sources with the SAME toolchain: the product library (packed-fp32 VALU instructions removed
from the device target features) and a variant with the compiler's default feature set.  Each runs
lram_selftest_concurrent -- mlstm_pre_kernel on one stream, a bf16x3 projection on another, outputs compared
bit-for-bit with the solo launch -- in a process of its own.

    python scripts/hazard_evidence.py --build      # here (no GPU): cross-compile the variant into csrc/_hazard/
    python scripts/hazard_evidence.py --run OUT    # on the GPU box: run both, write OUT (json)
"""
import ctypes
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from lram_amd import build  # noqa: E402

VARIANT = os.path.join(build.CSRC, "_hazard", "liblram_hip_packed.so")


def build_variant():
    os.makedirs(os.path.dirname(VARIANT), exist_ok=True)
    flags = [f for f in build.FLAGS if f not in ("-Xclang", "-target-feature", "-packed-fp32-ops")]
    cmd = [build._hipcc()] + flags + ["-shared", "-o", VARIANT] + [os.path.join(build.CSRC, s) for s in build.SOURCES]
    subprocess.run(cmd, check=True, cwd=build.CSRC)
    return VARIANT


def one(lib_path, iters):
    lib = ctypes.CDLL(lib_path)
    d = ctypes.c_int64(-1)
    rc = lib.lram_selftest_concurrent(iters, ctypes.byref(d))
    print(json.dumps({"rc": rc, "n_diff": d.value}))


if __name__ == "__main__":
    if sys.argv[1] == "--build":
        print(build_variant())
    elif sys.argv[1] == "--one":
        one(sys.argv[2], int(sys.argv[3]))
    elif sys.argv[1] == "--run":
        iters = 40
        res = {"what": "lram_selftest_concurrent(%d): elements of mlstm_pre_kernel's outputs that differ from the solo launch "
                       "when a bf16x3 GEMM of another stream runs beside it" % iters,
               "toolchain": subprocess.run(["/opt/rocm/bin/hipcc", "--version"], capture_output=True, text=True).stdout.splitlines()[0]}
        for tag, path in (("product_build_without_packed_fp32", build.LIB), ("variant_default_target_features", VARIANT)):
            runs = []
            for _ in range(3):
                p = subprocess.run([sys.executable, os.path.abspath(__file__), "--one", path, str(iters)], capture_output=True,
                                   text=True, timeout=600)
                runs.append(json.loads(p.stdout.strip().splitlines()[-1]) if p.returncode == 0 and p.stdout.strip()
                            else {"rc": p.returncode, "err": p.stderr[-300:]})
            res[tag] = {"library": os.path.relpath(path, ROOT), "runs": runs}
        with open(sys.argv[2], "w") as fh:
            json.dump(res, fh, indent=1)
        print(json.dumps(res, indent=1))
