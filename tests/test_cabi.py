"""The C-ABI library builds for gfx950, loads without a GPU, exports every symbol include/lram_hip.h declares,
and the ctypes mirror of `lram_config` has the C layout.  No compute calls (no GPU here)."""
import ctypes
import os
import re
import subprocess

import pytest

from lram_amd import engine

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "lram_hip.h")


def _declared_symbols():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(lram_[a-z0-9_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported_and_bound(hip_lib):
    declared = _declared_symbols()
    assert len(declared) >= 20
    assert sorted(engine._SYMBOLS) == declared, "ctypes table and header disagree"
    for name in declared:
        assert getattr(hip_lib, name) is not None
    assert hip_lib.lram_abi_version() == engine.LRAM_ABI_VERSION
    assert hip_lib.lram_last_error() == b""


def test_config_struct_layout_matches_c(tmp_path):
    fields = [f[0] for f in engine.LramConfig._fields_]
    prog = ['#include <stdio.h>', '#include <stddef.h>', f'#include "{HEADER}"', "int main(void){",
            'printf("%zu\\n", sizeof(lram_config));']
    prog += [f'printf("%zu\\n", offsetof(lram_config, {f}));' for f in fields]
    prog += ["return 0;}"]
    src = tmp_path / "layout.c"
    src.write_text("\n".join(prog))
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", str(src), "-o", str(exe)], check=True)
    nums = [int(x) for x in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()]
    assert nums[0] == ctypes.sizeof(engine.LramConfig)
    for f, off in zip(fields, nums[1:]):
        assert getattr(engine.LramConfig, f).offset == off, f


def test_header_is_plain_c_and_error_path_without_gpu(hip_lib):
    # creating an engine needs a device: the call must fail with a message, not crash
    from lram_amd import preset
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    cfg = engine.make_config(preset("xlstm_tiny"))
    h = ctypes.c_void_p()
    rc = hip_lib.lram_create(ctypes.byref(cfg), 0, ctypes.byref(h))
    assert rc != 0 and len(hip_lib.lram_last_error()) > 0
    cfg.abi_version = 99
    assert hip_lib.lram_create(ctypes.byref(cfg), 0, ctypes.byref(h)) != 0
    assert b"abi_version" in hip_lib.lram_last_error()


def test_build_identity_follows_the_sources(hip_lib, monkeypatch):
    """lram_build_id() = sha256 over sources + headers + flags; build.needs_build() compares the marker inside the .so file
    with the tree, so any edit (here: a flag) makes the library on disk stale -- mtimes play no part."""
    from lram_amd import build
    want = build.source_hash()
    assert len(want) == 64 and hip_lib.lram_build_id().decode() == want == build.library_build_id()
    assert not build.needs_build()
    monkeypatch.setattr(build, "FLAGS", build.FLAGS + ["-DX=1"])
    assert build.source_hash() != want and build.needs_build()
    assert build.library_build_id(os.path.join(ROOT, "no_such.so")) is None
