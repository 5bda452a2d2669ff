import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def hip_lib():
    """Build (if stale) and load the in-tree HIP library; GPU tests fail loudly when it is missing."""
    from lram_amd import build, engine
    build.build(force=False, verbose=False)
    lib = engine.load_library()
    # the library this process runs was built from the checked-out sources (lram_build_id = sha256 over sources + headers
    # + flags, compiled in): a stale .so that travelled with a snapshot fails here instead of running old kernels
    if not os.environ.get("LRAM_LIB_VARIANT"):
        assert lib.lram_build_id().decode() == build.source_hash(), "liblram_hip.so was not built from these sources"
    return lib
