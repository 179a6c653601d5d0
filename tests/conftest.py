import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # built artefacts are kept out of git; a fresh checkout builds them once (hipcc cross-compiles without a GPU)
    needed = [os.path.join(ROOT, "doppler_amd", "lib", "libdoppler_hip.so"), os.path.join(ROOT, "doppler_amd", "bin", "doppler"),
              os.path.join(ROOT, "oracle", "liboracle.so"), os.path.join(ROOT, "oracle", "check_sincosf"),
              os.path.join(ROOT, "tests", "cpp", "test_dsp")]
    if not all(os.path.exists(p) for p in needed):
        import __graft_entry__
        __graft_entry__.build()


@pytest.fixture(scope="session")
def orc():
    """The CPU oracle (test infrastructure; oracle/)."""
    from oracle import oracle
    return oracle


@pytest.fixture(scope="session")
def ctx():
    """A GPU context through the C ABI. Only gpu-marked tests may request it."""
    import doppler_amd
    c = doppler_amd.Context(int(os.environ.get("LOCAL_RANK", "0")))
    yield c
    c.close()
