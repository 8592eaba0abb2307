import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLD = os.path.join(ROOT, "tests", "golden")

# The sessions' mother codes are grown by progressive edge growth (0.1 - 10 s each); libqldpc keeps their edge lists in this directory
# (QLDPC_CODE_CACHE, checksummed files) so that the suite -- dozens of sessions, two daemons per loopback, 32 codes per daemon -- builds
# each code once.  tests/test_host.py checks that a cached code is the built code and that a damaged file is rebuilt.
import tempfile  # noqa: E402

os.environ.setdefault("QLDPC_CODE_CACHE", os.path.join(tempfile.gettempdir(), "qldpc_code_cache_%d" % os.getuid()))
os.makedirs(os.environ["QLDPC_CODE_CACHE"], exist_ok=True)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def q():
    """The product package (ctypes over libqldpc.so). Built on demand; never falls back to CPU."""
    import _qldpc_loader
    lib = os.path.join(ROOT, "qcrypto-ldpc_amd", "libqldpc.so")
    if not os.path.exists(lib):
        import __graft_entry__
        __graft_entry__.build()
    return _qldpc_loader.load()


@pytest.fixture(scope="session")
def O():
    """The CPU oracle (test infrastructure)."""
    from oracle import oracle
    oracle.build()
    return oracle


@pytest.fixture(scope="session")
def gold():
    return GOLD
