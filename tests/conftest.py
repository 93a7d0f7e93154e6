import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `pytest -m gpu` on the GPU box)")


@pytest.fixture(scope="session")
def lib():
    """libmaelsim.so, built in-tree if missing/stale."""
    from maelstrom_amd import build, _abi
    build.build(verbose=False)
    return _abi.load()


_guard_seen = [0]


@pytest.fixture(autouse=True)
def _guard_after_every_test(request):
    """Under MSIM_GUARD: the damaged-byte count after every test, so that a damaged zone names the test that ran over it."""
    yield
    if not os.environ.get("MSIM_GUARD"):
        return
    import ctypes as C
    from maelstrom_amd import _abi
    if _abi._lib is None:
        return
    _abi._lib.msim_guard_check.restype = C.c_ulonglong
    damaged = int(_abi._lib.msim_guard_check(None))
    if damaged != _guard_seen[0]:
        new, _guard_seen[0] = damaged - _guard_seen[0], damaged
        pytest.fail(f"[msim guard] {new} byte(s) outside a device slab were overwritten during {request.node.nodeid}", pytrace=False)


def pytest_sessionfinish(session, exitstatus):
    """Under MSIM_GUARD (csrc/guard.cpp: fenced device slabs) the run ends with the count of bytes written outside any slab;
    tools/guard_sweep.sh reads the line.  A damaged byte fails the session."""
    if not os.environ.get("MSIM_GUARD"):
        return
    import ctypes as C
    from maelstrom_amd import _abi
    if _abi._lib is None:
        return
    lib = _abi._lib
    lib.msim_guard_check.restype = C.c_ulonglong
    n_allocs = C.c_ulonglong(0)
    damaged = int(lib.msim_guard_check(C.byref(n_allocs)))
    print(f"\n[msim guard] {damaged} damaged byte(s) around {n_allocs.value} slabs (MSIM_GUARD={os.environ['MSIM_GUARD']})")
    if damaged:
        session.exitstatus = 1
