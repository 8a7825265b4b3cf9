import os
import sys

import pytest

# The HIP runtime reports a queue error (a faulting kernel) only through its log; at the default level the process
# just aborts.  Level 1 = errors only.  Must be set before the runtime is loaded (torch / the library import).
os.environ.setdefault("AMD_LOG_LEVEL", "1")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if os.environ.get("ICIKT_SEGV_BT"):   # development aid: native backtrace on SIGSEGV (tools/dbg/segv_bt.c)
    import ctypes
    ctypes.CDLL(os.environ["ICIKT_SEGV_BT"])
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="session")
def expected(golden_dir):
    import json
    with open(os.path.join(golden_dir, "expected.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def hip_ctx():
    from icikendalltau_amd import _lib
    ctx = _lib.Context(0)
    yield ctx
    ctx.close()


@pytest.fixture
def plan_ctx(hip_ctx):
    """The session context for tests that override the pair kernel's launch plan (icikt_debug_set_plan);
    the library's own choices are restored afterwards."""
    yield hip_ctx
    hip_ctx.debug_set_plan(None)
