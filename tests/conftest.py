import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


# One ROCm stack per test process: torch carries its own libamdhip64 / librccl, and a process that maps the system copies first
# (through libfrostdb_amd.so / fdb_comm's dlopen) and torch's afterwards ends up with two RCCLs and handles that belong to the
# wrong runtime ("invalid resource handle" in the torch.distributed tests). bench.py imports torch first for the same reason; a Go
# host has no torch and simply uses the system libraries.
try:
    import torch  # noqa: F401
except ImportError:  # pragma: no cover
    pass


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _build_oracle():
    import oracle
    oracle.build()


@pytest.fixture(scope="module", autouse=True)
def _no_device_allocation_outlives_a_test_module():
    """≙ memory.CheckedAllocator.AssertSize(0) after every logictest file (logictest/logic_test.go:169-177): once a test module is
    done — every plan closed, every resident batch and result record released — the library owns no device block and no pinned
    result block (fdb_live_allocations; idle blocks of the caching allocator are not counted)."""
    yield
    import gc
    from frostdb_amd import physicalplan as pp
    if pp._lib is None:
        return
    gc.collect()
    live = pp.live_allocations()
    assert live == {"device_blocks": 0, "device_bytes": 0, "pinned_blocks": 0}, f"device allocations leaked by this test module: {live}"
