import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def dev():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU visible")
    return torch.device("cuda:0")


def install_guard_allocator(mode):
    """Route EVERY device allocation of this process through tests/guard/libdpig_guard_alloc.so (one virtual-memory reservation per
    tensor with unmapped pages on both sides; the tensor ends on the last mapped 16 bytes in mode 'hi', starts on the first in 'lo').
    Must run before the first device allocation.  hipGraph capture pools are not available under a pluggable allocator."""
    import torch
    import __graft_entry__
    so = __graft_entry__.build_guard()
    os.environ["DPIG_GUARD_MODE"] = mode
    alloc = torch.cuda.memory.CUDAPluggableAllocator(so, "dpig_guard_alloc", "dpig_guard_free")
    torch.cuda.memory.change_current_allocator(alloc)
    return alloc


# `DPIG_GUARD=hi|lo python -m pytest tests -m gpu -v -p no:cacheprovider -k "not graph"`: the whole GPU suite on guard pages, so that an
# out-of-bounds access of any kernel any test reaches is a fault in that test (run with AMD_SERIALIZE_KERNEL=3 and -v; faulthandler,
# which pytest enables, prints the Python stack = the C-ABI entry point).  tests/test_guard_gpu.py is the permanent, bounded form.
if os.environ.get("DPIG_GUARD") in ("hi", "lo"):
    _GUARD = install_guard_allocator(os.environ["DPIG_GUARD"])
