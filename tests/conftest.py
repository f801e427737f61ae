import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "forks_before_gpu: forks worker processes; ordered before every test that initialises the GPU")


def pytest_collection_modifyitems(config, items):
    import torch
    # tests that FORK worker processes must run before anything in this process touches the GPU (HIP does not survive a fork, and a process
    # that has initialised the GPU must not start another program): they go first, and the check below only counts devices, which does not
    # initialise anything
    items.sort(key=lambda item: 0 if "forks_before_gpu" in item.keywords else 1)
    if torch.cuda.device_count() > 0:
        return
    skip = pytest.mark.skip(reason="no GPU in this process")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def golden():
    import numpy as np
    import torch

    def load(name):
        d = np.load(os.path.join(GOLDEN, name + ".npz"))
        return {k: torch.from_numpy(d[k]) for k in d.files}
    return load


@pytest.fixture(autouse=True)
def poisoned_allocator(request):
    """GENS_TEST_POISON=1: before every GPU test the caching allocator's free blocks are filled with NaN bit patterns, so that a kernel reading
    memory it (or its caller) never initialised -- torch.empty scratch, stash and output buffers -- fails its test instead of passing on
    whatever an earlier test left behind."""
    if os.environ.get("GENS_TEST_POISON") and "gpu" in request.keywords:
        import torch
        if torch.cuda.is_available():
            torch.cuda.synchronize()
            free = [torch.empty(0)]
            try:
                reserved = torch.cuda.memory_reserved() - torch.cuda.memory_allocated()
                for size in (min(reserved, 4 << 30), 1 << 28, 1 << 24, 1 << 20):       # the cached blocks, largest first, then typical small ones
                    for _ in range(4):
                        if size <= 0:
                            break
                        try:
                            free.append(torch.full((size // 4,), float("nan"), device="cuda"))
                        except RuntimeError:
                            break
            finally:
                del free
                torch.cuda.synchronize()
    yield
