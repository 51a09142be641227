import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def _usable_cores():
    """min(affinity, cgroup quota): the GPU box shows 256 logical CPUs under a 16-CPU quota, and a torch thread pool
    sized for 256 makes every CPU-oracle matmul crawl."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, n)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    import torch
    torch.set_num_threads(min(_usable_cores(), 32))


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def pytest_collection_modifyitems(config, items):
    """`-m gpu` tests need a ROCm GPU and the built HIP library: without them they are skipped, not errors (a plain
    `pytest tests` on a CPU box then runs the CPU suite)."""
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="needs a ROCm GPU (MI355X): there is no CPU path for the engine")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


def pytest_terminal_summary(terminalreporter):
    """The margin under the 3-way logit band (tests/parity_util.BAND), visible in every gate log: the worst recorded ratios."""
    from tests import parity_util as P
    if not P.RATIOS:
        return
    worst = sorted(P.RATIOS, key=lambda t: -t[0])[:6]
    terminalreporter.write_line(f"[parity band] {len(P.RATIOS)} three-way checks, band {P.BAND}: worst (err - slack) / reference-err = "
                                + ", ".join(f"{t[0]:.3f} ({t[4]})" for t in worst))
