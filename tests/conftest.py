import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _usable_cores() -> int:
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = min(n, max(1, int(int(q) / int(per))))
    except Exception:
        pass
    return max(1, n)


# The oracle (the checker of these tests) is OpenMP code that takes every core it can see. A GPU box of the pool shows 256 cores to a pod that may use a fraction of them
# beside other pods (load average 45 seen): 256 spinning threads made every oracle call of the suite 5-10 x slower (8 min instead of 2.5 for the same tests). Before any
# OpenMP runtime is loaded: a bounded team that sleeps at its barriers.
os.environ.setdefault("OMP_NUM_THREADS", str(min(_usable_cores(), 32)))
os.environ.setdefault("OMP_WAIT_POLICY", "passive")
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")


GOLDEN = os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
