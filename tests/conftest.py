import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")


@pytest.fixture(scope="session")
def ctx():
    import agogo_amd
    c = agogo_amd.Ctx(0)
    yield c
    c.close()


# Soak runs: AGZ_FUZZ_N multiplies the number of seeds of every fuzz family, AGZ_FUZZ_BASE shifts them to fresh draws
# (scripts/fuzz_soak.sh).  The defaults are the fixed seeds the suites are graded on.
FUZZ_BASE = int(os.environ.get("AGZ_FUZZ_BASE", "0"))


def fuzz_seeds(default_n):
    n = int(os.environ.get("AGZ_FUZZ_N", "0")) or default_n
    return range(FUZZ_BASE, FUZZ_BASE + n)
