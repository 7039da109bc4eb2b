import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(REPO, "tests", "golden")


@pytest.fixture(autouse=True)
def _collect_engines_between_tests():
    """A ``StepEngine`` sits in reference cycles (networks <-> engine), so it -- with its HIP streams, events and
    instantiated hipGraphs -- is only released by the cyclic collector.  Left to the collector's own schedule, a long
    GPU session piles up hundreds of streams and graph executables; round 3's larger suite then died with a
    segmentation fault inside ``hipGraphLaunch`` of a branched graph (only in the full run, never in the test alone).
    Collect after every test."""
    yield
    import gc
    gc.collect()
