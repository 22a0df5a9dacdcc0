import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

GOLDEN = os.path.join(REPO, "tests", "golden")

# The suite checks the window batching itself (ragged tails, streams, ranges): batch_size means exactly batch_size windows per network
# call here, as in the reference; the product default (score_fn.py::window_batch_floor) has its own tests, which set the attribute.
os.environ.setdefault("C2W_WINDOW_BATCH_FLOOR", "0")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
