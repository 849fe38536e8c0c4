import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
# the concurrency tests keep 16 proofs in flight: give them as many hardware queues as bench.py does (HIP's default of 4 would
# serialise them four to a queue); read by the HIP runtime when it initialises
os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    with open(os.path.join(ROOT, "tests", "golden", "golden.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def orc():
    import oracle_lib
    return oracle_lib.load()


@pytest.fixture(scope="session")
def gpu():
    """The product library with a live context; fails loudly if the HIP extension or the GPU is missing."""
    import plonky2_demo_amd as p
    ctx = p.default_context()
    ctx.capture_intermediates(True)       # the parity tests compare Z / partial products and quotient chunks too
    return p, ctx
