import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver on the GPU box)")


GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="session")
def golden():
    import json

    return json.load(open(os.path.join(GOLDEN_DIR, "golden.json")))


@pytest.fixture(scope="session")
def fixture_bytes():
    d = os.path.join(GOLDEN_DIR, "inputs")
    return {n: open(os.path.join(d, n), "rb").read() for n in sorted(os.listdir(d))}


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O

    O.lib()
    return O


@pytest.fixture(scope="session")
def hip_lib():
    """The product library. GPU tests must load the in-tree .so -- never a fallback."""
    import lilliput_amd

    if not os.path.exists(lilliput_amd.lib_path()):
        lilliput_amd.build()
    return lilliput_amd.lib()


@pytest.fixture(scope="session")
def batch(hip_lib):
    import lilliput_amd

    b = lilliput_amd.Batch(0)
    yield b
    b.close()
