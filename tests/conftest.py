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


def fresh_seed(seed):
    """Seed of a LIVE differential test (product against the reference's libraries under oracle/_ref): the committed value by default;
    LILLIPUT_FUZZ_SEED_OFFSET=k moves every such test to streams no earlier run has seen (scripts/r06_fresh_fuzz.sh runs a range of
    offsets, with the sanitizer build of the library when there is one). Tests against RECORDED answers never use this."""
    return seed + int(os.environ.get("LILLIPUT_FUZZ_SEED_OFFSET", "0"))


def _patch_default_rng():
    """LILLIPUT_FUZZ_RNG_OFFSET=k: every np.random.default_rng(int) of the test modules becomes default_rng(int + k) -- the differential GPU
    tests (tests/test_damaged.py, test_gpu_sweep.py, test_progressive.py, test_gpu_parity.py: product against oracle / reference libraries,
    computed live) then run on streams, sizes and option mixes no earlier run has seen (scripts/r06_fresh_gpu.sh). Not for the tests that
    compare with RECORDED answers."""
    k = int(os.environ.get("LILLIPUT_FUZZ_RNG_OFFSET", "0"))
    if not k:
        return
    import numpy as np

    real = np.random.default_rng

    def shifted(seed=None, *a, **kw):
        return real(seed + k if isinstance(seed, int) else seed, *a, **kw)

    np.random.default_rng = shifted


_patch_default_rng()
