import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _native_library_is_built():
    """liblde.so is git-ignored: build it (hipcc cross-compiles without a GPU) if it is missing or stale, so that a
    fresh checkout runs the suite. Building is not a fallback: without the library every product call raises."""
    import latentdiffeq_amd
    latentdiffeq_amd.build_lib()


@pytest.fixture(scope="session")
def o32():
    from oracle import oracle as O
    return O.Oracle("f32")


@pytest.fixture(scope="session")
def o64():
    from oracle import oracle as O
    return O.Oracle("f64")
