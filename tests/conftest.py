import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # children of the GPU tests are started by a helper that exists before any test touches the GPU (tests/spawner.py)
    expr = config.getoption("markexpr", "") or ""
    if "gpu" in expr and "not gpu" not in expr:
        import spawner
        spawner.start()


def pytest_unconfigure(config):
    import spawner
    spawner.stop()


@pytest.fixture(scope="session")
def oracle_lib():
    from oracle import binding
    binding.build()
    return binding.load()


@pytest.fixture(scope="session")
def device():
    """One swr context for the whole GPU session (tests run in one process)."""
    from softwarerenderer_amd import Device
    dev = Device(0)
    yield dev
    dev.close()
