import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.dirname(os.path.abspath(__file__))):
    if p not in sys.path:
        sys.path.insert(0, p)

import lk_pkg  # noqa: E402

lk_pkg.load()


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")


@pytest.fixture(scope="session")
def oracle_lib():
    import oracle_binding as ob

    ob.build()
    return ob


@pytest.fixture(scope="session")
def hip_lib():
    """The product library.  GPU tests must run the HIP path: no skip, no fallback."""
    import torch  # noqa: F401  (some GPU tests hand torch device pointers to the library: binding.lib() then brings torch's HIP runtime up first)
    from legkilo_amd import binding

    assert os.path.exists(binding.LIB_PATH), "liblegkilo_hip.so missing: run __graft_entry__.build()"
    binding.lib()
    return binding
