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


def pytest_collection_modifyitems(config, items):
    """LK_TEST_ORDER=reverse | shuffle:<seed>: the collected tests in another order (tools/gpu_order_suite.sh).  The tests share one process and
    one GPU: anything that only works behind a particular predecessor - memory an earlier handle left behind, a static of the library - shows up
    under another order (round 5 found such a fault in the default order)."""
    order = os.environ.get("LK_TEST_ORDER", "")
    if order == "reverse":
        items.reverse()
    elif order.startswith("shuffle:"):
        import random

        random.Random(int(order.split(":", 1)[1])).shuffle(items)


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
