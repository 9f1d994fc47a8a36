import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _build_oracle():
    """The oracle is the checker for every test module; compile it once (gcc, a second or two).
    Where /root/reference exists the real reference is (re)built too; elsewhere the prebuilt
    oracle/_ref/*.so that travelled with the snapshot is used as is."""
    from oracle import bindings
    bindings.build(ref=True)
