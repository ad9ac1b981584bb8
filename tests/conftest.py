import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    import oracle_lib
    return oracle_lib.Oracle()


@pytest.fixture(scope="session")
def ref():
    import oracle_lib
    if not oracle_lib.have_ref():
        pytest.skip("reference build (oracle/_ref) not available")
    return oracle_lib.Ref()


@pytest.fixture(scope="session")
def ref_mt():
    import oracle_lib
    if not oracle_lib.have_ref_mt():
        pytest.skip("threaded reference build (oracle/_ref/libaocl_ref_mt.so) not available")
    return oracle_lib.RefMT()


@pytest.fixture(scope="session")
def eng():
    """The GPU engine.  Fails (does not skip) when the HIP library or the GPU is missing: -m gpu tests
    must never pass on a fallback."""
    import aocl_compression_amd as a
    return a.engine(0)
