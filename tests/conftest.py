import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "ref: needs the reference build oracle/_ref (authoring container only)")


@pytest.fixture(scope="session")
def client():
    from tfhe_amd import Client
    return Client()


@pytest.fixture(scope="session")
def keys(client):
    """client-side keys from seed 42 (same bytes on every machine)"""
    return client.keygen(42)


@pytest.fixture(scope="session")
def oracle():
    from tests.oracle_lib import Oracle
    return Oracle()


@pytest.fixture(scope="session")
def okeys(oracle, keys):
    return oracle.keys_from_coef(keys.bk_coef, keys.ks)


@pytest.fixture(scope="session")
def engine(keys):
    from tfhe_amd import Engine
    e = Engine(0)
    e.load_keys_coef(keys.bk_coef, keys.ks)
    yield e
    e.close()


@pytest.fixture(scope="session")
def fast(keys):
    """a second context on GPU 0 in the FAST arithmetic mode (tfhe_hip_set_arithmetic), the bench headline's mode"""
    from tfhe_amd import Engine
    e = Engine(0)
    e.set_arithmetic("fast")          # before the keys: the FAST key is derived at load
    e.load_keys_coef(keys.bk_coef, keys.ks)
    assert e.arithmetic == "fast"
    yield e
    e.close()
