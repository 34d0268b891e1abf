import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

DATA21 = os.path.join(ROOT, "data", "problem-21-11315-pre.txt")
DATA39 = os.path.join(ROOT, "data", "problem-39-18060-pre.txt")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")


def to_oracle(p):
    """bundleadjustment_benchmarks_amd.Problem -> tests.oracle_lib.Problem (same arrays)."""
    import oracle_lib as O
    a = p.arrays()
    return O.Problem(p.N, p.M, p.K, a["cam_idx"], a["pt_idx"], a["meas"], a["cams9"], a["pts"])


@pytest.fixture(scope="session")
def ba():
    import bundleadjustment_benchmarks_amd as m
    m.lib()
    return m


@pytest.fixture(scope="session")
def O():
    import oracle_lib
    oracle_lib.lib()
    return oracle_lib


@pytest.fixture(scope="session")
def prob21(ba):
    return ba.Problem.load_bal(DATA21)


@pytest.fixture(scope="session")
def prob39(ba):
    return ba.Problem.load_bal(DATA39)


@pytest.fixture(scope="session")
def gpu_ok(ba):
    try:
        name, cus = ba.device_info()
    except ba.BAError as e:
        pytest.fail("no HIP device for a gpu-marked test: %s" % e)
    return name, cus
