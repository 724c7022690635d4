import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

RK_HUBER = ((1, float(np.sqrt(5.991))), (1, float(np.sqrt(7.815))))   # samples/sample_comparison_with_g2o.cpp:195-200
RK_NONE = ((0, 0.0), (0, 0.0))
RK_TUKEY = ((2, 4.0), (2, 5.0))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def small_graph():
    from cuba_amd.synth import synth_ba
    return synth_ba(40, 600, 2400, seed=1)


@pytest.fixture(scope="session")
def small_fp(small_graph):
    from cuba_amd.graph import flatten
    return flatten(small_graph)


def with_fixed(g, fixed_pose_rows=(), fixed_lm_rows=()):
    """Copy of a Graph with extra fixed vertices."""
    import copy
    h = copy.deepcopy(g)
    h.pose_fixed[list(fixed_pose_rows)] = True
    h.lm_fixed[list(fixed_lm_rows)] = True
    return h
