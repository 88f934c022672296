"""Shared pytest configuration.

``-m "not gpu"`` runs on CPU (oracle vs golden vectors, host logic, C-ABI
symbol checks, gloo multi-process); ``-m gpu`` runs the parity tests proper
on an MI355X through the C-ABI library.
"""
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def oracle():
    import oracle as O
    O.build()
    return O


@pytest.fixture(scope="session")
def vectors():
    with open(os.path.join(GOLDEN, "vectors.json")) as f:
        return json.load(f)
