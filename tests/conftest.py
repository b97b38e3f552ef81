"""pytest configuration: marker registration and shared fixture loaders."""
import csv
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
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_csv(rel):
    """Return dict column-name -> float64 array for a fixture CSV under tests/golden."""
    with open(os.path.join(GOLDEN, rel), newline="") as f:
        rows = list(csv.reader(f))
    names = rows[0]
    data = np.array([[float(v) for v in r] for r in rows[1:]], dtype=np.float64)
    return {n: np.ascontiguousarray(data[:, i]) for i, n in enumerate(names)}


def load_json(rel):
    with open(os.path.join(GOLDEN, rel)) as f:
        return json.load(f)


def nan_or(v):
    return np.nan if v in ("NA", None) else float(v)


def rel_err(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return np.abs(a - b) / np.maximum(np.abs(b), 1e-300)


@pytest.fixture(scope="session")
def known_answers():
    return load_json("known_answers.json")
