"""Shared helpers for the test-suite: golden-vector loading and limb conversions."""
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_json(name):
    with open(os.path.join(GOLDEN, name)) as fh:
        return json.load(fh)


def load_npz(name):
    return np.load(os.path.join(GOLDEN, name))


def unhex(limbs):
    return np.array([int(v, 16) for v in limbs], dtype=np.uint64)


def unhex_rows(rows):
    return np.array([[int(v, 16) for v in r] for r in rows], dtype=np.uint64)
