"""Shared helpers of the test-suite."""
import json
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
SCENES = os.path.join(ROOT, "scenes")
ASSETS = os.path.join(ROOT, "assets")
REF_RES = "/root/reference/CUDA_RayTracing/resources"


def hexf(v):
    """Golden JSON stores doubles as C99 hex-float strings (exact)."""
    if isinstance(v, list):
        return [hexf(x) for x in v]
    return float.fromhex(v)


def load_parse_golden(name):
    with open(os.path.join(GOLD, "parse", name + ".json")) as f:
        return json.load(f)


def same_double(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return a.shape == b.shape and bool(np.all((a == b) | (np.isnan(a) & np.isnan(b))))
