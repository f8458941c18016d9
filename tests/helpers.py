"""Shared helpers for the test-suite: golden fixtures, synthetic inputs, oracle construction."""
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


def synthetic():
    from core.utils import synthetic as syn
    return syn


def oracle_for(g):
    """Build the numpy oracle with the seeded weights a golden fixture was generated with."""
    import danbo_oracle as o
    syn = synthetic()
    cfg = syn.model_config(str(g["cfg_name"]))
    rest = syn.rest_pose(cfg["rest_scale"])
    sd = syn.make_state_dict(cfg, seed=int(g["weight_seed"]), n_framecodes=int(g["n_framecodes"]), rest=rest)
    cls = o.DanboOracle if cfg["nerf_type"] == "danbo" else o.AnerfOracle
    return cls(cfg, sd, rest), cfg, sd, rest


def max_err(a, b):
    return float(np.max(np.abs(np.asarray(a, dtype=np.float64) - np.asarray(b, dtype=np.float64))))


def rel_err(a, b, floor=1e-3):
    """max |a-b| / max(|b|, floor): relative error with an absolute floor for near-zero entries."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), floor)))
