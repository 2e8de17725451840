"""Shared helpers for the test-suite (fixture loading, error metrics)."""
import glob
import json
import os

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def load_npz(name):
    z = np.load(os.path.join(GOLDEN, name), allow_pickle=False)
    return {k: z[k] for k in z.files}


def load_json(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


def g2_names(prefix="g2_mha_"):
    return sorted(os.path.basename(p) for p in glob.glob(os.path.join(GOLDEN, prefix + "*.npz")))


def g3_names():
    return sorted(os.path.basename(p) for p in glob.glob(os.path.join(GOLDEN, "g3_mask_*.npz")))


def t(a, dtype=torch.float32):
    return torch.from_numpy(np.asarray(a)).to(dtype)


def rel_err(got, want):
    """max-norm relative error: max|got-want| / max(max|want|, tiny)."""
    got = torch.as_tensor(got).double()
    want = torch.as_tensor(want).double()
    return float((got - want).abs().max() / want.abs().max().clamp_min(1e-30))


def parse_float(s):
    return float(s)
