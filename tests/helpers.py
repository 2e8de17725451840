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


def hot_shape_inputs(seed, B=256, M=3, E=512, H=8):
    """Seeded bf16-representable inputs and parameters at the headline shape (d=512, 8 heads, M=3), built the same way by
    tests/golden/make_golden.py (fixture G6) and by the GPU tests: CPU generator, so identical on every machine."""
    g = torch.Generator().manual_seed(seed)
    r = lambda *shape: torch.randn(*shape, generator=g)
    bf = lambda t_: t_.to(torch.bfloat16).to(torch.float32)
    d = dict(B=B, M=M, E=E, H=H)
    d["x"] = bf(r(B, M, E) * torch.tensor([1.0, 1.5, 2.0][:M]).view(1, M, 1))
    d["query"] = bf(r(1, 1, E) * (2.0 / E) ** 0.5)
    d["w_in"] = bf(r(3 * E, E) * (1.0 / E) ** 0.5)
    d["b_in"] = bf(r(3 * E) * 0.05)
    d["w_out"] = bf(r(E, E) * (1.0 / E) ** 0.5)
    d["b_out"] = bf(r(E) * 0.05)
    d["dy"] = bf(r(B, 1, E))
    d["dwbar"] = bf(r(B, 1, M))
    return d


def record_errors(name, **kw):
    """Append measured parity errors to gpurun_out/parity_errors.jsonl (merged back from the GPU box): the per-tensor bounds
    of the bf16 tests are set from these records."""
    path = os.path.join(ROOT, "gpurun_out")
    os.makedirs(path, exist_ok=True)
    with open(os.path.join(path, "parity_errors.jsonl"), "a") as f:
        f.write(json.dumps(dict(test=name, **{k: float(v) for k, v in kw.items()})) + "\n")


from aecf_amd._tolerances import BF16_BOUNDS, BF16_F32GRAD_BOUNDS, BF16_F32GRAD_HILO_BOUNDS  # noqa: E402,F401  (one table: the package's)


def f32grad_bounds(B, M, E, H):
    """Bounds of float32-stored parameter gradients of the bf16 kernels for this shape: the tight table where the library
    builds the hi + lo weight-gradient products (they are then on by themselves), the bf16-operand table elsewhere."""
    import ctypes
    from aecf_amd import _lib
    desc = _lib.PoolDesc(B, M, E, H, _lib.AECF_BF16, 1, 1, 0.15, 0.7, 1e-8)
    hilo = _lib.load().aecf_pool_hilo_bwd_workspace_bytes(ctypes.byref(desc)) > 0
    return BF16_F32GRAD_HILO_BOUNDS if hilo else BF16_F32GRAD_BOUNDS


def assert_bf16_bounds(errs, bounds, what, scale=1.0):
    for k, e in errs.items():
        assert e < scale * bounds[k], (what, k, e, scale * bounds[k])
