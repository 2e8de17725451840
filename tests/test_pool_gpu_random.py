"""Seeded random sweep of the bf16 hot path over the shapes the fast kernels specialise on (embed sizes 256/512/768/1024,
1..4 modalities, head sizes 32..256, ragged batch sizes, key_padding_mask on/off, gradient on the weights on/off)
against the CPU oracle.  Complements the fixed list in test_pool_gpu_shapes.py."""
import random

import pytest
import torch

from tests.helpers import assert_bf16_bounds, f32grad_bounds
from tests.test_pool_gpu_shapes import _case

pytestmark = pytest.mark.gpu


def _cases(n, seed):
    rng = random.Random(seed)
    out = []
    while len(out) < n:
        E = rng.choice([256, 512, 512, 768, 1024])
        hd = rng.choice([32, 64, 128, 256])
        if E % hd or E // hd > 16:
            continue
        H = E // hd
        M = rng.randint(1, 4)
        B = rng.choice([1, 7, 16, 17, 31, 33, 64, 100, 255, 257, 513, 1000])
        kpm = M > 1 and rng.random() < 0.5
        out.append((B, M, E, H, kpm))
    return out


@pytest.mark.parametrize("case", _cases(24, 20260101), ids=lambda c: "B%d_M%d_E%d_H%d_%s" % (c[0], c[1], c[2], c[3], "kpm" if c[4] else "nomask"))
def test_random_bf16_case(case):
    B, M, E, H, kpm = case
    errs, agree = _case(B, M, E, H, torch.bfloat16, kpm, seed=B * 7 + M * 3 + E + H)
    assert_bf16_bounds(errs, f32grad_bounds(B, M, E, H), case)       # per tensor, the same table at every batch size
    assert agree > 0.99
