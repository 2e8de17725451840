"""the nn.MultiheadAttention options outside the shared-query hot path, on the HIP general
path (aecf_mha_forward/backward), against outputs of the reference module itself (g11, g2 perq) and the oracle."""
import numpy as np
import pytest
import torch

from tests.helpers import load_npz, rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _pool(c, dev, batch_first=True, dropout=0.0, cm=None, dtype=torch.float32):
    import aecf_amd
    pool = aecf_amd.MultimodalAttentionPool(int(c["E"]), num_heads=int(c["H"]), dropout=dropout, batch_first=batch_first,
                                            curriculum_masking=cm)
    with torch.no_grad():
        pool.attention.in_proj_weight.copy_(torch.from_numpy(c["w_in"]))
        pool.attention.in_proj_bias.copy_(torch.from_numpy(c["b_in"]))
        pool.attention.out_proj.weight.copy_(torch.from_numpy(c["w_out"]))
        pool.attention.out_proj.bias.copy_(torch.from_numpy(c["b_out"]))
    return pool.to(dev, dtype)


@pytest.mark.parametrize("name", ["bool2d", "float3d_kpm", "kv_diff", "seqfirst"])
def test_general_options_match_reference(name):
    g = load_npz("g11_general.npz")
    c = {k.split(".", 1)[1]: g[k] for k in g if k.startswith(name + ".")}
    bfirst = bool(int(c["batch_first"]))
    pool = _pool(c, DEV, batch_first=bfirst).eval()
    T = lambda k: torch.from_numpy(np.asarray(c[k])).to(DEV)
    q, key = T("query").requires_grad_(True), T("key").requires_grad_(True)
    value = T("value").requires_grad_(True) if "value" in c else None
    am = T("attn_mask") if "attn_mask" in c else None
    kpm = T("key_padding_mask") if "key_padding_mask" in c else None
    tr = (lambda t_: t_) if bfirst else (lambda t_: t_.transpose(0, 1))
    y, info = pool(tr(q), tr(key), None if value is None else tr(value), key_padding_mask=kpm, attn_mask=am,
                   return_info=True)
    wbar = info["attention_weights"]
    ((tr(y) * T("dy")).sum() + (wbar * T("dwbar")).sum()).backward()
    torch.cuda.synchronize()
    cpu = lambda t_: t_.detach().float().cpu()
    a = pool.attention
    got = dict(y=cpu(tr(y)), wbar=cpu(wbar), dquery=cpu(q.grad), dkey=cpu(key.grad), dw_in=cpu(a.in_proj_weight.grad),
               db_in=cpu(a.in_proj_bias.grad), dw_out=cpu(a.out_proj.weight.grad), db_out=cpu(a.out_proj.bias.grad))
    if value is not None:
        got["dvalue"] = cpu(value.grad)
    for k_, v_ in got.items():
        assert rel_err(v_, c[k_]) < 1e-5, (name, k_, rel_err(v_, c[k_]))


def test_per_sample_queries_tgt_len_2_match_reference():
    """g2 perq_t2: distinct queries per sample, two query positions, gradients on the output and on the weights."""
    c = load_npz("g2_mha_e64h2m3_perq_t2.npz")
    pool = _pool(c, DEV).train()
    T = lambda k: torch.from_numpy(np.asarray(c[k])).to(DEV)
    x, q = T("x").requires_grad_(True), T("query").requires_grad_(True)
    y, info = pool(q, x, return_info=True)
    ((y * T("dy")).sum() + (info["attention_weights"] * T("dwbar")).sum()).backward()
    torch.cuda.synchronize()
    cpu = lambda t_: t_.detach().float().cpu()
    a = pool.attention
    got = dict(y=cpu(y), wbar=cpu(info["attention_weights"]), dx=cpu(x.grad), dquery=cpu(q.grad),
               dw_in=cpu(a.in_proj_weight.grad), db_in=cpu(a.in_proj_bias.grad), dw_out=cpu(a.out_proj.weight.grad),
               db_out=cpu(a.out_proj.bias.grad))
    for k_, v_ in got.items():
        assert rel_err(v_, c[k_]) < 1e-5, (k_, rel_err(v_, c[k_]))


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-5), (torch.bfloat16, 1e-3 + 2.0 ** -8)])
def test_attention_dropout_matches_oracle_for_the_same_uniforms(dtype, tol):
    """Training-mode attention dropout: keep = (u >= p), weights scaled by 1/(1-p) and returned after dropout
    (torch functional.py:6591-6606).  The uniforms are the module's own torch.rand draw, reproduced by seed."""
    from oracle import aecf_oracle as O
    B, T, S, E, H, pd = 40, 3, 5, 128, 4, 0.25
    g = torch.Generator().manual_seed(21)
    bf = lambda t_: t_.to(torch.bfloat16).float()
    c = dict(E=E, H=H, w_in=bf(torch.randn(3 * E, E, generator=g) / E ** 0.5).numpy(),
             b_in=bf(torch.randn(3 * E, generator=g) * 0.05).numpy(), w_out=bf(torch.randn(E, E, generator=g) / E ** 0.5).numpy(),
             b_out=bf(torch.randn(E, generator=g) * 0.05).numpy())
    q, k, v = bf(torch.randn(B, T, E, generator=g) * 0.5), bf(torch.randn(B, S, E, generator=g)), bf(torch.randn(B, S, E, generator=g))
    dy, dw = bf(torch.randn(B, T, E, generator=g)), torch.randn(B, T, S, generator=g)
    pool = _pool(c, DEV, dropout=pd, dtype=dtype).train()
    qd, kd, vd = (t_.to(DEV, dtype).requires_grad_(True) for t_ in (q, k, v))
    torch.manual_seed(77)
    y, info = pool(qd, kd, vd, return_info=True)
    torch.manual_seed(77)
    U = torch.rand(B * H, T, S, device=DEV).cpu().reshape(B, H, T, S)
    ((y.float() * dy.to(DEV)).sum() + (info["attention_weights"].float() * dw.to(DEV)).sum()).backward()
    torch.cuda.synchronize()
    W = lambda k_: torch.from_numpy(c[k_])
    f = O.mha_forward(q, k, v, W("w_in"), W("b_in"), W("w_out"), W("b_out"), H, None, None, U, pd)
    b = O.mha_backward(q, k, v, W("w_in"), W("b_in"), W("w_out"), H, f, dy, dw)
    cpu = lambda t_: t_.detach().float().cpu()
    assert float((f["wbar"] == 0).float().mean()) > 0.0            # some weights were dropped in every head
    assert rel_err(cpu(y), f["y"]) < tol and rel_err(cpu(info["attention_weights"]), f["wbar"]) < tol
    assert rel_err(cpu(qd.grad), b["dquery"]) < 2 * tol and rel_err(cpu(kd.grad), b["dkey"]) < 2 * tol
    assert rel_err(cpu(vd.grad), b["dvalue"]) < 2 * tol
    a = pool.attention
    ptol = tol if dtype == torch.float32 else 6e-3
    assert rel_err(cpu(a.in_proj_weight.grad), b["dw_in"]) < ptol and rel_err(cpu(a.out_proj.weight.grad), b["dw_out"]) < ptol
    assert rel_err(cpu(a.in_proj_bias.grad), b["db_in"]) < ptol and rel_err(cpu(a.out_proj.bias.grad), b["db_out"]) < ptol


def test_general_path_with_curriculum_masking_info_contract():
    """tgt_len 2 + curriculum masking in training mode: the hook runs on the pooled [B,T,M] weights exactly as the
    reference applies it (ref aecf/AECFLayer.py:526-541): same keys, shapes, dtypes and gradient flags."""
    import aecf_amd
    from aecf_amd import layer
    from oracle import aecf_oracle as O
    B, T, M, E, H = 64, 2, 3, 64, 2
    cm = aecf_amd.CurriculumMasking(base_mask_prob=0.3)
    pool = aecf_amd.MultimodalAttentionPool(E, num_heads=H, curriculum_masking=cm).to(DEV).train()
    g = torch.Generator().manual_seed(4)
    x = (torch.randn(B, M, E, generator=g) * torch.tensor([1.0, 2.0, 3.0]).view(1, 3, 1)).to(DEV)
    q = torch.randn(B, T, E, generator=g).to(DEV)
    U = torch.rand(B, T, M, generator=g)
    y, info = pool(q, x, return_info=True, uniforms=U)
    assert y.shape == (B, T, E)
    assert set(info) == {"entropy", "mask_rate", "target_entropy", "attention_weights", "masked_attention_weights"}
    assert info["attention_weights"].shape == (B, T, M) and info["attention_weights"].requires_grad
    assert info["mask_rate"].dtype == torch.float32 and not info["entropy"].requires_grad
    m = O.curriculum_mask_train(info["attention_weights"].detach().cpu(), U, 0.3)
    assert torch.equal((info["masked_attention_weights"] != 0).cpu(), m["masked"] != 0)
    assert cm._last_seq_len == M


def test_general_path_masks_rows_of_100_keys():
    """A pooled sequence of 100 keys with curriculum masking (the reference's module is length-agnostic, ref :130-283; the
    mask kernels take any row length): weights, masked weights and pattern against the oracle on the kernel's own weights."""
    import aecf_amd
    from oracle import aecf_oracle as O
    B, S, E, H = 48, 100, 64, 4
    g = torch.Generator().manual_seed(100)
    q, pool = aecf_amd.create_fusion_pool(E, S, num_heads=H)
    pool.curriculum_masking = aecf_amd.CurriculumMasking(base_mask_prob=0.5, min_active=2)
    pool = pool.to(DEV).train()
    x = torch.randn(B, S, E, generator=g)
    U = torch.rand(B, 1, S, generator=g)
    a = pool.attention
    cpu = lambda t_: t_.detach().float().cpu()
    y, info = pool(q.to(DEV).expand(B, -1, -1), x.to(DEV), return_info=True, uniforms=U)
    f, m = O.pool_forward_train(cpu(q).expand(B, -1, -1), x, cpu(a.in_proj_weight), cpu(a.in_proj_bias), cpu(a.out_proj.weight),
                                cpu(a.out_proj.bias), H, U, 0.5, min_active=2)
    assert rel_err(cpu(y), f["y"]) < 1e-5 and rel_err(cpu(info["attention_weights"]), f["wbar"]) < 1e-5
    mine = O.curriculum_mask_train(cpu(info["attention_weights"]), U, 0.5, min_active=2)
    assert torch.equal(cpu(info["masked_attention_weights"]) != 0, mine["masked"] != 0)
    assert rel_err(cpu(info["masked_attention_weights"]), mine["masked"]) < 1e-5
    assert rel_err(cpu(info["entropy"]), m["entropy"]) < 1e-5


def test_functional_slow_path_matches_reference_g7():
    """SURVEY 8a row A8, ref aecf/AECFLayer.py:643-652: anything but the projection-free fast path builds a FRESH randomly
    initialised module per call (CPU generator state at call time) and runs it.  Fixtures g7 slow_seed71_h4 (E = 32, 4
    heads of 8, eval) and slow_seed72_train (2 heads, training-mode curriculum masking, value = None) are the
    reference's outputs under those seeds."""
    import aecf_amd
    g = load_npz("g7_functional.npz")
    q, k, v = (torch.from_numpy(g[n]).to(DEV) for n in ("q", "k", "v"))
    torch.manual_seed(71)
    out = aecf_amd.multimodal_attention_pool(q, k, v, num_heads=4)
    assert rel_err(out.cpu(), g["slow_seed71_h4"]) < 1e-5
    torch.manual_seed(72)
    out = aecf_amd.multimodal_attention_pool(q[:, :1], k, None, num_heads=2,
                                             curriculum_masking=aecf_amd.CurriculumMasking(0.3).to(DEV), training=True)
    assert rel_err(out.cpu(), g["slow_seed72_train"]) < 1e-5


@pytest.mark.parametrize("E,H,B,T,S,dtype,tol", [
    (32, 4, 5, 2, 5, torch.float32, 1e-5),         # head_dim 8
    (96, 3, 4, 3, 7, torch.float32, 1e-5),         # E % 64 != 0
    (192, 6, 3, 2, 4, torch.bfloat16, 1e-2),       # bf16, 6 heads of 32
    (64, 2, 2, 100, 150, torch.float32, 1e-5),     # tgt_len, src_len > 64: one query chunk, long rows
    (64, 4, 2, 200, 700, torch.float32, 2e-5),     # several query chunks: dk / dv carried over the chunks
])
def test_general_path_reach(E, H, B, T, S, dtype, tol):
    """nn.MultiheadAttention accepts any embed_dim divisible by num_heads and any sequence lengths
    (ref aecf/AECFLayer.py:384-391); so does the general path -- against the oracle, forward and backward."""
    import aecf_amd
    from oracle import aecf_oracle as O
    g = torch.Generator().manual_seed(E + T + S)
    pool = aecf_amd.MultimodalAttentionPool(E, num_heads=H)
    with torch.no_grad():
        pool.attention.in_proj_bias.normal_(0, 0.1, generator=g)
        pool.attention.out_proj.bias.normal_(0, 0.1, generator=g)
        if dtype == torch.bfloat16:
            for prm in pool.parameters():
                prm.copy_(prm.to(dtype).float())
    rnd = lambda *sh: torch.randn(*sh, generator=g).to(dtype).float()
    q, k, v, dy, dw = rnd(B, T, E), rnd(B, S, E), rnd(B, S, E), rnd(B, T, E), rnd(B, T, S)
    kpm = torch.rand(B, S, generator=g) < 0.2
    kpm[:, 0] = False
    a = pool.attention
    w = [t_.detach().clone() for t_ in (a.in_proj_weight, a.in_proj_bias, a.out_proj.weight, a.out_proj.bias)]
    f = O.mha_forward(q, k, v, w[0], w[1], w[2], w[3], H, kpm)
    b = O.mha_backward(q, k, v, w[0], w[1], w[2], H, f, dy, dw)
    pool = pool.to(DEV, dtype).eval()
    qd, kd, vd = (t_.to(DEV, dtype).requires_grad_(True) for t_ in (q, k, v))
    y, info = pool(qd, kd, vd, key_padding_mask=kpm.to(DEV), return_info=True)
    ((y.float() * dy.to(DEV)).sum() + (info["attention_weights"].float() * dw.to(DEV)).sum()).backward()
    torch.cuda.synchronize()
    cpu = lambda t_: t_.detach().float().cpu()
    got = dict(y=cpu(y), wbar=cpu(info["attention_weights"]), dquery=cpu(qd.grad), dkey=cpu(kd.grad), dvalue=cpu(vd.grad),
               dw_in=cpu(pool.attention.in_proj_weight.grad), db_in=cpu(pool.attention.in_proj_bias.grad),
               dw_out=cpu(pool.attention.out_proj.weight.grad), db_out=cpu(pool.attention.out_proj.bias.grad))
    want = dict(y=f["y"], wbar=f["wbar"], dquery=b["dquery"], dkey=b["dkey"], dvalue=b["dvalue"], dw_in=b["dw_in"],
                db_in=b["db_in"], dw_out=b["dw_out"], db_out=b["db_out"])
    for k_ in want:
        assert rel_err(got[k_], want[k_]) < tol, (k_, rel_err(got[k_], want[k_]))
