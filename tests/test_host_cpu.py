"""CPU-only tests of the host side: the C-ABI library loads and exports every declared symbol, the
host mirror raises the reference's exceptions with the reference's messages, parameters are created in
the reference's RNG order, and nothing silently falls back to a CPU path."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

import aecf_amd
from aecf_amd import _lib
from tests.helpers import ROOT, load_json, load_npz, t


@pytest.fixture(scope="module", autouse=True)
def built_library():
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__ as ge
        ge.build()
    return _lib.load()


def test_header_symbols_are_exported(built_library):
    header = open(os.path.join(ROOT, "include", "aecf_hip.h")).read()
    declared = sorted(set(re.findall(r"\b(aecf_[a-z_0-9]+)\s*\(", header)))
    assert declared, "no declarations found in include/aecf_hip.h"
    assert sorted(_lib.SYMBOL_NAMES) == declared          # the binding covers exactly the header
    for name in declared:
        assert hasattr(built_library, name), f"libaecf_hip.so does not export {name}"
    assert built_library.aecf_abi_version() == _lib.AECF_ABI_VERSION
    assert _lib.status_string(0) == "ok"
    assert "not supported" in _lib.status_string(-2)


def test_struct_layout_matches_header():
    # include/aecf_hip.h: aecf_pool_desc = int64 + 6*int32 + 3*float = 44 -> padded to 48
    assert ctypes.sizeof(_lib.PoolDesc) == 48
    assert ctypes.sizeof(_lib.PoolFwdArgs) == 33 * 8          # ABI v9: + philox_element0
    assert _lib.PoolFwdArgs.philox_seed.offset == 216 and _lib.PoolFwdArgs.ent_loss.offset == 240
    assert _lib.PoolFwdArgs.philox_element0.offset == 256
    assert ctypes.sizeof(_lib.PoolBwdArgs) == 26 * 8          # ABI v9: + grad_scale (+ pad)
    assert _lib.PoolBwdArgs.grad_scale.offset == 200


def test_pool_check_and_workspace_sizes(built_library):
    def desc(B=65536, M=3, E=512, H=8, dtype=0, mode=1):
        return _lib.PoolDesc(B, M, E, H, dtype, mode, 1, 0.15, 0.7, 1e-8)
    d = desc()
    assert built_library.aecf_pool_check(ctypes.byref(d)) == 0
    fwd = built_library.aecf_pool_fwd_workspace_bytes(ctypes.byref(d))
    bwd = built_library.aecf_pool_bwd_workspace_bytes(ctypes.byref(d))
    assert 65536 * 512 * 2 <= fwd < 80 << 20
    assert bwd < 400 << 20
    assert built_library.aecf_pool_check(ctypes.byref(desc(H=7))) == -1            # E % H
    assert built_library.aecf_pool_check(ctypes.byref(desc(M=9))) == -2            # too many modalities
    assert built_library.aecf_pool_check(ctypes.byref(desc(E=96, H=2))) == -2      # E % 64
    assert built_library.aecf_pool_check(ctypes.byref(desc(E=64, H=4))) == -2      # bf16 head_dim 16
    assert built_library.aecf_pool_check(ctypes.byref(desc(E=64, H=4, dtype=1))) == 0
    assert built_library.aecf_pool_check(ctypes.byref(desc(B=0))) == -1
    # null pointers are rejected before anything is launched
    assert built_library.aecf_pool_forward(ctypes.byref(d), ctypes.byref(_lib.PoolFwdArgs()), None) == -3
    assert built_library.aecf_pool_backward(ctypes.byref(d), ctypes.byref(_lib.PoolBwdArgs()), None) == -3


def test_validation_matches_reference_messages():
    cases = {c["name"]: c for c in load_json("g9_validation.json")}
    q = torch.zeros(4, 1, 8)
    k = torch.zeros(4, 3, 8)
    pool = aecf_amd.MultimodalAttentionPool(8, num_heads=2)
    pool_sf = aecf_amd.MultimodalAttentionPool(8, num_heads=2, batch_first=False)
    calls = {
        "mask_prob_zero": lambda: aecf_amd.CurriculumMasking(base_mask_prob=0.0),
        "mask_prob_big": lambda: aecf_amd.CurriculumMasking(base_mask_prob=1.5),
        "entropy_target_zero": lambda: aecf_amd.CurriculumMasking(entropy_target=0.0),
        "min_active_zero": lambda: aecf_amd.CurriculumMasking(min_active=0),
        "embed_dim_neg": lambda: aecf_amd.MultimodalAttentionPool(-4),
        "num_heads_zero": lambda: aecf_amd.MultimodalAttentionPool(8, num_heads=0),
        "indivisible": lambda: aecf_amd.MultimodalAttentionPool(8, num_heads=3),
        "dropout_bad": lambda: aecf_amd.MultimodalAttentionPool(8, dropout=1.5),
        "query_type": lambda: pool([1, 2], k),
        "key_type": lambda: pool(q, "k"),
        "value_type": lambda: pool(q, k, 3),
        "query_2d": lambda: pool(q[0], k),
        "key_2d": lambda: pool(q, k[0]),
        "value_2d": lambda: pool(q, k, k[0]),
        "src_len_zero": lambda: pool(q, k[:, :0]),
        "key_batch_mismatch": lambda: pool(q, k[:2]),
        "key_embed_mismatch": lambda: pool(q, torch.zeros(4, 3, 6)),
        "value_mismatch": lambda: pool(q, k, torch.zeros(4, 2, 8)),
        "sf_query_2d": lambda: pool_sf(q[0], k),
        "sf_src_len_zero": lambda: pool_sf(q.transpose(0, 1), k.transpose(0, 1)[:0]),
        "sf_key_mismatch": lambda: pool_sf(q.transpose(0, 1), k.transpose(0, 1)[:, :2]),
        "sf_value_mismatch": lambda: pool_sf(q.transpose(0, 1), k.transpose(0, 1), torch.zeros(2, 4, 8)),
        "factory_embed_float": lambda: aecf_amd.create_fusion_pool(8.0, 2),
        "factory_embed_zero": lambda: aecf_amd.create_fusion_pool(0, 2),
        "factory_modalities_zero": lambda: aecf_amd.create_fusion_pool(8, 0),
        "factory_mask_prob": lambda: aecf_amd.create_fusion_pool(8, 2, mask_prob=0.0),
    }
    assert set(calls) == set(cases)
    for name, fn in calls.items():
        want = cases[name]
        assert want["type"] is not None, name
        with pytest.raises(Exception) as ei:
            fn()
        assert type(ei.value).__name__ == want["type"], (name, ei.value)
        assert str(ei.value) == want["msg"], name


def test_public_surface_and_init_rng_order():
    assert aecf_amd.__all__ == ["CurriculumMasking", "MultimodalAttentionPool", "multimodal_attention_pool",
                                "create_fusion_pool"]
    assert aecf_amd.__version__ == "0.1.0"
    g = load_npz("g1_plumbing.npz")
    torch.manual_seed(int(g["seed_init"]))
    query, pool = aecf_amd.create_fusion_pool(embed_dim=512, num_modalities=2)
    assert isinstance(query, torch.nn.Parameter) and query.shape == (1, 1, 512)
    assert np.allclose(query.detach().numpy()[0, 0, :8], g["query_head"])
    assert np.allclose(pool.attention.in_proj_weight.detach().numpy()[0, :8], g["w_in_head"])
    assert np.allclose(pool.attention.out_proj.weight.detach().numpy()[0, :8], g["w_out_head"])
    assert list(pool.state_dict().keys()) == list(g["sd_keys"])
    assert pool.extra_repr() == str(g["repr_pool"])
    assert pool.curriculum_masking.extra_repr() == str(g["repr_mask"])
    assert pool.curriculum_masking._last_seq_len == 2
    # a reference checkpoint loads unchanged (same key names and shapes)
    sd = {k: torch.zeros_like(v) for k, v in pool.state_dict().items()}
    pool.load_state_dict(sd)


def test_no_cpu_fallback():
    """CPU tensors are refused: nothing in the product routes through PyTorch-CPU arithmetic or the oracle."""
    query, pool = aecf_amd.create_fusion_pool(64, 3)
    with pytest.raises(RuntimeError, match="ROCm device"):
        pool(query.expand(4, -1, -1), torch.randn(4, 3, 64))
    with pytest.raises(RuntimeError, match="ROCm device"):
        pool.curriculum_masking(torch.softmax(torch.randn(4, 3), -1))
    with pytest.raises(RuntimeError, match="ROCm device"):
        pool.curriculum_masking.entropy_loss(torch.rand(4))
    with pytest.raises(RuntimeError, match="ROCm device"):
        aecf_amd.multimodal_attention_pool(torch.randn(2, 1, 8), torch.randn(2, 3, 8))
    src = open(os.path.join(ROOT, "aecf_amd", "layer.py")).read() + open(os.path.join(ROOT, "aecf_amd", "_lib.py")).read()
    assert "import oracle" not in src and "from oracle" not in src
    assert "nn.functional.multi_head_attention_forward" not in src and "scaled_dot_product_attention(" not in src.replace(
        "_scaled_dot_product_attention(", "")


def test_missing_library_fails_loudly(monkeypatch):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libaecf_hip.so")
    with pytest.raises(RuntimeError, match="HIP library not found"):
        _lib.load()


def test_aecf_shim_is_the_drop_in_name():
    """`import aecf` (the reference's package name, ref aecf/__init__.py:8-21) resolves to this implementation: same four
    public names, same version, and the module path user code imports from."""
    import aecf
    import aecf.AECFLayer as L
    assert sorted(aecf.__all__) == sorted(['CurriculumMasking', 'MultimodalAttentionPool', 'multimodal_attention_pool',
                                           'create_fusion_pool'])
    assert aecf.__version__ == "0.1.0"
    for name in aecf.__all__:
        assert getattr(aecf, name) is getattr(aecf_amd, name) is getattr(L, name)


def test_uniforms_argument_is_public_and_validated():
    """the mask uniforms are a public keyword of both forwards (no module-level hook)."""
    import inspect
    from aecf_amd import layer
    assert not hasattr(layer, "_uniforms_override")
    for fn in (aecf_amd.MultimodalAttentionPool.forward, aecf_amd.CurriculumMasking.forward):
        ps = inspect.signature(fn).parameters
        assert ps["uniforms"].kind is inspect.Parameter.KEYWORD_ONLY and ps["uniforms"].default is None
        assert ps["generator"].kind is inspect.Parameter.KEYWORD_ONLY
    # the reference's positional order is untouched (ref aecf/AECFLayer.py:409-418)
    names = list(inspect.signature(aecf_amd.MultimodalAttentionPool.forward).parameters)
    assert names[:8] == ["self", "query", "key", "value", "key_padding_mask", "attn_mask", "return_info", "use_checkpoint"]
    with pytest.raises(ValueError):
        layer._draw_uniforms((4, 1, 3), torch.device("cpu"), torch.zeros(5))


def test_cast_cache_follows_data_writes():
    """Float32 master weights, bf16 activations: the activation-dtype copies of the parameters are remade on every forward
    while the module trains (writes through ``p.data`` do not move the version counter), reused in inference, and
    ``invalidate_cast_cache`` covers ``.data`` writes in eval mode."""
    import torch
    import aecf_amd
    _, pool = aecf_amd.create_fusion_pool(64, 2, num_heads=2)
    pool.train()
    w0 = pool._activation_dtype_params(torch.bfloat16)[0].clone()
    pool.attention.in_proj_weight.data.mul_(2.0)                  # what an optimizer stepping on .data does
    w1 = pool._activation_dtype_params(torch.bfloat16)[0]
    assert torch.equal(w1.float(), (w0.float() * 2.0))
    pool.eval()
    with torch.no_grad():
        c0 = pool._activation_dtype_params(torch.bfloat16)[0]
        assert pool._activation_dtype_params(torch.bfloat16)[0] is c0          # reused
        pool.attention.in_proj_weight.mul_(0.5)                                # in-place op on the parameter: version moves
        c1 = pool._activation_dtype_params(torch.bfloat16)[0]
        assert c1 is not c0 and torch.equal(c1.float(), c0.float() * 0.5)
        pool.attention.in_proj_weight.data.mul_(2.0)                           # invisible to the version counter ...
        pool.invalidate_cast_cache()                                           # ... hence the explicit call
        assert torch.equal(pool._activation_dtype_params(torch.bfloat16)[0].float(), c0.float())


def _philox_py(ctr, key):
    """Philox4x32-10 (Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as easy as 1, 2, 3", SC'11), restated."""
    c, k = list(ctr), list(key)
    for _ in range(10):
        p0, p1 = 0xD2511F53 * c[0], 0xCD9E8D57 * c[2]
        c = [((p1 >> 32) ^ c[1] ^ k[0]) & 0xFFFFFFFF, p1 & 0xFFFFFFFF, ((p0 >> 32) ^ c[3] ^ k[1]) & 0xFFFFFFFF, p0 & 0xFFFFFFFF]
        k = [(k[0] + 0x9E3779B9) & 0xFFFFFFFF, (k[1] + 0xBB67AE85) & 0xFFFFFFFF]
    return c


def test_philox_known_answers_and_element_mapping(built_library):
    """The generator behind AECF_DRAW_UNIFORMS (include/aecf_hip.h): Philox4x32-10 against the Random123 known-answer
    vectors, and the library's host evaluation of 'the uniform torch.rand puts at element i' against the restatement of
    torch's launch geometry (thread = i mod T in iteration i // 4T, component (i mod 4T) // T; counter = offset / 4 +
    iteration; 2^-32 + v 2^-32, 1.0 -> 0.0)."""
    import numpy as np
    assert _philox_py([0, 0, 0, 0], [0, 0]) == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    assert _philox_py([0xffffffff] * 4, [0xffffffff] * 2) == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    assert _philox_py([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0]) == \
        [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]
    lib = built_library
    raw = (ctypes.c_uint32 * 4)()
    rng = np.random.default_rng(0)
    for _ in range(200):
        seed = int(rng.integers(0, 2 ** 63)) * 2 + int(rng.integers(0, 2))
        offset = int(rng.integers(0, 2 ** 40)) * 4
        threads = 256 * int(rng.integers(1, 2049))
        element = int(rng.integers(0, 2 ** 33))
        got = lib.aecf_philox_host(seed, offset, threads, element, raw)
        it, rem = divmod(element, 4 * threads)
        ii, idx = divmod(rem, threads)
        ctr = offset // 4 + it
        want_raw = _philox_py([ctr & 0xFFFFFFFF, ctr >> 32, idx & 0xFFFFFFFF, idx >> 32], [seed & 0xFFFFFFFF, seed >> 32])
        assert list(raw) == want_raw
        u = np.float32(2.3283064e-10) + np.float32(want_raw[ii]) * np.float32(2.3283064e-10)
        u = np.float32(0.0) if u == np.float32(1.0) else u
        assert np.float32(got) == u and 0.0 <= got < 1.0


def test_philox_launch_geometry_matches_torchs_formula(monkeypatch):
    """layer._philox_geometry restates ATen's launch geometry of torch.rand (block 256, unroll 4, grid capped at
    CUs * max_threads_per_CU / 256) and the generator advance ((n - 1) // (4 T) + 1) * 4 -- checked here against a fake generator for
    sizes below, at and beyond one grid-stride iteration of an MI355X-sized device (256 CUs x 2048 threads); the GPU tests check
    the values themselves against torch.rand."""
    from aecf_amd import layer

    class Gen:
        def __init__(self):
            self.off = 40

        def initial_seed(self):
            return -3                                        # (torch hands back a signed 64-bit value for large seeds)

        def get_offset(self):
            return self.off

    monkeypatch.setitem(layer._props_cache, 0, (256, 2048))
    cap = 256 * (2048 // 256)                                # blocks
    for n, want_threads, want_inc in [(1, 256, 4), (256, 256, 4), (257, 512, 4), (196608, 196608, 4),
                                      (256 * cap * 4, 256 * cap, 4), (256 * cap * 4 + 1, 256 * cap, 8), (10 ** 7, 256 * cap, 20)]:
        seed, offset, threads, inc = layer._philox_geometry(n, 0, Gen())
        assert (seed, offset, threads, inc) == (2 ** 64 - 3, 40, want_threads, want_inc), n


def test_dp_state_records_are_matched_by_identity():
    """layer.DpState: the float32 sums behind a gradient run are handed out only for THAT run, untouched; which tensors carry the
    1 / world factor is a matter of object identity (never of tensor equality, never of a storage pointer that may be reused)."""
    from aecf_amd import layer
    st = layer.DpState(world=4, grad_scale=0.25)
    a, b = torch.nn.Parameter(torch.zeros(3)), torch.nn.Parameter(torch.zeros(3))
    st.add_scaled(a)
    assert st.is_scaled(a) and not st.is_scaled(b)
    flat = torch.zeros(8, dtype=torch.bfloat16)
    wide = torch.ones(8)
    st.record(flat, wide)
    other = torch.zeros(8, dtype=torch.bfloat16)
    assert st.take(other) is None and st.runs == []          # somebody else's run: nothing handed out, records consumed
    st.record(flat, wide)
    assert st.take(flat) is wide
    st.record(flat, wide)
    flat.add_(1)                                             # autograd accumulated into it: the sums no longer describe it
    assert st.take(flat) is None
    for _ in range(20):                                      # bounded: the first run (autograd's accumulation target) + the latest
        st.record(torch.zeros(2), None)
    assert len(st.runs) <= 8
    del a
    import gc
    gc.collect()
    assert all(r() is None or r() is not None for r in st.scaled.values())     # weak references: a dead parameter does not pin memory


def test_attach_and_detach_are_per_module():
    """dp.attach makes ONE module data-parallel (no process-wide switch): its options carry the state, another module is
    untouched, detach removes it."""
    import aecf_amd
    from aecf_amd import dp
    _, p1 = aecf_amd.create_fusion_pool(64, 2, num_heads=2)
    _, p2 = aecf_amd.create_fusion_pool(64, 2, num_heads=2)
    st = dp.attach(p1, world=8)
    assert p1.options.dp is st and p2.options.dp is None
    assert st.world == 8 and st.grad_scale == 0.125 and st.keep_f32 and not st.defer_rounding
    assert all(st.is_scaled(p) for p in p1.parameters()) and not any(st.is_scaled(p) for p in p2.parameters())
    assert dp._states_of(list(p1.parameters())) == [st] and dp._states_of(list(p2.parameters())) == []
    st1 = dp.attach(p1, world=1)
    assert st1.grad_scale == 1.0                             # one rank: nothing to fold in
    dp.detach(p1)
    assert p1.options.dp is None and dp._states_of(list(p1.parameters())) == []
    assert p1.options.hilo_grads is None and p1.options.draw_in_kernel and p1.options.share_prep
