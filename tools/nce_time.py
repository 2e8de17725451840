"""Times the contrastive side at BASELINE configs[2] size on one GPU: 8192 local rows against 65536 gathered keys,
d = 768, bf16 (N = 1 emulates the all-gather with resident keys).
usage: nce_time.py [mode rows cols d reps]   mode: stream | gemm (one direction each) | sym (both directions, one logits block)
Prints ms per call with HIP events."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from aecf_amd import _lib
    from aecf_amd.layer import _ptr, _stream
    argv = sys.argv[1:]
    mode = argv[0] if argv else "sym"
    rows, cols, d, reps = (int(a) for a in (argv[1:5] + ["8192", "65536", "768", "5"][len(argv[1:5]):]))
    dev = torch.device("cuda:0")
    lib = _lib.load()
    g = torch.Generator(device=dev).manual_seed(5)
    nrm = lambda t: (t / t.norm(dim=-1, keepdim=True)).to(torch.bfloat16)
    q = nrm(torch.randn(rows, d, device=dev, generator=g))
    k = nrm(torch.randn(cols, d, device=dev, generator=g))
    f32 = dict(dtype=torch.float32, device=dev)
    loss_rows, dq, dk = torch.empty(rows, **f32), torch.empty(rows, d, **f32), torch.empty(cols, d, **f32)
    cs = torch.empty(cols, **f32)
    dq16, dk16 = torch.empty(rows, d, dtype=torch.bfloat16, device=dev), torch.empty(cols, d, dtype=torch.bfloat16, device=dev)
    if mode == "stream":
        ws_bytes = lib.aecf_nce_stream_workspace_bytes(rows, cols, d, _lib.AECF_BF16)
    elif mode == "gemm":
        ws_bytes = lib.aecf_nce_workspace_bytes(rows, cols, d, _lib.AECF_BF16)
    else:
        ws_bytes = lib.aecf_nce_sym_workspace_bytes(rows, cols, d)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)

    def call():
        if mode == "sym":
            _lib.check(lib.aecf_nce_sym_pass1(rows, cols, d, 0.07, _ptr(q), _ptr(k), _ptr(ws), ws_bytes, _ptr(cs), _stream()), "pass1")
            _lib.check(lib.aecf_nce_sym_loss(rows, cols, 0, d, 0.07, _ptr(q), _ptr(k), _ptr(cs), _ptr(ws), ws_bytes, _ptr(loss_rows),
                                             0, 2, 0.0, None, 1.0, None, None, _stream()), "loss")
            _lib.check(lib.aecf_nce_sym_grads(rows, cols, 0, d, 0.07, 0.5 / cols, _ptr(q), _ptr(k), _ptr(ws), ws_bytes, None,
                                              _lib.AECF_BF16, _ptr(dq16), _ptr(dk16), _stream()), "grads")
        else:
            _lib.check(lib.aecf_nce_fwd_bwd(rows, cols, 0, d, _lib.AECF_BF16, 0.07, 0.5 / cols, _ptr(q), _ptr(k),
                                            _ptr(loss_rows), _ptr(dq), _ptr(dk), _ptr(ws), ws_bytes, _stream()),
                       "aecf_nce_fwd_bwd")

    call()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
    ev[0].record()
    for i in range(reps):
        call()
        ev[i + 1].record()
    torch.cuda.synchronize()
    ms = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(reps))
    med = ms[len(ms) // 2]
    flops = (8.0 if mode == "stream" else 6.0) * rows * cols * d
    what = {"stream": "one direction, streaming form (8 R C d)", "gemm": "one direction, tile GEMMs (6 R C d)",
            "sym": "BOTH directions, tile GEMMs on one logits block (6 R C d)"}[mode]
    print(f"nce {mode} rows={rows} cols={cols} d={d}: median {med:.3f} ms  min {ms[0]:.3f}  -- {what}: "
          f"{flops / med / 1e9:.1f} TFLOP/s executed = {flops / med / 1e9 / 2500:.3f} of 2.5 PF; ws {ws_bytes >> 20} MB")


if __name__ == "__main__":
    main()
