"""Runnable counterpart of the reference's example trainer around the fusion path (behaviour
of ``train_both_models`` / ``main`` in ``xrays/train_xrays_example.py:312-427, 736-783`` for the AECF model): AdamW
(lr 1e-4, weight decay 0.01) + BCEWithLogits, curriculum masking and missing-modality training switched on at epoch 40 of
60, per-epoch validation mAP with and without each modality -- data-parallel over the GPUs of one node.

    python -m aecf_amd.train_xray --epochs 60 --switch-epoch 40                     # one GPU
    python -m torch.distributed.run --nproc-per-node 4 --master-addr 127.0.0.1 -m aecf_amd.train_xray   # DP x 4

The reference trains on precomputed CLIP features of a chest X-ray set that is not part of its repository
(``.MISSING_LARGE_BLOBS``); ``--data train.pt,val.pt`` loads such files (dicts with image / text / labels), otherwise a
synthetic set of the same shape is generated (CLIP-like 512-d features whose label signal is split between the
modalities, so that the fusion has something to learn).  Plots, sklearn metrics and the baseline model of the reference
script are reporting code outside the path and are not rebuilt.
"""
from __future__ import annotations

import argparse
import json
import os
import time

import torch
import torch.distributed as dist

from . import dp
from .optim import FusedAdamW
from .xray import AECFModel


def synthetic_split(n: int, num_classes: int, dim: int, seed: int, device, proto_seed: int = 1234):
    gp = torch.Generator().manual_seed(proto_seed)        # the class prototypes are the "world": same for every split
    proto_img = torch.randn(num_classes, dim, generator=gp)
    proto_txt = torch.randn(num_classes, dim, generator=gp)
    g = torch.Generator().manual_seed(seed)
    labels = (torch.rand(n, num_classes, generator=g) < 0.2).float()
    half = num_classes // 2                       # the first classes show in the image, the rest in the text
    li, lt = labels.clone(), labels.clone()
    li[:, half:] *= 0.25
    lt[:, :half] *= 0.25
    image = li @ proto_img + 1.5 * torch.randn(n, dim, generator=g)
    text = lt @ proto_txt + 1.5 * torch.randn(n, dim, generator=g)
    return image.to(device), text.to(device), labels.to(device)


def mean_average_precision(scores: torch.Tensor, labels: torch.Tensor) -> float:
    """Macro average precision over the classes that have a positive (what sklearn's average_precision_score computes
    per class, ref xrays/train_xrays_example.py:260-295), on the device."""
    order = scores.argsort(dim=0, descending=True)
    hit = labels.gather(0, order)
    tp = hit.cumsum(0)
    prec = tp / torch.arange(1, scores.shape[0] + 1, device=scores.device, dtype=scores.dtype).unsqueeze(1)
    npos = labels.sum(0)
    ap = (prec * hit).sum(0) / npos.clamp_min(1.0)
    keep = npos > 0
    return float(ap[keep].mean()) if bool(keep.any()) else 0.0


@torch.no_grad()
def evaluate(model: AECFModel, image, text, labels, drop: str, batch: int = 4096) -> float:
    model.eval()
    outs = []
    for i in range(0, image.shape[0], batch):
        im, tx = image[i:i + batch], text[i:i + batch]
        if drop == "images":
            im = torch.zeros_like(im)
        elif drop == "texts":
            tx = torch.zeros_like(tx)
        outs.append(model(im, tx).float())
    return mean_average_precision(torch.cat(outs), labels)


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--epochs", type=int, default=60)
    ap.add_argument("--switch-epoch", type=int, default=40, help="curriculum masking + missing-modality training from here")
    ap.add_argument("--batch", type=int, default=64, help="GLOBAL batch per step (the reference's 64), sharded over the ranks")
    ap.add_argument("--lr", type=float, default=1e-4)
    ap.add_argument("--samples", type=int, default=8192)
    ap.add_argument("--val-samples", type=int, default=2048)
    ap.add_argument("--classes", type=int, default=15)
    ap.add_argument("--hidden", type=int, default=256)
    ap.add_argument("--data", default=None, help="train.pt,val.pt with image/text/labels tensors (default: synthetic)")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--save", default=None, help="state_dict file written by rank 0 at the end (ref :766-772)")
    args = ap.parse_args(argv)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    dev_index = local % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    if world > 1:
        backend = os.environ.get("AECF_DIST_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)

    if args.data:
        tr, va = (torch.load(p, map_location=device) for p in args.data.split(","))
        image, text, labels = tr["image"].float(), tr["text"].float(), tr["labels"].float()
        v_image, v_text, v_labels = va["image"].float(), va["text"].float(), va["labels"].float()
        args.classes = labels.shape[1]
    else:
        image, text, labels = synthetic_split(args.samples, args.classes, 512, args.seed + 1, device)
        v_image, v_text, v_labels = synthetic_split(args.val_samples, args.classes, 512, args.seed + 2, device)

    torch.manual_seed(args.seed + 17 * rank)                       # replicas differ until the broadcast below
    model = AECFModel(image.shape[1], text.shape[1], args.classes, args.hidden).to(device)
    params = list(model.parameters())
    dp.broadcast_parameters(params + list(model.buffers()))
    bucket = dp.FlatGradBucket(params) if world > 1 else None
    opt = FusedAdamW(params, lr=args.lr, weight_decay=0.01)                 # ref :322-323 (torch.optim.AdamW's update, one launch)
    crit = torch.nn.BCEWithLogitsLoss()
    n = image.shape[0]
    steps = (n + args.batch - 1) // args.batch                              # the short last batch is kept (DataLoader default)
    perm_gen = torch.Generator(device=device).manual_seed(args.seed + 3)    # same permutation on every rank
    mask_gen = torch.Generator(device=device).manual_seed(args.seed + 4)    # same global mask uniforms on every rank
    miss_gen = torch.Generator(device=device).manual_seed(args.seed + 5)    # same global missing-modality draws on every rank
    history = []
    for epoch in range(args.epochs):
        if epoch == args.switch_epoch:                                      # ref :346-349
            model.toggle_curriculum(True)
            model.missing_modality_training = True
        model.train()
        perm = torch.randperm(n, device=device, generator=perm_gen)
        t0 = time.perf_counter()
        loss_sum = torch.zeros((), device=device)
        ent_sum = torch.zeros((), device=device)
        for it in range(steps):
            idx = perm[it * args.batch:(it + 1) * args.batch]
            b = idx.numel()                                                 # < args.batch in the last step of an epoch
            lo, hi = dp.shard_bounds(b, rank, world)
            sel = idx[lo:hi]
            u = torch.rand(b, 1, 2, device=device, generator=mask_gen)[lo:hi]       # (every rank draws: the streams stay aligned)
            missing = None
            if model.missing_modality_training:
                da, db = model.draw_missing(b, device, generator=miss_gen)  # the GLOBAL batch's draws; this rank's rows
                missing = (da[lo:hi], db[lo:hi])
            if hi <= lo:                                # fewer rows than ranks: this rank only joins the collective
                bucket.zero()
                bucket.all_reduce(average=True)
                opt.step()
                continue
            logits, info = model(image[sel], text[sel], return_info=True, mask_uniforms=u, missing=missing)
            loss = crit(logits, labels[sel])
            if bucket is None:
                opt.zero_grad(set_to_none=True)
            else:
                bucket.zero()
            (loss * dp.shard_loss_scale(hi - lo, b, world)).backward()
            if bucket is not None:
                bucket.all_reduce(average=True)
            opt.step()
            loss_sum += loss.detach()
            if "entropy" in info:
                ent_sum += info["entropy"].float().mean()
        row = dict(epoch=epoch + 1, curriculum=model.curriculum_enabled,
                   train_loss=float(dp.all_reduce_mean_scalar(loss_sum / steps)),
                   gate_entropy=float(dp.all_reduce_mean_scalar(ent_sum / steps)),
                   sec=time.perf_counter() - t0)
        if rank == 0:
            row.update(val_map=evaluate(model, v_image, v_text, v_labels, "none"),
                       val_map_no_images=evaluate(model, v_image, v_text, v_labels, "images"),
                       val_map_no_texts=evaluate(model, v_image, v_text, v_labels, "texts"))
            print(json.dumps(row), flush=True)
        history.append(row)
        if world > 1:
            dist.barrier()
    if rank == 0 and args.save:
        torch.save({"aecf_state_dict": model.state_dict(), "history": history}, args.save)
    if world > 1:
        dist.destroy_process_group()
    return history


if __name__ == "__main__":
    main()
