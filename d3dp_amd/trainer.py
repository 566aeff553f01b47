"""Training loop of the hot path's model (reference main.py:305-593).

What runs where:
  * batches        d3dp_amd.data.ChunkedBatcher  -- pools in HBM, one gather launch per batch (generators.py:12-171)
  * forward/backward  D3DP(is_train=True)        -- d3dp_q_sample + d3dp_train_forward / d3dp_train_backward
  * loss           mpjpe (loss.py:6-13) and the reference's seeding ``loss.backward(loss.detach())`` (main.py:393)
  * update         d3dp_amd.optim.HipAdamW       -- one launch over all tensors (main.py:311: AdamW, weight decay 0.1)
  * per epoch      validation with a 1-hypothesis 1-step sampler (main.py:416-472), exponential lr decay (:519-522),
                   ``epoch_N.bin`` / ``best_epoch.bin`` checkpoints (:530-568) in the reference's dict layout
                   {'epoch','lr','random_state','optimizer','model_pos'} with DataParallel-style ``module.`` keys, so
                   the reference's --resume / --evaluate and this build's read each other's files.
"""
from __future__ import annotations

import os
from time import time
from typing import Callable, Dict, Iterable, List, Optional

import torch

from . import jpma
from .clips import clip_gather
from .optim import HipAdamW


def mpjpe(predicted: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
    """loss.py:6-13."""
    assert predicted.shape == target.shape
    return torch.mean(torch.norm(predicted - target, dim=len(target.shape) - 1))


def _strip(sd: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    return {(k[len("module."):] if k.startswith("module.") else k): v for k, v in sd.items()}


def checkpoint_dict(epoch: int, lr: float, batcher, optimizer, model) -> dict:
    """main.py:543-552."""
    return {"epoch": epoch, "lr": lr, "random_state": batcher.random_state(), "optimizer": optimizer.state_dict(),
            "model_pos": {"module." + k: v for k, v in _strip(model.state_dict()).items()}}


def load_checkpoint(path: str, model, optimizer=None, batcher=None, map_location="cpu") -> dict:
    """main.py:252-258, 335-343."""
    ck = torch.load(path, map_location=map_location, weights_only=False)
    model.load_state_dict(_strip(ck["model_pos"]), strict=False)
    if optimizer is not None:
        if ck.get("optimizer") is not None:
            optimizer.load_state_dict(ck["optimizer"])
            if batcher is not None:
                batcher.set_random_state(ck["random_state"])
        else:
            print("WARNING: this checkpoint does not contain an optimizer state. The optimizer will be reinitialized.")
    return ck


@torch.no_grad()
def validate(model_eval, sequences: Iterable, receptive_field: int, kps_left, kps_right, device, debug=False):
    """main.py:416-472: per-sequence clips, flip input, H=1/K=1 sampler, root zeroing, mpjpe_diffusion; returns the
    N-weighted mean error per step (K,) in metres."""
    total, N = None, 0
    for _, batch, batch_2d in sequences:
        s3 = torch.as_tensor(batch, dtype=torch.float32, device=device).reshape(-1, *batch.shape[-2:])
        s2 = torch.as_tensor(batch_2d, dtype=torch.float32, device=device).reshape(-1, *batch_2d.shape[-2:])
        x2, x2f = clip_gather(s2, receptive_field, kps_left, kps_right)
        x3, _ = clip_gather(s3, receptive_field)
        x3[:, :, 0] = 0
        pred = model_eval(x2, x3, input_2d_flip=x2f)
        pred[:, :, :, :, 0] = 0
        err = jpma.mpjpe_diffusion(pred, x3)
        w = x3.shape[0] * x3.shape[1]
        total = w * err if total is None else total + w * err
        N += w
        if debug:
            break
    return total / N


def fit(args, model_train, model_eval, batcher, test_sequences: Optional[Callable[[], Iterable]], device,
        kps_left=None, kps_right=None, log: Callable[[str], None] = print, on_iteration=None,
        forward_kwargs: Optional[Callable[[int, int], dict]] = None) -> Dict[str, List[float]]:
    """The training loop (main.py:305-568).  ``args`` carries the reference's fields: learning_rate, lr_decay, epochs,
    checkpoint, checkpoint_frequency, resume, coverlr, min_loss, no_eval, debug, number_of_frames.  Returns the loss
    curves the reference keeps (per-epoch means, metres).  ``forward_kwargs(epoch, iteration)`` may inject recorded
    diffusion draws (``t``, ``noise``) or DropPath masks for parity runs; production leaves it None."""
    lr = args.learning_rate
    optimizer = HipAdamW(model_train.parameters(), lr=lr, weight_decay=0.1)
    lr_decay = args.lr_decay
    hist = {"losses_3d_train": [], "losses_3d_valid": [], "iter_loss": [], "lr": []}
    epoch, min_loss = 0, getattr(args, "min_loss", 100000)
    if getattr(args, "resume", ""):
        ck = load_checkpoint(os.path.join(args.checkpoint, args.resume), model_train, optimizer, batcher)
        epoch = ck["epoch"]
        if not getattr(args, "coverlr", False):
            lr = ck["lr"]
        for g in optimizer.param_groups:      # the optimizer state dict carries the decayed lr as well
            g["lr"] = lr
    log("** Note: reported losses are averaged over all frames.")
    if args.checkpoint:
        os.makedirs(args.checkpoint, exist_ok=True)
    log_path = os.path.join(args.checkpoint, "training_log.txt") if args.checkpoint else None

    while epoch < args.epochs:
        start_time = time()
        epoch_loss, N, iteration = 0.0, 0, 0
        model_train.train()
        num_batches = batcher.batch_num()
        for _, batch_3d, batch_2d in batcher.next_epoch():
            if iteration % 1000 == 0:
                log("%d/%d" % (iteration, num_batches))
            inputs_3d, inputs_2d = batch_3d, batch_2d
            if not batcher.zero_root:
                inputs_3d = inputs_3d.clone()
                inputs_3d[:, :, 0] = 0                                   # main.py:364-365
            optimizer.zero_grad()
            kw = forward_kwargs(epoch, iteration) if forward_kwargs is not None else {}
            predicted = model_train(inputs_2d, inputs_3d, **kw)          # main.py:370
            loss = mpjpe(predicted, inputs_3d)
            loss.backward(loss.clone().detach())                         # main.py:393
            li = loss.item()
            w = inputs_3d.shape[0] * inputs_3d.shape[1]
            epoch_loss += w * li
            N += w
            optimizer.step()
            hist["iter_loss"].append(li)
            if on_iteration is not None:
                on_iteration(epoch, iteration, li)
            iteration += 1
            if args.debug and N == w:
                break
        hist["losses_3d_train"].append(epoch_loss / N)

        valid = None
        if not getattr(args, "no_eval", False) and test_sequences is not None:
            model_eval.load_state_dict(_strip(model_train.state_dict()), strict=False)       # main.py:412
            model_eval.eval()
            valid = validate(model_eval, test_sequences(), args.number_of_frames, kps_left, kps_right, device, args.debug)
            hist["losses_3d_valid"].append(valid.detach().cpu())
        elapsed = (time() - start_time) / 60
        hist["lr"].append(lr)
        if valid is None:
            line = "[%d] time %.2f lr %f 3d_train %f 3d_pos_train %f" % (
                epoch + 1, elapsed, lr, hist["losses_3d_train"][-1] * 1000, hist["losses_3d_train"][-1] * 1000)
        else:
            line = "[%d] time %.2f lr %f 3d_train %f 3d_pos_train %f 3d_pos_valid %f" % (
                epoch + 1, elapsed, lr, hist["losses_3d_train"][-1] * 1000, hist["losses_3d_train"][-1] * 1000,
                valid[0].item() * 1000)
        log(line)
        if log_path:
            with open(log_path, mode="a") as f:
                f.write(line + "\n")

        lr *= lr_decay                                                    # main.py:519-522
        for g in optimizer.param_groups:
            g["lr"] *= lr_decay
        epoch += 1

        if args.checkpoint and epoch % args.checkpoint_frequency == 0:    # main.py:530-552
            chk_path = os.path.join(args.checkpoint, "epoch_{}.bin".format(epoch))
            log("Saving checkpoint to " + chk_path)
            torch.save(checkpoint_dict(epoch, lr, batcher, optimizer, model_train), chk_path)
        if args.checkpoint and valid is not None and valid[0].item() * 1000 < min_loss:       # main.py:555-568
            min_loss = valid[0].item() * 1000
            log("save best checkpoint")
            torch.save(checkpoint_dict(epoch, lr, batcher, optimizer, model_train),
                       os.path.join(args.checkpoint, "best_epoch.bin"))
            with open(log_path, mode="a") as f:
                f.write("best epoch\n")
    hist["optimizer"] = optimizer
    return hist
