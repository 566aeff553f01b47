"""AdamW for the training loop (reference main.py:311
``optim.AdamW(model_pos_train.parameters(), lr=lr, weight_decay=0.1)``).

One HIP launch updates every parameter tensor (include/d3dp_hip.h: d3dp_adamw_step, a chunk table with one block
per 32 K elements) instead of torch's per-tensor op chains (4 kernels x 208 tensors for this model).  The class is a
``torch.optim.Optimizer`` whose state has torch.optim.AdamW's layout (``step``, ``exp_avg``, ``exp_avg_sq`` per
parameter), so ``state_dict()`` / ``load_state_dict()`` interchange with the reference's checkpoints
(main.py:337, 547).  The update follows torch's single-tensor AdamW operation order; scalars (bias corrections, step
size, decay factor) are formed in double on the host and rounded once, as torch does.
"""
from __future__ import annotations

from typing import List

import torch

from . import _lib

CHUNK = 32768


class HipAdamW(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, amsgrad=False):
        if amsgrad:
            raise ValueError("HipAdamW: amsgrad is not used by the reference and is not implemented")
        if lr < 0 or eps < 0 or not (0 <= betas[0] < 1) or not (0 <= betas[1] < 1) or weight_decay < 0:
            raise ValueError("HipAdamW: invalid hyper-parameter")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, amsgrad=False))
        self._tables = {}      # group index -> (signature, device table tensor, n_chunks)

    def _table(self, gi: int, ps: List[torch.Tensor]):
        sig = tuple((p.data_ptr(), p.grad.data_ptr(), p.numel()) for p in ps)
        hit = self._tables.get(gi)
        if hit is not None and hit[0] == sig:
            return hit[1], hit[2], False
        rows = []
        for p in ps:
            st = self.state[p]
            for off in range(0, p.numel(), CHUNK):
                n = min(CHUNK, p.numel() - off)
                rows.append(_lib.AdamChunk(p.data_ptr() + 4 * off, p.grad.data_ptr() + 4 * off,
                                           st["exp_avg"].data_ptr() + 4 * off, st["exp_avg_sq"].data_ptr() + 4 * off, n, 0))
        arr = (_lib.AdamChunk * len(rows))(*rows)
        host = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8)
        dev = host.to(ps[0].device)
        self._tables[gi] = (sig, dev, len(rows))
        return dev, len(rows), True

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        lib = _lib.load()
        for gi, group in enumerate(self.param_groups):
            ps = [p for p in group["params"] if p.grad is not None]
            if not ps:
                continue
            for p in ps:
                if not p.is_cuda:
                    raise _lib.D3DPHipError("HipAdamW updates parameters on an MI355X (parameter on %s); there is no "
                                            "CPU fallback" % p.device)
                if p.dtype != torch.float32 or not p.is_contiguous() or p.grad.dtype != torch.float32:
                    raise _lib.D3DPHipError("HipAdamW: parameters and gradients must be contiguous fp32")
                if not p.grad.is_contiguous():
                    p.grad = p.grad.contiguous()
                st = self.state[p]
                if len(st) == 0:
                    st["step"] = torch.tensor(0.0, dtype=torch.float32)
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            table, n_chunks, fresh = self._table(gi, ps)
            if fresh:      # (re)validated whenever the table is rebuilt: new tensors, loaded state, new gradients
                if len({int(self.state[p]["step"].item()) for p in ps}) != 1:
                    raise _lib.D3DPHipError("HipAdamW: parameters of one group must share a step count")
            step = int(self.state[ps[0]]["step"].item()) + 1
            b1, b2 = group["betas"]
            with torch.cuda.device(ps[0].device):
                _lib.check(lib.d3dp_adamw_step(table.data_ptr(), n_chunks, float(group["lr"]), float(b1), float(b2),
                                               float(group["eps"]), float(group["weight_decay"]), step,
                                               _lib.current_stream()), "d3dp_adamw_step")
            for p in ps:
                self.state[p]["step"] += 1
            # the kernel wrote through raw pointers: tell autograd / the model's packed-weight cache (model.py
            # MixSTE2._context keys on Tensor._version) that the parameters changed
            torch.autograd.graph.increment_version(ps)
        return loss

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        self._tables = {}
