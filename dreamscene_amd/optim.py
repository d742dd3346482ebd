"""Fused Adam over all parameter groups of a GaussianModel in one HIP launch.

Drop-in for the `torch.optim.Adam(l, lr=0.0, eps=1e-15)` the reference builds in `GaussianModel.training_setup`
(gs_renderer.py:615-653): a torch.optim.Optimizer with the same `param_groups` (per-group "lr" and "name", which the
reference's `update_learning_rate` mutates every step) and the same per-parameter `state` entries ("step", "exp_avg",
"exp_avg_sq"), so the optimizer surgery of densification / pruning (`cat_tensors_to_optimizer`, `_prune_optimizer`,
`replace_tensor_to_optimizer`) keeps working on it unchanged. Arithmetic: torch's single-tensor Adam, fp32.
No CPU fallback: parameters must live on a ROCm device."""
from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence

import torch

from . import _lib as L


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, params, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8):
        if not 0.0 <= betas[0] < 1.0 or not 0.0 <= betas[1] < 1.0:
            raise ValueError(f"invalid betas {betas}")
        super().__init__(params, dict(lr=lr, betas=tuple(betas), eps=eps))

    def _state_for(self, p: torch.Tensor) -> dict:
        st = self.state[p]
        if len(st) == 0:
            st["step"] = torch.tensor(0.0, dtype=torch.float32)
            st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
        return st

    @torch.no_grad()
    def step(self, closure=None, grads: Optional[Sequence[Optional[torch.Tensor]]] = None, zero_grad: bool = False,
             set_to_none: bool = False):
        """grads: optional gradient tensors, one per parameter in param_groups order (e.g. views of a GradArena),
        instead of `p.grad`. zero_grad: clear the gradients inside the same pass over memory (persistent buffers such as
        an arena). set_to_none: drop the `.grad` tensors after the step (like optimizer.zero_grad(set_to_none=True)): the
        next backward's gradients then become `.grad` without an accumulation pass."""
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        lib = L.load()
        flat = [(g, p) for g in self.param_groups for p in g["params"]]
        if grads is not None and len(grads) != len(flat):
            raise ValueError("grads must list one tensor (or None) per parameter")
        # parameters that share (betas, eps, step) go into one launch
        batches = {}
        keep = []
        updated = []
        for k, (group, p) in enumerate(flat):
            gr = grads[k] if grads is not None else p.grad
            if gr is None:
                continue
            if p.device.type != "cuda":
                raise L.GsrError("FusedAdam needs parameters on a cuda (ROCm) device; there is no CPU fallback")
            if p.dtype != torch.float32 or not p.is_contiguous():
                raise ValueError("FusedAdam parameters must be contiguous fp32")
            if gr.dtype != torch.float32 or gr.shape != p.shape or gr.device != p.device:
                raise ValueError("gradient does not match its parameter")
            if not gr.is_contiguous():
                gr = gr.contiguous()
            st = self._state_for(p)
            st["step"] += 1
            step = int(st["step"].item())
            for name in ("exp_avg", "exp_avg_sq"):
                if not st[name].is_contiguous():
                    st[name] = st[name].contiguous()
            key = (p.device, group["betas"], float(group["eps"]), step)
            e = L.GsrAdamGroup()
            e.param, e.grad, e.exp_avg, e.exp_avg_sq = p.data_ptr(), gr.data_ptr(), st["exp_avg"].data_ptr(), \
                st["exp_avg_sq"].data_ptr()
            e.numel, e.lr = p.numel(), float(group["lr"])
            if any(ptr % 16 for ptr in (e.param, e.grad, e.exp_avg, e.exp_avg_sq)):
                raise ValueError("FusedAdam tensors must be 16-byte aligned")
            batches.setdefault(key, []).append(e)
            keep.append(gr)
            updated.append(p)
        for (dev, betas, eps, step), entries in batches.items():
            stream = torch.cuda.current_stream(dev).cuda_stream
            with torch.cuda.device(dev):
                for o in range(0, len(entries), L.GSR_MAX_ADAM_GROUPS):
                    chunk = entries[o:o + L.GSR_MAX_ADAM_GROUPS]
                    arr = (L.GsrAdamGroup * len(chunk))(*chunk)
                    L.check(lib.gsr_adam_step(arr, len(chunk), step, betas[0], betas[1], eps, int(zero_grad), stream),
                            "gsr_adam_step")
        # the launch wrote the parameters through raw pointers: tell autograd's version counters (the rasterizer's
        # "inputs unchanged since ..." checks -- rasterizer._check_versions, _SideStreams -- rely on them)
        for p in updated:
            torch.autograd.graph.increment_version(p)
        if set_to_none and grads is None:
            for _, p in flat:
                p.grad = None          # (the launch above holds its own references through `keep` until enqueued)
        return loss
