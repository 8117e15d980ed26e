"""`FusedAdamW`: torch.optim.AdamW's interface and state layout (state_dict / load_state_dict are inherited, so
`opt{step:09d}.pt` files interchange with the reference's, train/training_loop.py:96, :343-348) with the update
done by ONE native launch over all tensors (mst_adamw_step), which also returns the trainer's norms."""
import ctypes as C

import torch

from . import _native as N


class FusedAdamW(torch.optim.AdamW):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2):
        super().__init__(params, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, foreach=False, fused=False)
        self._ws = None
        self._norms = None
        self._tables_for = None            # the pointer tuple the device-side tables in self._ws were written from
        self.last_sq_norms = None          # device tensor [sum g^2, sum p^2 before the update] of the last step()

    @torch.no_grad()
    def step(self, closure=None):
        if closure is not None:
            raise NotImplementedError("closures are not used by the training loop")
        self.last_sq_norms = None
        for group in self.param_groups:
            if group.get("amsgrad") or group.get("maximize"):
                raise NotImplementedError("amsgrad / maximize")
            ps = [p for p in group["params"] if p.grad is not None]
            if not ps:
                continue
            dev = ps[0].device
            if dev.type != "cuda":
                raise RuntimeError("FusedAdamW needs GPU parameters (the native kernel is the only implementation)")
            steps = set()
            for p in ps:
                st = self.state[p]
                if len(st) == 0:                                     # torch.optim.AdamW's lazy state, same keys / types
                    st["step"] = torch.tensor(0.0, dtype=torch.float32)
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                if p.dtype != torch.float32 or not p.is_contiguous() or not p.grad.is_contiguous() or p.grad.dtype != torch.float32:
                    raise RuntimeError("FusedAdamW: float32 contiguous parameters and gradients only")
                st["step"] += 1
                steps.add(int(st["step"]))
            if len(steps) != 1:
                raise RuntimeError("FusedAdamW: parameters of one group must share their step count")
            n = len(ps)
            numel = (C.c_int64 * n)(*[p.numel() for p in ps])
            need = N.lib().mst_adamw_workspace_bytes(n, numel)
            if len(self.param_groups) > 1:
                raise NotImplementedError("FusedAdamW keeps one table workspace: a single parameter group")
            if self._ws is None or self._ws.numel() < need or self._ws.device != dev:
                self._ws = torch.empty(need, dtype=torch.uint8, device=dev)       # owned for the optimizer's lifetime
                self._norms = torch.zeros(2, dtype=torch.float32, device=dev)
                self._tables_for = None
            self._norms.zero_()
            tensors = (ps, [p.grad for p in ps], [self.state[p]["exp_avg"] for p in ps], [self.state[p]["exp_avg_sq"] for p in ps])
            key = tuple(t.data_ptr() for ts in tensors for t in ts) + tuple(numel)
            dirty = key != self._tables_for      # first step, a tensor moved (new gradient buffer, load_state_dict), new workspace
            arr = lambda ts: (C.c_void_p * n)(*[t.data_ptr() for t in ts])
            b1, b2 = group["betas"]
            N.check(N.lib().mst_adamw_step(n, arr(tensors[0]), arr(tensors[1]), arr(tensors[2]), arr(tensors[3]), numel,
                                           float(group["lr"]), float(b1), float(b2), float(group["eps"]),
                                           float(group["weight_decay"]), steps.pop(), N.ptr(self._norms), N.ptr(self._ws),
                                           self._ws.numel(), int(dirty), N.stream_ptr(dev)))
            self._tables_for = key
            for p in ps:                                   # written through raw pointers: tell autograd (and the engine's
                torch.autograd.graph.increment_version(p)  # weight-version check, which re-uploads f16 copies) they changed
            self.last_sq_norms = self._norms if self.last_sq_norms is None else self.last_sq_norms + self._norms
        return None
