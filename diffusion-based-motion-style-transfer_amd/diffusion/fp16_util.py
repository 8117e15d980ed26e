"""`MixedPrecisionTrainer` of the reference (diffusion/fp16_util.py:148-231) for the only mode the loop uses
(`use_fp16 = False  # deprecating this option`, train/training_loop.py:58): zero_grad / backward / optimize and the
state-dict helpers.  `optimize` logs grad_norm and param_norm like :208-223, but from the optimizer kernel's two
device-side sums and ONE host sync instead of 192 `.item()` calls; the norm of parameters that never receive a
gradient (the frozen motion encoder) is computed once -- they do not change."""
import numpy as np
import torch as th

from . import logger


class MixedPrecisionTrainer:
    def __init__(self, *, model, use_fp16=False, fp16_scale_growth=1e-3, initial_lg_loss_scale=20.0):
        if use_fp16:
            raise NotImplementedError("use_fp16 is deprecated in the reference's loop and not implemented")
        self.model = model
        self.use_fp16 = False
        self.model_params = list(self.model.parameters())
        self.master_params = self.model_params
        self.param_groups_and_shapes = None
        self.lg_loss_scale = initial_lg_loss_scale
        self._frozen_sq = None

    def zero_grad(self):
        grads = [p.grad for p in self.model_params if p.grad is not None]       # in place, like the reference (:133-138):
        if grads:                                                                 # gradient buffers (and any bucket views)
            for g in grads:                                                       # keep their addresses across iterations
                if not g._is_view():                                              # (views of a flat sink / bucket are
                    g.detach_()                                                   #  graph-free already)
            th._foreach_zero_(grads)

    def backward(self, loss):
        loss.backward()

    def _frozen_param_sq(self):
        if self._frozen_sq is None:
            with th.no_grad():
                self._frozen_sq = float(sum(th.sum(p.float() ** 2) for p in self.master_params if not p.requires_grad))
        return self._frozen_sq

    def optimize(self, opt):
        fused = hasattr(opt, "last_sq_norms")
        if not fused:                                   # any torch optimizer: norms the reference's way, then step
            grad_norm, param_norm = self._compute_norms()
            logger.logkv_mean("grad_norm", grad_norm)
            logger.logkv_mean("param_norm", param_norm)
            opt.step()
            return True
        opt.step()
        sq = opt.last_sq_norms
        if sq is None:                                  # no parameter had a gradient
            g2, p2 = 0.0, float(sum(th.sum(p.float() ** 2) for p in self.master_params if p.requires_grad))
        else:
            g2, p2 = (float(v) for v in sq.tolist())    # the one sync
            with th.no_grad():                           # trainable parameters the step skipped (no gradient this time)
                p2 += float(sum(th.sum(p.float() ** 2) for p in self.master_params if p.requires_grad and p.grad is None))
        self.last_norms = (float(np.sqrt(g2)), float(np.sqrt(p2 + self._frozen_param_sq())))
        logger.logkv_mean("grad_norm", self.last_norms[0])
        logger.logkv_mean("param_norm", self.last_norms[1])
        return True

    def _compute_norms(self, grad_scale=1.0):
        grad_norm, param_norm = 0.0, 0.0
        for p in self.master_params:
            with th.no_grad():
                param_norm += th.norm(p, p=2, dtype=th.float32).item() ** 2
                if p.grad is not None:
                    grad_norm += th.norm(p.grad, p=2, dtype=th.float32).item() ** 2
        return np.sqrt(grad_norm) / grad_scale, np.sqrt(param_norm)

    def master_params_to_state_dict(self, master_params):
        return self.model.state_dict()

    def state_dict_to_master_params(self, state_dict):
        return [state_dict[name] for name, _ in self.model.named_parameters()]
