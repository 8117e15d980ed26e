"""Timestep samplers of the training loop (reference `diffusion/resample.py:8-73`).  The loop hard-codes
'uniform' (train/training_loop.py:92-94); `sample` keeps the reference's np.random call pattern, so a seeded
run draws the same indices."""
import numpy as np
import torch as th


def create_named_schedule_sampler(name, diffusion):
    if name == "uniform":
        return UniformSampler(diffusion)
    raise NotImplementedError(f"unknown schedule sampler: {name}")      # 'loss-second-moment' is never created by the scripts


class ScheduleSampler:
    def weights(self):
        raise NotImplementedError

    def sample(self, batch_size, device, data_range=None):
        """(:42-63) indices ~ weights (restricted to `data_range` when given), and the importance weights."""
        w = self.weights()
        p = w / np.sum(w)
        if data_range is None:
            indices_np = np.random.choice(len(p), size=(batch_size,), p=p)
        else:
            w_1 = self.weights()[data_range]
            p = w_1 / np.sum(w_1)
            indices_np = np.random.choice(data_range, size=(batch_size,), p=p)
        indices = th.from_numpy(indices_np).long().to(device)
        weights_np = 1 / (len(p) * p[indices_np])
        weights = th.from_numpy(weights_np).float().to(device)
        return indices, weights


class UniformSampler(ScheduleSampler):
    def __init__(self, diffusion):
        self.diffusion = diffusion
        self._weights = np.ones([diffusion.num_timesteps])

    def weights(self):
        return self._weights


class LossAwareSampler(ScheduleSampler):
    """Marker base class (the loop's isinstance check, training_loop.py:263); no instance is ever created."""
