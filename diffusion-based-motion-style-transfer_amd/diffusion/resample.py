"""Timestep sampler of the fine-tune loop: the counterpart of the reference's `diffusion/resample.py` for the one sampler the
scripts create ('uniform', train/training_loop.py:92-94).

What has to match the reference is observable behaviour, not text: for a seeded numpy global RNG, `sample` must consume the
generator exactly as the reference does (ONE `np.random.choice` call with a probability vector: resample.py:52-59) so that a
seeded fine-tune run visits the same timesteps, and the importance weights of a uniform draw are all 1.
"""
import numpy as np
import torch


def host_to_device_async(t, device):
    """A small host tensor onto the device WITHOUT tying the host to the GPU.  `tensor.to(device)` from pageable memory (the reference:
    resample.py:59-62, `th.from_numpy(indices_np).long().to(device)`) is a blocking copy: the host waits until the stream has reached it,
    i.e. until the whole previous iteration has finished on the GPU, and only then starts enqueuing this one -- measured on MI355X
    (tools/ft_events.py, round 6): 11.7 ms per fine-tune iteration with the blocking copy of the 64 timesteps, 10.4 ms without.  Through a
    pinned staging buffer (torch's caching host allocator: no allocation after the first call) the copy is just another command in
    the stream; the values are the same."""
    device = torch.device(device)
    if device.type != "cuda":
        return t.to(device)
    return t.pin_memory().to(device, non_blocking=True)


class UniformSampler:
    """Uniform timesteps over the whole process, or over the loop's restricted `data_range` (the first
    (1000 - skip_steps) / 1000 * 20 indices of the respaced process, training_loop.py:241-242)."""

    def __init__(self, diffusion):
        self.diffusion = diffusion
        self.n = int(diffusion.num_timesteps)

    def weights(self):
        return np.ones([self.n])

    def sample(self, batch_size, device, data_range=None):
        support = np.arange(self.n) if data_range is None else np.asarray(data_range)
        k = len(support)
        # same generator consumption as the reference: choice(population, size, p) with p = 1/k each
        # (np.ones / sum, as there, so that the probabilities are the same doubles bit for bit)
        prob = np.ones([k]) / np.sum(np.ones([k]))
        drawn = np.random.choice(len(prob) if data_range is None else data_range, size=(batch_size,), p=prob)
        steps = host_to_device_async(torch.from_numpy(np.asarray(drawn)).long(), device)
        # importance weight 1 / (k p) = 1 for every draw of a uniform sampler
        return steps, torch.ones(batch_size, dtype=torch.float32, device=device)


class LossAwareSampler:
    """Never instantiated by the scripts; the loop only asks `isinstance(sampler, LossAwareSampler)` (training_loop.py:263)."""


def create_named_schedule_sampler(name, diffusion):
    if name != "uniform":
        raise NotImplementedError(f"unknown schedule sampler: {name}")      # 'loss-second-moment' is never created by the scripts
    return UniformSampler(diffusion)
