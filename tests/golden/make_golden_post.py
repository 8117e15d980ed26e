"""Golden vectors for SURVEY section 8f-3: what the reference's scripts do to a finished sample
(sample/demo_style_transfer.py:265-267, train/finetune_style_diffusion.py:331-332):
    sample = dataset.inv_transform(sample.cpu().permute(0, 2, 3, 1)).float()     # data * std + mean
    sample = recover_from_ric(sample, n_joints)                                   # -> [B, 1, T, J, 3]
Run in the authoring container only (imports /root/reference):  python tests/golden/make_golden_post.py -> post.npz"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402
from make_golden import SEED, syn  # noqa: E402


def main():
    mg.install_shims()
    import importlib
    mp = importlib.import_module("data_loaders.humanml.scripts.motion_process")
    out = {}
    for tag, (F, T, J, B) in {"hml": (263, 196, 22, 2), "short": (263, 37, 22, 3), "j21": (251, 60, 21, 1)}.items():
        sample = torch.from_numpy(syn.normal(SEED, f"post/{tag}/sample", (B, F, 1, T)))
        mean = torch.from_numpy(syn.normal(SEED, f"post/{tag}/mean", (F,))) * 0.3
        std = torch.from_numpy(syn.uniform(SEED, f"post/{tag}/std", (F,), 0.2, 1.5))
        den = (sample.permute(0, 2, 3, 1) * std + mean).float()
        joints = mp.recover_from_ric(den.clone(), J)
        assert tuple(joints.shape) == (B, 1, T, J, 3)
        out[f"{tag}|joints"] = joints.numpy()
        out[f"{tag}|shape"] = np.array([F, T, J, B])
    np.savez_compressed(os.path.join(HERE, "post.npz"), **out)
    print("post.npz", os.path.getsize(os.path.join(HERE, "post.npz")) // 1024, "KiB")


if __name__ == "__main__":
    main()
