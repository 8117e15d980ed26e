"""inv_transform + recover_from_ric.  CPU: the oracle restatement against the reference's own
outputs (tests/golden/post.npz).  GPU: the one-launch kernel (through the C ABI) against the same goldens."""
import numpy as np
import pytest
import torch

import mst_amd  # noqa: F401
import mst_amd.synthetic as syn
from conftest import SEED, rel_l2

CASES = ("hml", "short", "j21")


def case(golden, tag):
    g = golden["post"]
    F, T, J, B = (int(v) for v in g[f"{tag}|shape"])
    sample = syn.normal(SEED, f"post/{tag}/sample", (B, F, 1, T))
    mean = syn.normal(SEED, f"post/{tag}/mean", (F,)) * 0.3
    std = syn.uniform(SEED, f"post/{tag}/std", (F,), 0.2, 1.5)
    return sample, mean.astype(np.float32), std.astype(np.float32), J, g[f"{tag}|joints"]


@pytest.mark.parametrize("tag", CASES)
def test_oracle_matches_reference(golden, tag):
    from oracle import postprocess
    sample, mean, std, J, want = case(golden, tag)
    got = postprocess.recover_joints(sample, mean, std, J)
    assert got.shape == want.shape
    assert rel_l2(got, want) < 2e-5          # fp32 running sums of ~200 O(1) yaw velocities feed sin/cos: summation-order noise


@pytest.mark.gpu
@pytest.mark.parametrize("tag", CASES)
def test_kernel_matches_reference(golden, tag):
    from mst_amd.utils.motion_process import recover_from_ric, recover_joints
    sample, mean, std, J, want = case(golden, tag)
    dev = torch.device("cuda:0")
    got = recover_joints(torch.from_numpy(sample).to(dev), mean, std, J)
    assert tuple(got.shape) == want.shape
    assert rel_l2(got.cpu().numpy(), want) < 3e-5
    # drop-in signature on denormalised rows, arbitrary leading dims
    den = (torch.from_numpy(sample).permute(0, 2, 3, 1) * torch.from_numpy(std) + torch.from_numpy(mean)).float().to(dev)
    got2 = recover_from_ric(den, J)
    assert rel_l2(got2.cpu().numpy(), want) < 3e-5
    with pytest.raises(RuntimeError, match="GPU only"):
        recover_joints(torch.from_numpy(sample), mean, std, J)
    with pytest.raises(RuntimeError, match="cannot hold"):
        recover_joints(torch.from_numpy(sample).to(dev), mean, std, 200)
