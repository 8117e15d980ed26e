"""Post-sampling tensor ops on the GPU: counterparts of
`data_loaders/humanml/scripts/motion_process.py:444-461` (`recover_from_ric`) and the `inv_transform` that precedes it
in the reference's scripts (dataset.py:478-479), as one native launch -- batched sampling can emit joint positions
without the `.cpu()` round trip the scripts make."""
import torch

from .. import _native as N


def _f32(t, dev):
    return torch.as_tensor(t, dtype=torch.float32).to(dev).contiguous()


def recover_joints(sample, mean, std, joints_num):
    """sample: [B, F, 1, T] normalised hml_vec GPU tensor (what p_sample_loop / ddim_sample_loop return);
    mean, std: [F].  Returns joint positions [B, 1, T, joints_num, 3] = recover_from_ric(inv_transform(permute(sample)))."""
    if not sample.is_cuda:
        raise RuntimeError("recover_joints runs on the GPU only (no CPU fallback); move the sample to cuda")
    x = sample.to(torch.float32).contiguous()
    B, F, one, T = x.shape
    assert one == 1, x.shape
    out = torch.empty(B, 1, T, joints_num, 3, dtype=torch.float32, device=x.device)
    m, s = _f32(mean, x.device), _f32(std, x.device)          # named: must outlive the launch (a freed temporary's block is
    N.check(N.lib().mst_recover_from_ric(N.ptr(x), N.ptr(m), N.ptr(s), B, F, T, joints_num,      # handed to the next allocation)
                                         N.ptr(out), N.stream_ptr(x.device)))
    return out


def recover_from_ric(data, joints_num):
    """Drop-in signature of the reference function: data [..., T, F] DENORMALISED rows -> [..., T, joints_num, 3]."""
    lead = data.shape[:-2]
    T, F = data.shape[-2:]
    x = data.reshape(-1, T, F).permute(0, 2, 1).unsqueeze(2)                 # [N, F, 1, T]
    out = recover_joints(x, torch.zeros(F), torch.ones(F), joints_num)       # [N, 1, T, J, 3]
    return out.reshape(*lead, T, joints_num, 3)
