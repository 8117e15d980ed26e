"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): CPU restatement of the post-sampling tensor ops.

Follows the reference line by line:
  inv_transform            data_loaders/humanml/data/dataset.py:478-479      data * std + mean
  recover_root_rot_pos     data_loaders/humanml/scripts/motion_process.py:389-410
  recover_from_ric         data_loaders/humanml/scripts/motion_process.py:444-461
  qrot                     data_loaders/humanml/common/quaternion.py:88-99
Pinned by tests/golden/post.npz (outputs of the reference functions themselves, tests/golden/make_golden_post.py)."""
import numpy as np


def qrot(q, v):
    s, u = q[..., :1], q[..., 1:]
    u, v = np.broadcast_arrays(u, v)
    uv = np.cross(u, v)
    uuv = np.cross(u, uv)
    return (v + 2 * (s * uv + uuv)).astype(np.float32)


def recover_root_rot_pos(data):
    rot_vel = data[..., 0]
    ang = np.zeros_like(rot_vel)
    ang[..., 1:] = rot_vel[..., :-1]
    ang = np.cumsum(ang, axis=-1, dtype=np.float32)
    quat = np.zeros(data.shape[:-1] + (4,), np.float32)
    quat[..., 0] = np.cos(ang)                      # sic: the full angle, not the half angle ("Revised by HL", :404)
    quat[..., 2] = np.sin(ang)
    pos = np.zeros(data.shape[:-1] + (3,), np.float32)
    pos[..., 1:, [0, 2]] = data[..., :-1, 1:3]
    pos = qrot(quat, pos)
    pos = np.cumsum(pos, axis=-2, dtype=np.float32)
    pos[..., 1] = data[..., 3]
    return quat, pos


def recover_from_ric(data, joints_num):
    """data: [..., T, F] denormalised hml_vec rows -> [..., T, joints_num, 3]."""
    data = np.asarray(data, np.float32)
    quat, r_pos = recover_root_rot_pos(data)
    positions = data[..., 4:(joints_num - 1) * 3 + 4]
    positions = positions.reshape(positions.shape[:-1] + (-1, 3))
    positions = qrot(np.broadcast_to(quat[..., None, :], positions.shape[:-1] + (4,)), positions)
    positions[..., 0] += r_pos[..., 0:1]
    positions[..., 2] += r_pos[..., 2:3]
    return np.concatenate([r_pos[..., None, :], positions], axis=-2)


def recover_joints(sample, mean, std, joints_num):
    """sample [B, F, 1, T] normalised -> [B, 1, T, J, 3]: permute + inv_transform + recover_from_ric."""
    den = (np.transpose(np.asarray(sample, np.float32), (0, 2, 3, 1)) * std + mean).astype(np.float32)
    return recover_from_ric(den, joints_num)
