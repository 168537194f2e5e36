"""ORACLE -- test infrastructure only, never the product path.

numpy restatement of the per-sample arithmetic of the reference's skeleton feeder,
/root/reference/feeder/feeder_nucla_gcn.py:85-130 (``__getitem__`` without the RGB part), as a pure function of the raw
clip, the drawn view parameters and the drawn frame indices.  Pinned by tests/golden/feeder.npz, which
tests/golden/make_golden_feeder.py produced by running the reference's own Feeder on synthetic clips
(tests/test_feeder_oracle.py).  Only tests/ may import this file."""
import math

import numpy as np

# reference :27-28 (1-based pairs)
BONE = [(1, 2), (2, 3), (3, 3), (4, 3), (5, 3), (6, 5), (7, 6), (8, 7), (9, 3), (10, 9), (11, 10),
        (12, 11), (13, 1), (14, 13), (15, 14), (16, 15), (17, 1), (18, 17), (19, 18), (20, 19)]


def rand_view_transform(X, agx, agy, s):
    """:75-83"""
    agx, agy = math.radians(agx), math.radians(agy)
    Rx = np.asarray([[1, 0, 0], [0, math.cos(agx), math.sin(agx)], [0, -math.sin(agx), math.cos(agx)]])
    Ry = np.asarray([[math.cos(agy), 0, -math.sin(agy)], [0, 1, 0], [math.sin(agy), 0, math.cos(agy)]])
    Ss = np.asarray([[s, 0, 0], [0, s, 0], [0, 0, s]])
    X0 = np.dot(np.reshape(X, (-1, 3)), np.dot(Ry, np.dot(Rx, Ss)))
    return np.reshape(X0, X.shape)


def transform(value, agx, agy, s, idx, stream='joint', time_steps=52):
    """value (L, 20, 3) float64 -> (3, time_steps, 20, 1) float32.  stream: 'joint' | 'bone' | 'motion' | 'bone_motion'
    ('bone_motion' = motion of bone; the reference's label-path dispatch never reaches it, :119-127)."""
    value = np.asarray(value, dtype=np.float64)
    center = value[0, 1, :]                                            # :98
    value = value - center
    sv = rand_view_transform(value, agx, agy, s)                       # :100
    sv = np.reshape(sv, (-1, 3))
    v_min, v_max = np.min(sv, axis=0), np.max(sv, axis=0)              # :102
    sv = (sv - v_min) / (v_max - v_min + 1e-6)
    sv = sv * 2 - 1
    sv = np.reshape(sv, (-1, 20, 3))
    data = np.zeros((time_steps, 20, 3))
    data[:, :, :] = sv[np.asarray(idx), :, :]                          # :113 / :117

    def bone(d):
        out = np.zeros_like(d)
        for v1, v2 in BONE:                                            # :121-122
            out[:, v1 - 1, :] = d[:, v1 - 1, :] - d[:, v2 - 1, :]
        return out

    def motion(d):
        out = np.zeros_like(d)
        out[:-1, :, :] = d[1:, :, :] - d[:-1, :, :]                    # :126
        return out

    if stream == 'bone':
        data = bone(data)
    elif stream == 'motion':
        data = motion(data)
    elif stream == 'bone_motion':
        data = motion(bone(data))
    data = np.transpose(data, (2, 0, 1))                               # :129
    return np.reshape(data, (3, time_steps, 20, 1)).astype(np.float32)


def val_indices(length, time_steps=52):
    return np.linspace(0, length - 1, time_steps).astype(int)          # :116
