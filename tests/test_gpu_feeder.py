"""-m gpu: the GPU feeder (tam_gcn_amd.feeder.feeder_nucla_gcn.Feeder -> tamgcn_feeder_transform) and the stream
derivation kernel against vectors the reference's own Feeder produced (tests/golden/feeder.npz) and the numpy oracle.
Val path (identity view): bit-exact.  Train path (random view matrix): <= 2e-6 absolute on [-1, 1] data (the 3x3
rotation is an fp64 dot product whose summation order numpy's BLAS does not specify)."""
import json
import os
import random

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import feeder_oracle as FO                                  # noqa: E402
from tam_gcn_amd import ops                                             # noqa: E402
from tam_gcn_amd.feeder.feeder_nucla_gcn import Feeder, BONE_PARENT     # noqa: E402

GOLD = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'feeder.npz'))
PICKS = {'val': [0, 7, 123, 463], 'train': [0, 11, 500, 1019]}


def _dataset(tmp_path, split):
    dd = []
    for i in PICKS[split]:
        name = str(GOLD[f'{split}/{i}/name'])
        os.makedirs(tmp_path / name, exist_ok=True)
        with open(tmp_path / name / (name + '.json'), 'w') as f:
            json.dump({'skeletons': GOLD[f'{split}/{i}/raw'].tolist()}, f)
        dd.append({'file_name': name, 'label': int(GOLD[f'{split}/{i}/label']) + 1})
    return dd


@pytest.mark.parametrize('label_path', ['val', 'val_bone', 'val_motion', 'val_bone_motion'])
def test_val_path_bit_exact(label_path, tmp_path):
    dd = _dataset(tmp_path, 'val')
    fd = Feeder(str(tmp_path), label_path, data_dict=dd)
    assert len(fd) == 4
    for k, i in enumerate(PICKS['val']):
        data, rgb, label, index = fd[k]
        ref = GOLD[f'{label_path}/{i}/data']
        assert data.dtype == np.float32 and data.shape == (3, 52, 20, 1) and tuple(rgb.shape) == (3, 299, 299)
        assert label == int(GOLD[f'val/{i}/label']) and index == k
        assert np.array_equal(data, ref), f'{label_path}/{i}: max diff {np.abs(data - ref).max()}'
    out, lab, _ = fd.batch(range(4))                       # the batch form: one launch, tensors on the device
    assert out.is_cuda and tuple(out.shape) == (4, 3, 52, 20, 1) and lab.tolist() == [int(GOLD[f'val/{i}/label']) for i in PICKS['val']]
    for k, i in enumerate(PICKS['val']):
        assert np.array_equal(out[k].cpu().numpy(), GOLD[f'{label_path}/{i}/data'])


@pytest.mark.parametrize('label_path', ['train', 'train_bone', 'train_motion'])
def test_train_path_same_rng_consumption(label_path, tmp_path):
    dd = _dataset(tmp_path, 'train')
    fd = Feeder(str(tmp_path), label_path, data_dict=dd)
    for k, i in enumerate(PICKS['train']):
        random.seed(1000 + i)
        data, _, label, _ = fd[k]
        ref = GOLD[f'{label_path}/{i}/data']
        assert np.abs(data - ref).max() <= 2e-6, f'{label_path}/{i}: max diff {np.abs(data - ref).max()}'
        assert label == int(GOLD[f'train/{i}/label'])


def test_bone_motion_stream_and_derivation_kernel(tmp_path):
    """Motion-of-bone (asked for explicitly) against the oracle, and the 4-stream derivation from a resident joint batch
    (tamgcn_stream_derive) against the same formulas in torch."""
    dd = _dataset(tmp_path, 'val')
    fd = Feeder(str(tmp_path), 'val', data_dict=dd, stream='bone_motion')
    for k, i in enumerate(PICKS['val']):
        raw = GOLD[f'val/{i}/raw']
        want = FO.transform(raw, 0, 0, 1.0, FO.val_indices(len(raw)), 'bone_motion')
        assert np.array_equal(fd[k][0], want)
    dev = torch.device('cuda:0')
    g = torch.Generator().manual_seed(3)
    for shape in [(5, 3, 52, 20, 1), (3, 3, 16, 25, 2)]:
        V = shape[3]
        x = (torch.rand(*shape, generator=g) * 2 - 1).to(dev)
        parent = torch.tensor((list(BONE_PARENT) + [3, 7, 0, 24, 11])[:V], dtype=torch.int32, device=dev)
        pl = parent.long()
        bone = x - x[:, :, :, pl, :]
        motion = torch.zeros_like(x)
        motion[:, :, :-1] = x[:, :, 1:] - x[:, :, :-1]
        bm = torch.zeros_like(x)
        bm[:, :, :-1] = bone[:, :, 1:] - bone[:, :, :-1]
        assert torch.equal(ops.stream_derive(x, parent, 'bone'), bone)
        assert torch.equal(ops.stream_derive(x, parent, 'motion'), motion)
        assert torch.equal(ops.stream_derive(x, parent, 'bone_motion'), bm)
        assert ops.stream_derive(x, parent, 'joint') is x
    # the derived bone stream of the feeder's joint output equals the feeder's own bone output to fp32 rounding
    fj, fb = Feeder(str(tmp_path), 'val', data_dict=dd), Feeder(str(tmp_path), 'val_bone', data_dict=dd)
    xj, _, _ = fj.batch(range(4))
    xb, _, _ = fb.batch(range(4))
    assert (ops.stream_derive(xj, fj._parent, 'bone') - xb).abs().max() <= 3e-7
