"""CPU: the numpy restatement of the reference feeder's per-sample arithmetic (oracle/feeder_oracle.py) against the
vectors the reference's own Feeder produced (tests/golden/feeder.npz, written by tests/golden/make_golden_feeder.py)."""
import os
import random

import numpy as np
import pytest

from oracle import feeder_oracle as FO

GOLD = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'feeder.npz'))
PICKS = {'val': [0, 7, 123, 463], 'train': [0, 11, 500, 1019]}
PATHS = [('val', 'joint'), ('val_bone', 'bone'), ('val_motion', 'motion'), ('val_bone_motion', 'bone'),
         ('train', 'joint'), ('train_bone', 'bone'), ('train_motion', 'motion')]


def draw(split, i, length):
    """The reference's RNG consumption for sample i under random.seed(1000 + i) (feeder_nucla_gcn.py:88-93, :110-117)."""
    if split == 'val':
        return 0, 0, 1.0, FO.val_indices(length)
    random.seed(1000 + i)
    agx, agy, s = random.randint(-60, 60), random.randint(-60, 60), random.uniform(0.5, 1.5)
    idx = random.sample(list(np.arange(length)) * 100, 52)
    idx.sort()
    return agx, agy, s, idx


@pytest.mark.parametrize('label_path,stream', PATHS)
def test_oracle_reproduces_reference_feeder(label_path, stream):
    split = 'val' if 'val' in label_path else 'train'
    assert str(GOLD[f'{label_path}/stream']) == stream
    for i in PICKS[split]:
        raw = GOLD[f'{split}/{i}/raw']
        agx, agy, s, idx = draw(split, i, len(raw))
        got = FO.transform(raw, agx, agy, s, idx, stream)
        ref = GOLD[f'{label_path}/{i}/data']
        assert got.shape == ref.shape == (3, 52, 20, 1) and got.dtype == np.float32
        assert np.array_equal(got, ref), f'{label_path}/{i}: max diff {np.abs(got - ref).max()}'
        if stream == 'joint':
            assert got.min() >= -1.0 and got.max() <= 1.0


def test_bone_motion_label_path_collapses_to_bone():
    """The reference's `elif` (feeder_nucla_gcn.py:119-127): a label path with both words yields the bone stream."""
    for i in PICKS['val']:
        assert np.array_equal(GOLD[f'val_bone_motion/{i}/data'], GOLD[f'val_bone/{i}/data'])


def test_labels_follow_the_action_id():
    from tam_gcn_amd.feeder.feeder_nucla_gcn import ACTION_TO_LABEL
    for split in ('val', 'train'):
        for i in PICKS[split]:
            name = str(GOLD[f'{split}/{i}/name'])
            assert int(GOLD[f'{split}/{i}/label']) == ACTION_TO_LABEL[int(name.split('_')[0][1:])] - 1
            assert (name.split('_')[3] == 'v03') == (split == 'val')          # cross-view split: camera 3 = val
    assert int(GOLD['val/len']) == 464 and int(GOLD['train/len']) == 1020
