"""Golden vectors for the skeleton feeder's per-sample contract.  Runs ONLY in the build container.

The reference's feeder/feeder_nucla_gcn.py imports torchvision.transforms and rarfile, which are not installed; both
are used only for the RGB transform object (:43-47), never in the skeleton arithmetic, so they are stubbed in
sys.modules for the import.  The dataset is absent: every entry of the reference's own split lists gets
a synthetic JSON clip {"skeletons": [[[x, y, z] x 20] x length]} of the listed length in a temporary directory.  What is
stored is data only: for a handful of samples the raw clip, the label, and the reference's outputs on the val path
(joint / bone / motion label paths) and on the train path under random.seed(s).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_feeder.py
"""
import json
import os
import random
import sys
import tempfile
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = '/root/reference'
sys.dont_write_bytecode = True
sys.path.insert(0, REF)

tv = types.ModuleType('torchvision')
tvt = types.ModuleType('torchvision.transforms')
for name in ('Compose', 'Resize', 'ToTensor', 'Normalize'):
    setattr(tvt, name, lambda *a, **k: None)
tv.transforms = tvt
sys.modules.setdefault('torchvision', tv)
sys.modules.setdefault('torchvision.transforms', tvt)
sys.modules.setdefault('rarfile', types.ModuleType('rarfile'))

from feeder import feeder_nucla_gcn as RF      # noqa: E402  (reference's)


def clip(name, length):
    r = np.random.RandomState(sum(map(ord, name)) * 7919 % (2 ** 31 - 1))
    base = r.uniform(-1.0, 1.0, size=(1, 20, 3)) * np.array([0.6, 0.9, 0.3]) + np.array([0.1, 0.0, 2.5])
    walk = np.cumsum(r.normal(0, 0.02, size=(length, 20, 3)), axis=0)
    return base + walk


def main():
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        # the reference reads its hard-coded split lists in __init__ (load_data touches every entry)
        probe = RF.Feeder.__new__(RF.Feeder)
        for label_path in ('val', 'train'):
            try:
                RF.Feeder.__init__(probe, tmp, label_path)
            except FileNotFoundError:
                pass
            for info in probe.data_dict:
                d = os.path.join(tmp, info['file_name'])
                if not os.path.isdir(d):
                    os.makedirs(d)
                    with open(os.path.join(d, info['file_name'] + '.json'), 'w') as f:
                        json.dump({'skeletons': clip(info['file_name'], int(info['length'])).tolist()}, f)
        picks = {'val': [0, 7, 123, 463], 'train': [0, 11, 500, 1019]}
        for label_path, stream in (('val', 'joint'), ('val_bone', 'bone'), ('val_motion', 'motion'), ('val_bone_motion', 'bone'),
                                   ('train', 'joint'), ('train_bone', 'bone'), ('train_motion', 'motion')):
            fd = RF.Feeder(tmp, label_path)
            split = 'val' if 'val' in label_path else 'train'
            out[f'{label_path}/len'] = np.array(len(fd))
            for i in picks[split]:
                random.seed(1000 + i)
                data, rgb, label, index = fd[i]
                key = f'{label_path}/{i}'
                out[key + '/data'] = np.asarray(data)
                out[key + '/label'] = np.array(label)
                out[key + '/name'] = np.array(fd.data_dict[i]['file_name'])
                out[f'{split}/{i}/raw'] = np.asarray(fd.data[i], dtype=np.float64)
                assert index == i and tuple(rgb.shape) == (3, 299, 299) and data.dtype == np.float32
            out[f'{label_path}/stream'] = np.array(stream)
    np.savez_compressed(os.path.join(HERE, 'feeder.npz'), **out)
    print('feeder.npz', len(out), 'arrays')


if __name__ == '__main__':
    main()
