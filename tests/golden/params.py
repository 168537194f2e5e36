"""Deterministic parameter / input generation shared by the golden generator
and the tests (numpy's frozen legacy RandomState => identical on every box).

Full-model fixtures would be ~7 MB per case if they carried the state-dict, so
instead both sides regenerate it from a seed: ``fill_state_(sd, seed)``
overwrites every tensor of a state-dict in sorted-key order with values of a
sensible scale (de-degenerated: alpha != 0, unit_gcn.bn.weight ~ 1, non-zero
offset_conv, perturbed PA, non-zero biases, non-trivial BN running stats).
"""
import numpy as np
import torch


def _rs(seed, key):
    h = 0
    for ch in key:                      # stable, python-hash independent
        h = (h * 131 + ord(ch)) % 1000003
    return np.random.RandomState((seed * 1000003 + h) % (2 ** 31 - 1))


def fill_state_(sd, seed):
    with torch.no_grad():
        for k in sorted(sd.keys()):
            v = sd[k]
            r = _rs(seed, k)
            last = k.split('.')[-1]
            if last == 'num_batches_tracked':
                v.fill_(3)
                continue
            shape = tuple(v.shape)
            n = r.standard_normal(shape).astype(np.float32)
            if last == 'alpha':
                a = (0.3 + 0.7 * r.random_sample(shape)).astype(np.float32)
            elif last == 'PA':
                a = v.detach().cpu().numpy().astype(np.float32) + 0.05 * n
            elif last == 'running_mean':
                a = 0.1 * n
            elif last == 'running_var':
                a = (0.5 + r.random_sample(shape)).astype(np.float32)
            elif last == 'bias':
                a = 0.05 * n
            elif last == 'weight' and v.dim() == 1:          # BN gamma
                a = 1 + 0.05 * n
            elif last == 'weight' and v.dim() == 2:          # fc
                a = n * (2.0 / shape[0]) ** 0.5 * 0.5
            elif last == 'weight' and v.dim() == 4:          # conv (O,I,k,1)
                fan_in = shape[1] * shape[2]
                a = n * (1.0 / fan_in) ** 0.5
            else:
                a = n
            v.copy_(torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).reshape(shape))
    return sd


def make_input(shape, seed, lo=-1.0, hi=1.0):
    r = np.random.RandomState(seed)
    return torch.from_numpy((lo + (hi - lo) * r.random_sample(shape)).astype(np.float32))


def make_labels(n, num_class, seed):
    return torch.from_numpy(np.random.RandomState(seed).randint(0, num_class, size=(n,)).astype(np.int64))


N_SAMPLE = 4096


def sample_indices(numel, key, k=N_SAMPLE):
    """k seeded flat indices into a tensor of `numel` elements (regenerable from the fixture key: only the VALUES are stored)."""
    return _rs(20241005, key).randint(0, numel, size=k).astype(np.int64)


def sample(t, key, k=N_SAMPLE):
    """Values of `t` at sample_indices(t.numel(), key): element-wise pin of a tensor too large to store (VERDICT r03 weak #3:
    sums + 16 elements would let a localised error in the interior pass)."""
    idx = torch.from_numpy(sample_indices(t.numel(), key, k))
    return t.detach().cpu().reshape(-1)[idx].numpy().copy()


def digest(t, k=8):
    """Compact fingerprint of a tensor: [sum, sum|x|, sum x^2, first k, last k]."""
    a = t.detach().cpu().double().reshape(-1)
    head = a[:k]
    tail = a[-k:]
    if head.numel() < k:
        head = torch.cat([head, torch.zeros(k - head.numel(), dtype=a.dtype)])
        tail = torch.cat([tail, torch.zeros(k - tail.numel(), dtype=a.dtype)])
    return torch.cat([torch.stack([a.sum(), a.abs().sum(), (a * a).sum()]), head, tail]).numpy()
