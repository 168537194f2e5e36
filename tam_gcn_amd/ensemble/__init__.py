"""Score-level ensemble of the evaluation scripts on the GPU.

Mirrors what the reference's two ensemble scripts compute once every model has produced its per-sample class scores:

    ensemble/ensemble_resnet_ctrgcn.py:42-61          final = score_a + alpha * score_b on the RAW scores, samples matched by
                                                      name, names missing from either file skipped, top-1 accuracy
    ensemble/ensemble_ctrgcn_resnet_eval.py:99-108    the same on softmax-normalised scores
    ensemble/ensemble_ctrgcn_resnet_eval.py:217-234   compute_accuracy: overall + per-class (correct, total, ratio)

Only the arithmetic runs on the device (one launch of tamgcn_score_fuse for any number of score sets); matching names is
host bookkeeping, as in the reference.  Reading / writing the score pickles, the models' inference loops and the plots are
the scripts' own business (out of scope)."""
import numpy as np
import torch

from .. import ops

__all__ = ['fuse', 'compute_accuracy', 'ensemble_by_name']


def fuse(scores, weights, softmax=False, labels=None):
    """scores: sequence of S (N, K) arrays / tensors (one per model or stream), weights: S floats.
    -> fused (N, K) f32, pred (N,) int64, class_stats (K, 2) int32 [correct, total] or None -- HIP tensors."""
    dev = torch.device('cuda', torch.cuda.current_device())
    st = torch.stack([torch.as_tensor(np.asarray(s) if not torch.is_tensor(s) else s, dtype=torch.float32).to(dev) for s in scores]).contiguous()
    if st.dim() != 3 or len(weights) != st.shape[0]:
        raise ValueError('fuse: need S score sets of one (N, K) shape and S weights')
    w = torch.tensor([float(a) for a in weights], dtype=torch.float32, device=dev)
    lab = None
    if labels is not None:
        lab = torch.as_tensor(np.asarray(labels) if not torch.is_tensor(labels) else labels).to(dev, torch.int64).contiguous()
        if lab.shape != (st.shape[1],):
            raise ValueError('fuse: one label per sample')
        if bool(((lab < 0) | (lab >= st.shape[2])).any()):         # the kernel would leave it out of every class row, while
            raise ValueError('fuse: label outside [0, K)')         # the scripts' accuracy divides by len(labels)
    return ops.score_fuse(st, w, softmax, lab)


def compute_accuracy(scores, labels):
    """(acc, correct, total, {class: (correct, total, ratio)}) of one score set -- the reference's compute_accuracy."""
    scores = scores if torch.is_tensor(scores) else np.asarray(scores)
    _, _, stats = fuse([scores], [1.0], softmax=False, labels=labels)
    st = stats.cpu().numpy()
    correct, total = int(st[:, 0].sum()), int(st[:, 1].sum())
    cls = {c: ((int(st[c, 0]), int(st[c, 1]), st[c, 0] / st[c, 1]) if st[c, 1] > 0 else (0, 0, 0.0)) for c in range(st.shape[0])}
    return correct / total, correct, total, cls


def ensemble_by_name(score_dicts, weights, names, labels, softmax=False):
    """score_dicts: S mappings name -> (K,) scores; names / labels: the label file's two lists.  Samples missing from any
    mapping are skipped (the reference prints a warning and continues, ensemble_resnet_ctrgcn.py:46-48).
    -> dict(acc, correct, total, pred {name: class}, skipped [names], class_acc)"""
    keep = [i for i, n in enumerate(names) if all(n in d for d in score_dicts)]
    skipped = [n for n in names if not all(n in d for d in score_dicts)]
    if not keep:
        raise ValueError('ensemble_by_name: no sample is present in every score set')
    mats = [np.stack([np.asarray(d[names[i]], dtype=np.float32) for i in keep]) for d in score_dicts]
    lab = np.asarray([int(labels[i]) for i in keep], dtype=np.int64)
    _, pred, stats = fuse(mats, weights, softmax=softmax, labels=lab)
    pred = pred.cpu().numpy()
    st = stats.cpu().numpy()
    correct, total = int(st[:, 0].sum()), int(st[:, 1].sum())
    cls = {c: ((int(st[c, 0]), int(st[c, 1]), st[c, 0] / st[c, 1]) if st[c, 1] > 0 else (0, 0, 0.0)) for c in range(st.shape[0])}
    return dict(acc=correct / total, correct=correct, total=total, pred={names[i]: int(p) for i, p in zip(keep, pred)},
                skipped=skipped, class_acc=cls)
