"""TEST INFRASTRUCTURE ONLY (tests/, __graft_entry__.smoke(), bench.py's cpu_baseline): numpy restatement of the reference's
score-level ensemble.

    fuse_raw       ensemble/ensemble_resnet_ctrgcn.py:42-61   -- per name: score_a + alpha * score_b, numpy.argmax, top-1 count;
                                                                 names missing from either mapping are skipped
    fuse_softmax   ensemble/ensemble_ctrgcn_resnet_eval.py:99-108 -- scipy.special.softmax(axis=1) of each set, then a + alpha * b
    compute_accuracy   ensemble/ensemble_ctrgcn_resnet_eval.py:217-234

Pinning: fuse_raw is PINNED -- tests/golden/make_golden_ensemble.py imports the reference's ensemble_resnet_ctrgcn.py (numpy,
tqdm, pickle: all present here), runs its ensemble_fusion on synthetic label / score files and stores right_num / total_num,
the printed accuracy and the skipped names (tests/golden/ensemble.npz; tests/test_ensemble_oracle.py).  fuse_softmax and
compute_accuracy restate ensemble_ctrgcn_resnet_eval.py, whose module imports seaborn and torchvision (absent here: ordinary
ModuleNotFoundError), so those two formulas stay "unpinned beyond the numpy / scipy calls they share with the script"
(DESIGN.md section 4)."""
import numpy as np
from scipy.special import softmax


def fuse_raw(r_a, r_b, alpha, names, labels):
    right = total = 0
    pred = {}
    for i in range(len(names)):
        name, l = names[i], int(labels[i])
        if name not in r_a or name not in r_b:
            continue
        final = r_a[name] + (alpha * r_b[name])
        p = int(np.argmax(final))
        pred[name] = p
        right += int(p == l)
        total += 1
    return right / total, right, total, pred


def compute_accuracy(scores, labels, num_class):
    preds = np.argmax(scores, axis=1)
    correct = int((preds == labels).sum())
    total = len(labels)
    cls = {}
    for c in range(num_class):
        mask = labels == c
        if mask.sum() > 0:
            cc, ct = int((preds[mask] == labels[mask]).sum()), int(mask.sum())
            cls[c] = (cc, ct, cc / ct)
        else:
            cls[c] = (0, 0, 0.0)
    return correct / total, correct, total, cls


def fuse_softmax(a, b, alpha):
    return softmax(a, axis=1) + alpha * softmax(b, axis=1)
