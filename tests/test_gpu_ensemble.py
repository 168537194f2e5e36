"""-m gpu: the score-level ensemble (tamgcn_score_fuse behind tam_gcn_amd.ensemble) against (1) tests/golden/ensemble.npz = what
the REFERENCE's own ensemble_fusion (ensemble/ensemble_resnet_ctrgcn.py:11-61) printed for synthetic label / score files
(tests/golden/make_golden_ensemble.py), and (2) the numpy restatement oracle/ensemble_oracle.py, which
tests/test_ensemble_oracle.py holds to the same fixture."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import ensemble_oracle as EO          # noqa: E402
from tam_gcn_amd import ensemble as E              # noqa: E402


def _scores(n, k, seed):
    rng = np.random.default_rng(seed)
    return (rng.standard_normal((n, k)) * 3).astype(np.float32)


def test_raw_fusion_matches_the_reference_run():
    """right_num / total_num and the skipped names as the reference script printed them: four weights, names missing from
    either score file (or both), exact two- and three-way ties of the fused score (numpy.argmax: first maximum)."""
    import os
    G = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'ensemble.npz'))
    names = [str(n) for n in G['names']]
    labels, sa, sb = G['labels'], G['score_a'], G['score_b']
    ma, mb = set(G['missing_a'].tolist()), set(G['missing_b'].tolist())
    ra = {n: sa[i] for i, n in enumerate(names) if i not in ma}
    rb = {n: sb[i] for i, n in enumerate(names) if i not in mb}
    for alpha in (float(a) for a in G['alphas']):
        got = E.ensemble_by_name([ra, rb], [1.0, alpha], names, labels, softmax=False)
        assert [got['correct'], got['total']] == G[f'alpha{alpha}/right_total'].tolist()
        assert f"{got['acc']:.4f}" == f"{float(G[f'alpha{alpha}/acc4']):.4f}"
        assert got['skipped'] == [str(x) for x in G[f'alpha{alpha}/skipped']]
        assert got['pred'] == EO.fuse_raw(ra, rb, alpha, names, labels)[3]


def test_raw_fusion_by_name_matches_the_script_loop():
    n, k = 464, 10
    names = [f'a{i:03d}' for i in range(n)]
    rng = np.random.default_rng(0)
    labels = rng.integers(0, k, n)
    sa, sb = _scores(n, k, 1), _scores(n, k, 2)
    ra = {nm: sa[i] for i, nm in enumerate(names) if i % 37 != 5}          # some samples missing from one file
    rb = {nm: sb[i] for i, nm in enumerate(names) if i % 41 != 7}
    for alpha in (0.0, 0.5, 1.0, 2.5):
        acc, right, total, pred = EO.fuse_raw(ra, rb, alpha, names, labels)
        got = E.ensemble_by_name([ra, rb], [1.0, alpha], names, labels, softmax=False)
        assert (got['correct'], got['total']) == (right, total) and got['acc'] == acc
        assert got['pred'] == pred
        assert sorted(got['skipped']) == sorted(nm for nm in names if nm not in ra or nm not in rb)


def test_softmax_fusion_and_per_class_accuracy():
    n, k = 1000, 60
    rng = np.random.default_rng(3)
    labels = rng.integers(0, k - 3, n)            # the last three classes have no sample: (0, 0, 0.0) rows
    sa, sb = _scores(n, k, 4), _scores(n, k, 5)
    alpha = 0.7
    ref = EO.fuse_softmax(sb, sa, alpha)           # resnet_norm + alpha * ctrgcn_norm
    fused, pred, stats = E.fuse([sb, sa], [1.0, alpha], softmax=True, labels=labels)
    torch.cuda.synchronize()
    assert np.abs(fused.cpu().numpy() - ref).max() <= 2e-6
    top2 = np.sort(ref, axis=1)[:, -2:]
    clear = (top2[:, 1] - top2[:, 0]) > 1e-5      # fp32 exp implementations may order a near tie differently
    assert clear.mean() > 0.99
    assert np.array_equal(pred.cpu().numpy()[clear], np.argmax(ref, axis=1)[clear])
    # per-class statistics: exactly those of the device's own predictions, and the script's on the fused scores
    acc, correct, total, cls = EO.compute_accuracy(fused.cpu().numpy(), labels, k)
    st = stats.cpu().numpy()
    assert int(st[:, 0].sum()) == correct and int(st[:, 1].sum()) == total == n
    for c in range(k):
        assert (int(st[c, 0]), int(st[c, 1])) == cls[c][:2]
    acc1, cor1, tot1, cls1 = E.compute_accuracy(sa, labels)
    assert (acc1, cor1, tot1) == EO.compute_accuracy(sa, labels, k)[:3] and cls1 == EO.compute_accuracy(sa, labels, k)[3]


def test_first_maximum_wins_ties_and_four_streams():
    """numpy.argmax returns the first maximum; four score sets (the 4-stream recipe) in one launch."""
    s = np.zeros((4, 5, 6), np.float32)
    s[:, 0, 2] = s[:, 0, 4] = 1.0                   # tie between classes 2 and 4 -> 2
    s[:, 1, :] = 0.25                               # all equal -> 0
    s[0, 2, 5] = 3.0; s[1, 2, 1] = 2.0              # weights decide
    fused, pred, _ = E.fuse(list(s), [1.0, 2.0, 1.0, 1.0])
    assert pred.cpu().tolist()[:3] == [2, 0, 1]
    assert np.allclose(fused.cpu().numpy(), (s * np.array([1, 2, 1, 1], np.float32)[:, None, None]).sum(0))
    with pytest.raises(ValueError):
        E.fuse(list(s), [1.0])


def test_out_of_range_label_is_refused():
    """The kernel leaves a label outside [0, K) out of every class row while the scripts' accuracy divides by
    len(labels): the Python layer refuses such input instead of returning a different denominator."""
    s = _scores(8, 5, 9)
    for bad in (5, -1):
        lab = np.array([0, 1, 2, 3, 4, 0, 1, bad])
        with pytest.raises(ValueError, match='label outside'):
            E.fuse([s], [1.0], labels=lab)
        with pytest.raises(ValueError, match='label outside'):
            E.compute_accuracy(s, lab)
