"""CPU: oracle/ensemble_oracle.py against tests/golden/ensemble.npz, which tests/golden/make_golden_ensemble.py produced by
calling the REFERENCE's own ensemble_fusion (ensemble/ensemble_resnet_ctrgcn.py:11-61) on synthetic label / score files:
right_num / total_num, the printed 4-digit accuracy and the list of skipped sample names, for four weights, with names
missing from either file and exact ties of the fused score."""
import os

import numpy as np
import pytest

from oracle import ensemble_oracle as EO

GOLD = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'ensemble.npz'))


def load_case():
    names = [str(n) for n in GOLD['names']]
    labels = GOLD['labels']
    sa, sb = GOLD['score_a'], GOLD['score_b']
    ma, mb = set(GOLD['missing_a'].tolist()), set(GOLD['missing_b'].tolist())
    ra = {n: sa[i] for i, n in enumerate(names) if i not in ma}
    rb = {n: sb[i] for i, n in enumerate(names) if i not in mb}
    return names, labels, ra, rb


@pytest.mark.parametrize('alpha', [float(a) for a in GOLD['alphas']])
def test_fuse_raw_matches_the_reference_run(alpha):
    names, labels, ra, rb = load_case()
    acc, right, total, pred = EO.fuse_raw(ra, rb, alpha, names, labels)
    assert [right, total] == GOLD[f'alpha{alpha}/right_total'].tolist()
    assert f'{acc:.4f}' == f'{float(GOLD[f"alpha{alpha}/acc4"]):.4f}'
    skipped = [n for n in names if n not in ra or n not in rb]
    assert skipped == [str(s) for s in GOLD[f'alpha{alpha}/skipped']]
    assert set(pred) == set(names) - set(skipped)


def test_fixture_really_contains_ties():
    """At alpha = 2 the fixture's constructed rows tie exactly between >= 2 classes: the count that pins numpy.argmax's
    first-maximum rule (a last-maximum rule gives a different right_num)."""
    names, labels, ra, rb = load_case()
    ties = 0
    right_last = 0
    for i, n in enumerate(names):
        if n in ra and n in rb:
            f = ra[n] + 2.0 * rb[n]
            ties += int((f == f.max()).sum() > 1)
            right_last += int(len(f) - 1 - int(np.argmax(f[::-1])) == int(labels[i]))
    assert ties >= 20
    assert right_last != int(GOLD['alpha2.0/right_total'][0])
