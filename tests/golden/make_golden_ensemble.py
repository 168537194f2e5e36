"""Golden-vector generator for the score-level ensemble (second half).  Runs ONLY in the build
container: imports the reference's own ensemble/ensemble_resnet_ctrgcn.py from /root/reference (read-only,
PYTHONDONTWRITEBYTECODE=1; its imports are argparse, pickle, numpy, tqdm, os) and calls ITS `ensemble_fusion`
(ensemble/ensemble_resnet_ctrgcn.py:11-61) on label / score files of this script's own making in a temporary directory.
The function returns nothing; what it computes leaves through stdout: one warning line per sample name missing from
either score file (:46-48) and the line with right_num/total_num (:60).  Both are parsed and written, with the inputs,
as DATA ONLY into tests/golden/ensemble.npz.  No reference source travels.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_ensemble.py

The pickles read back here were written a few lines above by this script (lists / dicts of numpy arrays): they are
not files of the reference.  Cases:
  * 464 samples x 10 classes (the N-UCLA validation split's size), fp32 scores, alpha in {0.5, 1.0, 2.0, 0.3};
  * names missing from the first file only, the second only, and both;
  * exact ties of the fused score between two or three classes (numpy.argmax takes the first), built from small
    integers so that a + alpha * b is exact in fp32 for every alpha used;
  * a label equal to the tied-for second class (so a wrong tie-break changes right_num).
"""
import contextlib
import io
import os
import pickle
import re
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = '/root/reference'
sys.dont_write_bytecode = True
sys.path.insert(0, os.path.join(REF, 'ensemble'))
import ensemble_resnet_ctrgcn as R            # noqa: E402   (the reference's script; its __main__ block does not run on import)

N, K = 464, 10
ALPHAS = (0.5, 1.0, 2.0, 0.3)
N_TIES = 24


def build_inputs():
    rng = np.random.default_rng(20240611)
    names = [f'a{(i % 10) + 1:02d}_s{(i // 10) % 10 + 1:02d}_e{i // 100:02d}_v{(i % 3) + 1:02d}_{i:03d}' for i in range(N)]
    labels = rng.integers(0, K, N)
    sa = (rng.standard_normal((N, K)) * 3).astype(np.float32)
    sb = (rng.standard_normal((N, K)) * 3).astype(np.float32)
    # exact ties: small integers (every product with 0.5 / 1 / 2 and every sum is exact in fp32; alpha = 0.3 breaks them by
    # rounding, identically in numpy and in any IEEE fp32 multiply-then-add)
    for j in range(N_TIES):
        i = 7 + 19 * j
        c1, c2 = int(rng.integers(0, K - 1)), 0
        c2 = int(rng.integers(c1 + 1, K))
        sa[i] = rng.integers(-6, 3, K).astype(np.float32)
        sb[i] = rng.integers(-6, 3, K).astype(np.float32) * 2
        sa[i, c1], sb[i, c1] = 8.0, 4.0            # fused: 8 + alpha * 4
        sa[i, c2], sb[i, c2] = 4.0, 6.0            # fused: 4 + alpha * 6   (= at alpha 2; c1 wins below, c2 above)
        if j % 3 == 0:                             # three-way tie at alpha = 2 on some rows
            c3 = (c2 + 1) % K
            if c3 not in (c1, c2):
                sa[i, c3], sb[i, c3] = 12.0, 2.0   # 12 + alpha * 2
        labels[i] = c2 if j % 2 else c1            # a wrong tie-break flips right_num
    miss_a = {i for i in range(N) if i % 37 == 5}
    miss_b = {i for i in range(N) if i % 41 == 7}
    miss_a.add(48); miss_b.add(48)                 # missing from both
    return names, labels, sa, sb, sorted(miss_a), sorted(miss_b)


def main():
    names, labels, sa, sb, miss_a, miss_b = build_inputs()
    out = dict(names=np.array(names), labels=labels.astype(np.int64), score_a=sa, score_b=sb,
               missing_a=np.array(miss_a, np.int64), missing_b=np.array(miss_b, np.int64), alphas=np.array(ALPHAS, np.float64))
    with tempfile.TemporaryDirectory() as d:
        lp, ap, bp = (os.path.join(d, f) for f in ('val_label.pkl', 'resnet_score.pkl', 'ctrgcn_score.pkl'))
        with open(lp, 'wb') as f:
            pickle.dump((list(names), [int(l) for l in labels]), f)          # the label file's layout: (sample_names, labels), :38-39
        with open(ap, 'wb') as f:
            pickle.dump({n: sa[i] for i, n in enumerate(names) if i not in set(miss_a)}, f)
        with open(bp, 'wb') as f:
            pickle.dump({n: sb[i] for i, n in enumerate(names) if i not in set(miss_b)}, f)
        for alpha in ALPHAS:
            buf, err = io.StringIO(), io.StringIO()
            with contextlib.redirect_stdout(buf), contextlib.redirect_stderr(err):      # tqdm draws on stderr
                R.ensemble_fusion(lp, ap, bp, alpha)
            text = buf.getvalue()
            m = re.search(r'(\d+)/(\d+)', text.split('=' * 40)[1])
            right, total = int(m.group(1)), int(m.group(2))
            acc = float(re.search(r'Top-1 Accuracy: ([0-9.]+)', text).group(1))
            # the warning line reads "... mẫu <name> trong file kết quả."
            warned = [re.search(r'mẫu (\S+) trong', ln).group(1) for ln in text.splitlines() if 'Không tìm thấy mẫu' in ln]
            out[f'alpha{alpha}/right_total'] = np.array([right, total], np.int64)
            out[f'alpha{alpha}/acc4'] = np.array(acc)
            out[f'alpha{alpha}/skipped'] = np.array(warned)
            print(f'alpha {alpha}: {right}/{total} (acc {acc}), {len(warned)} names skipped')
    np.savez_compressed(os.path.join(HERE, 'ensemble.npz'), **out)
    print('wrote ensemble.npz')


if __name__ == '__main__':
    main()
