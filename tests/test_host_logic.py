"""Host-side logic that needs no GPU: the per-thread reduce batch (nn.DataParallel drives one autograd thread per
device), split-mode switch of the C ABI."""
import threading

import torch

from tam_gcn_amd import ops


def test_reduce_batch_is_per_thread():
    """Two threads inside `with ReduceBatch()` at the same time: each one's deferred reductions stay in its own batch
    (a process-global active batch would let device 0's slabs be launched on device 1's stream)."""
    barrier = threading.Barrier(2)
    seen, errors = {}, []

    def worker(tag, n_items):
        try:
            with ops.ReduceBatch() as rb:
                barrier.wait(timeout=10)                      # both batches are active now
                assert ops.ReduceBatch.active() is rb
                for i in range(n_items):
                    part = torch.full((3, 4), float(tag * 100 + i))
                    ops._reduce_piece(part, 3, 4, 0, 4, 1.0, torch.empty(4))
                barrier.wait(timeout=10)                      # the other thread has registered its items too
                assert ops.ReduceBatch.active() is rb
                seen[tag] = [float(it[0][0, 0]) for it in rb.items]
                rb.items.clear()                              # nothing to launch on a CPU-only box
            assert ops.ReduceBatch.active() is None
        except Exception as e:                                # noqa: BLE001
            errors.append((tag, repr(e)))

    ts = [threading.Thread(target=worker, args=(1, 5)), threading.Thread(target=worker, args=(2, 7))]
    for t in ts:
        t.start()
    for t in ts:
        t.join(20)
    assert not errors, errors
    assert seen[1] == [100.0 + i for i in range(5)]
    assert seen[2] == [200.0 + i for i in range(7)]
    assert ops.ReduceBatch.active() is None


def test_reduce_batch_nests_and_restores():
    with ops.ReduceBatch() as outer:
        with ops.ReduceBatch() as inner:
            assert ops.ReduceBatch.active() is inner
        assert ops.ReduceBatch.active() is outer
    assert ops.ReduceBatch.active() is None


def test_split_mode_switch():
    from tam_gcn_amd import _lib
    lib = _lib.load()
    m0 = lib.tamgcn_get_split_mode()
    assert m0 in (0, 1, 2)
    assert lib.tamgcn_set_split_mode(0) == 0 and lib.tamgcn_get_split_mode() == 0
    assert lib.tamgcn_set_split_mode(7) < 0 and b'tamgcn_set_split_mode' in lib.tamgcn_last_error()
    assert lib.tamgcn_set_split_mode(m0) == 0


def test_ensemble_oracle_hand_case():
    """oracle/ensemble_oracle.py on a case small enough to check by hand (the GPU path is compared with it in
    tests/test_gpu_ensemble.py)."""
    import numpy as np
    from oracle import ensemble_oracle as EO
    names = ['s0', 's1', 's2', 's3']
    labels = [1, 0, 2, 1]
    ra = {'s0': np.float32([0, 1, 0]), 's1': np.float32([0, 0, 3]), 's2': np.float32([1, 0, 0])}          # s3 missing
    rb = {'s0': np.float32([2, 0, 0]), 's1': np.float32([2, 0, 0]), 's2': np.float32([0, 0, 4]), 's3': np.float32([0, 9, 0])}
    acc, right, total, pred = EO.fuse_raw(ra, rb, 0.25, names, labels)
    assert (right, total) == (1, 3) and pred == {'s0': 1, 's1': 2, 's2': 0}       # s2: [1, 0, 1] ties -> first maximum
    acc, right, total, pred = EO.fuse_raw(ra, rb, 1.0, names, labels)
    assert pred == {'s0': 0, 's1': 2, 's2': 2} and (right, total) == (1, 3)
    sc = np.float32([[0, 1, 0], [3, 0, 0], [0, 0, 1], [0, 1, 0]])
    acc, correct, total, cls = EO.compute_accuracy(sc, np.asarray(labels), 4)
    assert (correct, total) == (4, 4) and cls[1] == (2, 2, 1.0) and cls[3] == (0, 0, 0.0)
    f = EO.fuse_softmax(sc, sc, 1.0)
    assert np.allclose(f.sum(1), 2.0, atol=1e-6)
