"""graph.* parity: bit-exact against arrays generated from the reference."""
import os

import numpy as np

from tam_gcn_amd.graph import ucla, ntu_rgb_d, synthetic, tools

GOLD = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'graphs.npz'))


def test_ucla_bit_exact():
    A = ucla.Graph(labeling_mode='spatial').A
    assert A.dtype == np.float64 and A.shape == (3, 20, 20)
    assert np.array_equal(A, GOLD['ucla'])
    assert [int((A[i] != 0).sum()) for i in range(3)] == [20, 19, 19]


def test_ntu_bit_exact():
    A = ntu_rgb_d.Graph().A
    assert A.shape == (3, 25, 25)
    assert np.array_equal(A, GOLD['ntu'])
    assert [int((A[i] != 0).sum()) for i in range(3)] == [25, 24, 24]


def test_synthetic_graph_recipe():
    g = synthetic.Graph(num_node=64)
    assert g.A.shape == (3, 64, 64)
    assert np.array_equal(g.A[0], np.eye(64))
    col = g.A[1].sum(0)
    assert np.allclose(col[col > 0], 1.0)            # column-normalised


def test_synthetic_graph_equals_reference_recipe():
    """The build's 64-node tree through OUR get_spatial_graph == the same parent table through the reference's
    graph.tools.get_spatial_graph (fixture written by tests/golden/make_golden.py)."""
    import os
    gold = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'graphs.npz'))
    assert np.array_equal(synthetic.Graph().A, gold['syn64'])


def test_bad_labeling_mode_raises():
    import pytest
    with pytest.raises(ValueError):
        ucla.Graph(labeling_mode='uniform')


def test_edge2mat_orientation():
    A = tools.edge2mat([(0, 1)], 3)
    assert A[1, 0] == 1 and A.sum() == 1
