"""Graph data layer vs golden fixtures produced by RUNNING the reference's numpy/scipy code
(tests/golden/make_golden.py): rows a13-a16, a18 of SURVEY.md §8. Bit-exact (same float32 values, same entry order)."""
import numpy as np
import pytest
import torch

from gnnkeras_amd import GraphObject, GraphTensor, CompositeGraphObject, CompositeGraphTensor

MODES = ['sum', 'average', 'normalized']


def coo(m):
    m = m.tocoo()
    return np.stack([m.row.astype(np.float64), m.col.astype(np.float64), m.data.astype(np.float64)], axis=1)


def test_mutag_loader_matches_reference(golden, mutag_graphs):
    gs = mutag_graphs
    assert len(gs) == 4337
    stats = np.array([[g.nodes.shape[0], g.arcs.shape[0]] for g in gs])
    assert np.array_equal(stats, golden['mutag_stats'])
    assert stats[:, 0].sum() == 131488 and stats[:, 1].sum() == 266894
    assert np.array_equal(np.concatenate([g.targets for g in gs]), golden['mutag_targets'])
    ac = np.array([float(np.sum(g.arcs * np.arange(1, g.arcs.shape[1] + 1))) for g in gs])
    assert np.array_equal(ac, golden['mutag_arcs_checksum'])
    nc = np.array([float(np.sum(g.nodes * np.arange(1, g.nodes.shape[1] + 1)[None, :]
                                * np.arange(1, g.nodes.shape[0] + 1)[:, None])) for g in gs])
    assert np.array_equal(nc, golden['mutag_nodes_checksum'])


@pytest.mark.parametrize('mode', MODES)
def test_single_graph_operands(golden, mutag_graphs, mode):
    for i in golden['single_ids']:
        g = mutag_graphs[int(i)].copy()
        g.setAggregation(mode)
        p = f'single{i}_{mode}_'
        assert np.array_equal(golden[p + 'nodes'], g.nodes)
        assert np.array_equal(golden[p + 'arcs'], g.arcs)
        assert np.array_equal(golden[p + 'targets'], g.targets)
        assert np.array_equal(golden[p + 'ArcNode'], coo(g.ArcNode))
        assert np.array_equal(golden[p + 'Adjacency'], coo(g.Adjacency))
        assert np.array_equal(golden[p + 'NodeGraph'], coo(g.NodeGraph))
        assert tuple(golden[p + 'NodeGraph_shape']) == g.NodeGraph.shape


@pytest.mark.parametrize('mode', MODES)
def test_merge32(golden, mutag_graphs, mode):
    gl = [g.copy() for g in mutag_graphs[:32]]
    for g in gl: g.setAggregation(mode)
    m = GraphObject.merge(gl, focus='g', aggregation_mode=mode)
    p = f'merge32_{mode}_'
    assert m.nodes.shape[0] == 935 and m.arcs.shape[0] == 1922
    for key, val in [('nodes', m.nodes), ('arcs', m.arcs), ('targets', m.targets), ('set_mask', m.set_mask),
                     ('output_mask', m.output_mask), ('sample_weight', m.sample_weight), ('ArcNode', coo(m.ArcNode)),
                     ('Adjacency', coo(m.Adjacency)), ('NodeGraph', coo(m.NodeGraph))]:
        assert np.array_equal(golden[p + key], val), key
    assert tuple(golden[p + 'NodeGraph_shape']) == m.NodeGraph.shape == (935, 32)


def test_all_136_batches_sizes(golden, mutag_graphs):
    sizes = []
    for b in range(0, len(mutag_graphs), 32):
        m = GraphObject.merge(mutag_graphs[b:b + 32], focus='g', aggregation_mode='sum')
        sizes.append([m.nodes.shape[0], m.arcs.shape[0]])
    assert np.array_equal(np.array(sizes), golden['merge32_all_stats'])
    assert len(sizes) == 136


@pytest.mark.parametrize('mode', MODES)
def test_toy_masks_duplicates_isolated(golden, mode):
    g = GraphObject(nodes=golden['toy_nodes'], arcs=golden['toy_arcs'], targets=golden['toy_targets'][:4], focus='n',
                    set_mask=golden['toy_set_mask'], output_mask=golden['toy_output_mask'], sample_weight=2.5,
                    aggregation_mode=mode)
    p = f'toy_{mode}_'
    assert g.arcs.shape[0] == 8                                          # duplicate arc dropped
    assert np.array_equal(golden[p + 'arcs_out'], g.arcs)
    assert np.array_equal(golden[p + 'ArcNode'], coo(g.ArcNode))
    assert np.array_equal(golden[p + 'Adjacency'], coo(g.Adjacency))
    assert np.array_equal(golden[p + 'NodeGraph'], coo(g.NodeGraph))
    assert tuple(golden[p + 'NodeGraph_shape']) == g.NodeGraph.shape
    assert np.array_equal(golden[p + 'sample_weight'], g.sample_weight)
    # column sums of ArcNode: in-degree ('sum'), 1 ('average'), in-degree/#arcs ('normalized')  
    indeg = np.bincount(g.arc_ids[:, 1], minlength=7)
    colsum = np.asarray(g.ArcNode.sum(axis=0)).reshape(-1)
    expect = {'sum': indeg, 'average': (indeg > 0).astype(float), 'normalized': indeg / 8}[mode]
    assert np.allclose(colsum, expect, atol=1e-6)


@pytest.mark.parametrize('mode', MODES + ['composite_average'])
def test_composite_operands_and_merge(golden, mode):
    dims = tuple(int(i) for i in golden['ctoy_dim_node_label'])
    cgs = [CompositeGraphObject(nodes=golden[f'ctoy{t}_nodes'], arcs=golden[f'ctoy{t}_arcs'],
                                targets=golden[f'ctoy{t}_targets'], type_mask=golden[f'ctoy{t}_type_mask'],
                                dim_node_label=dims, focus='n', aggregation_mode=mode) for t in range(2)]
    for ti, cg in enumerate(cgs + [CompositeGraphObject.merge(cgs, focus='n', aggregation_mode=mode)]):
        p = f'ctoy{ti}_{mode}_'
        assert np.array_equal(golden[p + 'arcs_out'], cg.arcs)
        assert np.array_equal(golden[p + 'type_mask_out'], cg.type_mask)
        assert np.array_equal(golden[p + 'ArcNode'], coo(cg.ArcNode))
        assert np.array_equal(golden[p + 'Adjacency'], coo(cg.Adjacency))
        for t, ca in enumerate(cg.CompositeAdjacencies):
            assert np.array_equal(golden[p + f'CA{t}'], coo(ca)), (mode, ti, t)


def test_graph_tensor_sparse_order_and_csr(mutag_graphs):
    """COO2SparseTensor = row-major reorder (reference graph_class.py:551-560); the cached by-destination CSR is a
    stable regrouping of it: ascending source inside every destination."""
    m = GraphObject.merge(mutag_graphs[:8], focus='g', aggregation_mode='average')
    gt = GraphTensor.fromGraphObject(m, device='cpu')
    adj = gt.Adjacency
    assert np.array_equal(adj.indices, m.arc_ids)                       # arcs are (src, dst) sorted => same order
    csr = adj.csr()
    assert csr.w is None and csr.row_scale is not None                  # 'average' = one value per destination
    indeg = np.bincount(m.arc_ids[:, 1], minlength=m.nodes.shape[0])
    assert np.array_equal(np.diff(csr.rowptr), indeg)
    for j in range(m.nodes.shape[0]):
        s = csr.src[csr.rowptr[j]:csr.rowptr[j + 1]]
        assert np.all(np.diff(s) > 0)
        assert np.array_equal(np.sort(m.arc_ids[m.arc_ids[:, 1] == j, 0]), s)
    assert np.allclose(csr.row_scale[indeg > 0], 1.0 / indeg[indeg > 0])
    full = adj.csr(uniform_rows=False)
    assert full.w is not None and len(full.w) == adj.nnz
    an = gt.ArcNode.csr()
    assert an.n_src == m.arcs.shape[0] and an.n_dst == m.nodes.shape[0]
    ng = gt.NodeGraph.csr()
    assert ng.n_dst == 8 and ng.n_src == m.nodes.shape[0]


def test_composite_tensor_layout():
    rng = np.random.default_rng(0)
    n = 12
    tm = np.zeros((n, 2), dtype=bool); tm[np.arange(n), rng.integers(0, 2, n)] = True
    arcs = np.array([[i, (i + 1) % n, 1.0] for i in range(n)] + [[i, (i + 5) % n, 0.5] for i in range(n)]
                    + [[i, (i + 2) % n, 0.25] for i in range(n)])
    cg = CompositeGraphObject(nodes=rng.normal(size=(n, 3)), arcs=arcs, targets=rng.normal(size=(n, 2)), type_mask=tm,
                              dim_node_label=(3, 2), focus='n', aggregation_mode='composite_average')
    ct = CompositeGraphTensor.fromGraphObject(cg, device='cpu')
    assert tuple(ct.type_mask.shape) == (2, n)                          # transposed, composite_graph_class.py:263
    assert len(ct.CompositeAdjacencies) == 2
    assert sum(c.nnz for c in ct.CompositeAdjacencies) == ct.Adjacency.nnz
    vals = cg.ArcNode.data
    mixed = any(len(set(vals[cg.arc_ids[:, 1] == j])) > 1 for j in range(n))
    assert mixed and ct.Adjacency.csr().w is not None                   # composite_average needs per-arc weights


@pytest.mark.parametrize('rate', [0.5, 0.3])
def test_transductive_retyping_matches_reference(golden, rate):
    """`get_transduction` (reference TransductiveGraphSequencers.py:62-95) draws the same split from a seeded numpy RNG."""
    from gnnkeras_amd.Sequencers.TransductiveGraphSequencers import TransductiveMultiGraphSequencer
    g = GraphObject(nodes=golden['trans_nodes'], arcs=golden['trans_arcs'], targets=golden['trans_targets'], focus='n',
                    set_mask=golden['trans_set_mask'], output_mask=golden['trans_output_mask'], aggregation_mode='sum')
    np.random.seed(123)
    cg = TransductiveMultiGraphSequencer.get_transduction(g, rate, 'n', 'float32')
    p = f'trans_{int(rate * 10)}_'
    assert np.array_equal(cg.nodes, golden[p + 'nodes']) and np.array_equal(cg.targets, golden[p + 'targets'])
    assert np.array_equal(cg.type_mask, golden[p + 'type_mask'])
    assert np.array_equal(cg.output_mask, golden[p + 'output_mask']) and np.array_equal(cg.set_mask, golden[p + 'set_mask'])
    assert list(cg.DIM_NODE_LABEL) == [3, 5] and cg.targets.shape[0] == int(cg.output_mask.sum())


def test_transductive_sequencers_layout_and_epoch_resampling():
    from gnnkeras_amd.Sequencers.TransductiveGraphSequencers import (TransductiveMultiGraphSequencer,
                                                                      TransductiveSingleGraphSequencer)
    rng = np.random.default_rng(0)
    def hg(n):
        om = rng.random(n) < 0.8
        return GraphObject(rng.normal(size=(n, 3)), np.array([[i, (i + 1) % n, 1.] for i in range(n)]),
                           np.eye(2)[rng.integers(0, 2, int(om.sum()))], focus='n', output_mask=om)
    gs = [hg(10), hg(14), hg(9)]
    np.random.seed(0)
    seq = TransductiveMultiGraphSequencer(gs, 'n', 'average', 0.5, batch_size=2, shuffle=False, device='cpu')
    x, y, sw = seq[0]
    assert len(x) == 10 and tuple(x[2].reshape(-1).tolist()) == (3, 5) and x[0].shape[1] == 5
    assert tuple(x[3].shape) == (2, 24, 1) and y.shape[0] == int((x[4].reshape(-1) & x[5].reshape(-1)).sum())
    first = x[3].clone()
    seq.on_epoch_end()
    assert not torch.equal(seq[0][0][3], first)                       # a new transductive split every epoch
    c = seq.copy()
    assert len(c) == len(seq) and 'transductive' in repr(c) and c.get_config()['transductive_rate'] == 0.5
    single = TransductiveSingleGraphSequencer(hg(30), 'n', 0.4, batch_size=5, shuffle=True, device='cpu')
    assert len(single) >= 1 and len(single[0][0]) == 10
    single.on_epoch_end()
    assert single.copy().transductive_rate == 0.4
