"""Graph data layer vs golden fixtures produced by RUNNING the reference's numpy/scipy code
(tests/golden/make_golden.py): rows a13-a16, a18 of SURVEY.md §8. Bit-exact (same float32 values, same entry order)."""
import os
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


# ----------------------------------------------------------------------------------------------------------------------
# on-disk formats: files WRITTEN BY THE REFERENCE's own savers (tests/golden/make_golden.py -> tests/golden/ref_files/)
# are read back by this repository's loaders, and this repository's savers write the same keys / shapes / dtypes / values
# ----------------------------------------------------------------------------------------------------------------------
REF_FILES = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'ref_files')


def _same_npz(path_a, path_b):
    a, b = np.load(path_a), np.load(path_b)
    assert sorted(a.files) == sorted(b.files), (a.files, b.files)
    for k in a.files:
        assert a[k].shape == b[k].shape and a[k].dtype == b[k].dtype, (k, a[k].shape, b[k].shape, a[k].dtype, b[k].dtype)
        assert np.array_equal(a[k], b[k]), k


def test_reference_written_npz_files_load_and_resave_identically(golden, tmp_path):
    from gnnkeras_amd import GraphObject, CompositeGraphObject
    for name, key, focus in (('mutag0', 'mutag0', 'g'), ('mutag1_compressed', 'mutag1', 'g'), ('merge4', 'merge4', 'g')):
        g = GraphObject.load(os.path.join(REF_FILES, name), focus=focus, aggregation_mode='average')
        assert np.array_equal(g.arcs, golden[f'ref_files_{key}_arcs']) and np.array_equal(g.nodes, golden[f'ref_files_{key}_nodes'])
        assert np.array_equal(g.targets, golden[f'ref_files_{key}_targets'])
        (g.save_compressed if 'compressed' in name else g.save)(str(tmp_path / name))
        _same_npz(os.path.join(REF_FILES, name + '.npz'), str(tmp_path / (name + '.npz')))
    m4 = GraphObject.load(os.path.join(REF_FILES, 'merge4.npz'), focus='g', aggregation_mode='average')
    ng = golden['ref_files_merge4_NodeGraph']
    assert m4.NodeGraph.shape == tuple(golden['ref_files_merge4_shape'])
    dense = np.zeros(m4.NodeGraph.shape); dense[ng[:, 0].astype(int), ng[:, 1].astype(int)] = ng[:, 2]
    assert np.array_equal(m4.NodeGraph.toarray(), dense.astype(np.float32))
    # masks / sample weights: the toy graph of the fixtures (set_mask, output_mask, sample_weight = 2.5)
    toy = GraphObject.load(os.path.join(REF_FILES, 'toy_masks'), focus='n', aggregation_mode='average')
    assert np.array_equal(toy.set_mask, golden['toy_set_mask'].astype(bool)) and np.array_equal(toy.output_mask, golden['toy_output_mask'].astype(bool))
    assert np.array_equal(toy.arcs, golden['toy_average_arcs_out']) and np.all(toy.sample_weight == 2.5)
    assert np.array_equal(_coo_of(toy.ArcNode), golden['toy_average_ArcNode'])
    toy.save(str(tmp_path / 'toy_masks'))
    _same_npz(os.path.join(REF_FILES, 'toy_masks.npz'), str(tmp_path / 'toy_masks.npz'))
    # composite: + type_mask / dim_node_label keys
    cg = CompositeGraphObject.load(os.path.join(REF_FILES, 'ctoy0'), focus='n', aggregation_mode='composite_average')
    assert np.array_equal(cg.type_mask, golden['ctoy0_type_mask']) and list(cg.DIM_NODE_LABEL) == list(golden['ctoy_dim_node_label'])
    assert np.array_equal(_coo_of(cg.ArcNode), golden['ctoy0_composite_average_ArcNode'])
    for t, ca in enumerate(cg.CompositeAdjacencies):
        assert np.array_equal(_coo_of(ca), golden[f'ctoy0_composite_average_CA{t}'])
    cg.save(str(tmp_path / 'ctoy0'))
    _same_npz(os.path.join(REF_FILES, 'ctoy0.npz'), str(tmp_path / 'ctoy0.npz'))


def _coo_of(m):
    m = m.tocoo()
    o = np.lexsort((m.col, m.row))
    return np.stack([m.row[o].astype(np.float64), m.col[o].astype(np.float64), m.data[o].astype(np.float64)], axis=1)


def test_reference_written_txt_folders_and_datasets(golden, tmp_path):
    from gnnkeras_amd import GraphObject, CompositeGraphObject
    toy = GraphObject.load_txt(os.path.join(REF_FILES, 'toy_masks_txt'), focus='n', aggregation_mode='average')
    ref = GraphObject.load(os.path.join(REF_FILES, 'toy_masks'), focus='n', aggregation_mode='average')
    assert np.allclose(toy.arcs, ref.arcs, rtol=1e-9) and np.allclose(toy.nodes, ref.nodes, rtol=1e-9)      # '%.10g' text
    assert np.array_equal(toy.set_mask, ref.set_mask) and np.array_equal(toy.output_mask, ref.output_mask)
    assert np.array_equal(toy.sample_weight, ref.sample_weight)
    m4 = GraphObject.load_txt(os.path.join(REF_FILES, 'merge4_txt'), focus='g', aggregation_mode='average')
    assert m4.NodeGraph.shape == tuple(golden['ref_files_merge4_shape']) and np.array_equal(m4.arcs, golden['ref_files_merge4_arcs'])
    cg = CompositeGraphObject.load_txt(os.path.join(REF_FILES, 'ctoy0_txt'), focus='n', aggregation_mode='composite_average')
    assert np.array_equal(cg.type_mask, golden['ctoy0_type_mask']) and list(cg.DIM_NODE_LABEL) == list(golden['ctoy_dim_node_label'])
    # text written by this repository == text written by the reference, file by file
    for src, g in (('toy_masks_txt', toy), ('merge4_txt', m4), ('ctoy0_txt', cg)):
        g.savetxt(str(tmp_path / src))
        for f in sorted(os.listdir(os.path.join(REF_FILES, src))):
            assert open(os.path.join(REF_FILES, src, f)).read() == open(str(tmp_path / src / f)).read(), (src, f)
        assert sorted(os.listdir(os.path.join(REF_FILES, src))) == sorted(os.listdir(str(tmp_path / src)))
    # datasets: folder of npz files / folder of txt folders (reference save_dataset / save_dataset_txt)
    ds = GraphObject.load_dataset(os.path.join(REF_FILES, 'dataset_npz'), focus='g', aggregation_mode='average')
    assert len(ds) == 3 and np.array_equal(ds[0].arcs, golden['ref_files_ds2_arcs']) and np.array_equal(ds[0].nodes, golden['ref_files_ds2_nodes'])
    dt = GraphObject.load_dataset_txt(os.path.join(REF_FILES, 'dataset_txt'), focus='g', aggregation_mode='average')
    assert len(dt) == 2 and np.array_equal(dt[0].arcs, golden['ref_files_ds5_arcs']) and np.array_equal(dt[0].targets, golden['ref_files_ds5_targets'])
    GraphObject.save_dataset(str(tmp_path / 'ds'), ds)
    assert sorted(os.listdir(str(tmp_path / 'ds'))) == ['g0.npz', 'g1.npz', 'g2.npz']
    for f in ('g0.npz', 'g1.npz', 'g2.npz'):
        _same_npz(os.path.join(REF_FILES, 'dataset_npz', f), str(tmp_path / 'ds' / f))
    GraphObject.save_dataset_txt(str(tmp_path / 'dst'), dt)
    for gdir in ('g0', 'g1'):
        for f in sorted(os.listdir(os.path.join(REF_FILES, 'dataset_txt', gdir))):
            assert open(os.path.join(REF_FILES, 'dataset_txt', gdir, f)).read() == open(str(tmp_path / 'dst' / gdir / f)).read()
    many = [ds[0]] * 12
    GraphObject.save_dataset(str(tmp_path / 'many'), many)
    assert len(GraphObject.load_dataset(str(tmp_path / 'many'), 'g', 'average')) == 12        # g10, g11 sort after g9
