"""Host-side mirror of the reference interface: dims arithmetic, builders, error behaviour, sequencer tuple layout,
on-disk formats. No GPU."""
import numpy as np
import pytest
import torch

from gnnkeras_amd import GraphObject, GraphTensor, CompositeGraphObject, SparseMatrix
from gnnkeras_amd.Models.MLP import MLP, Sequential, get_inout_dims, initialize
from gnnkeras_amd.Models.GNN import GNNnodeBased, GNNarcBased, GNNgraphBased
from gnnkeras_amd.Models.CompositeGNN import CompositeGNNnodeBased
from gnnkeras_amd.Sequencers.GraphSequencers import (MultiGraphSequencer, SingleGraphSequencer,
                                                     CompositeMultiGraphSequencer)


def test_get_inout_dims_mutag():
    # MUTAG d=32 => state in 14+14+3+64 = 95 -> 32; output in 46 -> 2; d=0 => 31 -> 14 and 14 -> 2
    assert get_inout_dims('state', 14, 3, 2, 'g', 32) == ([(95,)], [32])
    assert get_inout_dims('output', 14, 3, 2, 'g', 32) == ([(46,)], [2])
    assert get_inout_dims('state', 14, 3, 2, 'g', 0) == ([(31,)], [14])
    assert get_inout_dims('output', 14, 3, 2, 'g', 0) == ([(14,)], [2])
    assert get_inout_dims('output', 14, 3, 2, 'a', 32) == ([(2 * 46 + 3,)], [2])
    assert get_inout_dims('state', 14, 3, 2, 'n', 64, hidden_units=[100, 50]) == ([(159,)], [100, 50, 64])
    # composite: one input shape per node type: d_t + sum(d) + A + 2D   (MLP.py:117-121)
    assert get_inout_dims('state', (4, 2, 3), 2, 2, 'n', 5)[0] == [(4 + 9 + 2 + 10,), (2 + 9 + 2 + 10,), (3 + 9 + 2 + 10,)]
    assert get_inout_dims('output', (4, 2, 3), 2, 2, 'n', 5) == ([(5,)], [2])
    # LGNN layer > 0 rule with state and output fed forward (MLP.py:107-113)
    assert get_inout_dims('state', 14, 3, 2, 'g', 0, layer=1, get_state=True, get_output=True)[0] == [(2 * 30 + 3,)]
    with pytest.raises(ValueError):
        get_inout_dims('hidden', 14, 3, 2, 'g', 0)
    with pytest.raises(AssertionError):
        get_inout_dims('state', 14, 3, 2, 'x', 0)


def test_mlp_builder_weight_order_and_errors():
    m = MLP((95,), [64, 32], ['selu', 'linear'], 'lecun_normal', 'zeros', rng=0, device='cpu')
    w = m.get_weights()
    assert [a.shape for a in w] == [(95,)] * 4 + [(95, 64), (64,), (64, 32), (32,)]
    assert np.all(w[0] == 1) and np.all(w[1] == 0) and np.all(w[2] == 0) and np.all(w[3] == 1)   # BN at init (Q10)
    assert abs(w[4].std() - np.sqrt(1 / 95)) < 0.01 and np.abs(w[4]).max() <= 2 * np.sqrt(1 / 95) / 0.8796 + 1e-6
    assert np.all(w[5] == 0)
    assert len(m.trainable_variables) == 2 + 4
    m2 = MLP((10,), [3], 'softmax', 'glorot_normal', 'glorot_normal', batch_normalization=False, device='cpu')
    assert [a.shape for a in m2.get_weights()] == [(10, 3), (3,)]
    with pytest.raises(ValueError):
        MLP((10,), [3, 2], ['relu'], 'zeros', 'zeros', device='cpu')
    with pytest.raises(ValueError):
        MLP((10,), [3], 'relu', 'zeros', 'zeros', dropout_rate=[0.1, 0.2], dropout_pos=[0], device='cpu')
    with pytest.raises(ValueError):
        MLP((10,), [3], 'swishh', 'zeros', 'zeros', device='cpu')
    with pytest.raises(ValueError):
        m2.set_weights([np.zeros((10, 4)), np.zeros(3)])
    c = m.clone()
    assert all(np.array_equal(a, b) for a, b in zip(c.get_weights(), m.get_weights()))
    same = MLP((95,), [64, 32], ['selu', 'linear'], 'lecun_normal', 'zeros', rng=0, device='cpu')
    assert all(np.array_equal(a, b) for a, b in zip(same.get_weights(), m.get_weights()))       # seeded


def test_initializer_statistics():
    rng = np.random.default_rng(0)
    g = initialize('glorot_normal', (200, 100), rng)
    assert abs(g.std() - np.sqrt(2 / 300)) < 2e-3
    u = initialize('glorot_uniform', (200, 100), rng)
    assert np.abs(u).max() <= np.sqrt(6 / 300)
    b = initialize('lecun_normal', (50,), rng)         # 1-D: fan_in = fan_out = 50
    assert b.shape == (50,) and np.abs(b).max() <= 2 * np.sqrt(1 / 50) / 0.8796 + 1e-6


def test_model_ctor_asserts_and_config():
    ns = MLP((31,), [14], 'selu', 'zeros', 'zeros', device='cpu')
    no = MLP((14,), [2], 'softmax', 'zeros', 'zeros', device='cpu')
    for bad in [dict(state_vect_dim=-1, max_iteration=1, state_threshold=0.1),
                dict(state_vect_dim=0, max_iteration=-1, state_threshold=0.1),
                dict(state_vect_dim=0, max_iteration=1, state_threshold=-0.1)]:
        with pytest.raises(AssertionError):
            GNNnodeBased(ns, no, **bad)
    g = GNNgraphBased(ns, no, 0, 0, 0.01)               # homogeneous allows max_iteration == 0 (Q13)
    assert g.get_config()['max_iteration'] == 0 and g.name == 'graph'
    with pytest.raises(AssertionError):
        CompositeGNNnodeBased([ns], no, 0, 0, 0.01)     # composite requires max_iteration > 0 (CompositeGNN.py:27)
    c = g.copy()
    assert c is not g and c.net_state is not g.net_state and c.state_threshold == 0.01
    g.compile(optimizer='adam', loss='categorical_crossentropy', metrics=['accuracy'], average_st_grads=True,
              run_eagerly=True)
    assert g.average_st_grads is True and 'avg=True' in repr(g)
    assert GNNarcBased.name == 'arc'


def test_model_requires_device_tensors_no_cpu_fallback():
    from gnnkeras_amd._native import NativeError
    ns = MLP((31,), [14], 'selu', 'zeros', 'zeros', device='cpu')
    no = MLP((14,), [2], 'softmax', 'zeros', 'zeros', device='cpu')
    g = GraphObject(nodes=np.eye(14)[:5], arcs=np.array([[0, 1, 1, 0, 0], [1, 2, 0, 1, 0]]), targets=np.ones((1, 2)), focus='g')
    seq = MultiGraphSequencer([g], 'g', 'sum', 1, shuffle=False, device='cpu')
    with pytest.raises(NativeError):
        GNNgraphBased(ns, no, 0, 3, 0.01)(seq[0][0])


def test_graphobject_errors():
    nodes, arcs, t = np.ones((3, 2)), np.array([[0, 1, 1.], [1, 2, 1.]]), np.ones((3, 1))
    with pytest.raises(ValueError):
        GraphObject(nodes, arcs, t, set_mask=np.ones(3), output_mask=np.ones(2))
    with pytest.raises(ValueError):
        GraphObject(nodes, arcs, t, aggregation_mode='mean')
    with pytest.raises(ValueError):
        CompositeGraphObject(nodes, arcs, t, type_mask=np.ones((3, 1)), dim_node_label=(2,), aggregation_mode='bogus')
    g = GraphObject(nodes, arcs, t, focus='a')
    assert len(g.set_mask) == 2 and g.NodeGraph.shape == (1, 0)
    assert g.DIM_ARC_LABEL == 1 and g.DIM_TARGET == 1 and list(g.DIM_NODE_LABEL) == [2]


def test_sequencer_tuple_layout(mutag_graphs):
    gs = mutag_graphs[:70]
    seq = MultiGraphSequencer(gs, 'g', 'average', 32, shuffle=False, device='cpu')
    assert len(seq) == 3
    x, y, sw = seq[0]
    assert len(x) == 8
    nodes, arcs, dnl, sm, om, adj, an, ng = x
    N, E = nodes.shape[0], arcs.shape[0]
    assert arcs.shape[1] == 5 and tuple(dnl.shape) == (1, 1) and dnl.dtype == torch.int32 and int(dnl) == 14
    assert tuple(sm.shape) == (N, 1) and sm.dtype == torch.bool and tuple(om.shape) == (N, 1)
    for trip, shape in [(adj, (N, N)), (an, (E, N)), (ng, (N, 32))]:
        idx, val, shp = trip
        assert idx.dtype == torch.int64 and idx.shape[1] == 2 and tuple(val.shape) == (idx.shape[0], 1)
        assert shp.dtype == torch.int64 and tuple(shp.tolist()) == shape
    assert tuple(y.shape) == (32, 2) and tuple(sw.shape) == (32,)
    assert seq[2][1].shape[0] == 70 - 64
    # process_inputs squeezes [2:5] and turns [5:] into sparse matrices (GNN.py:181-193)
    p = GNNgraphBased.process_inputs(x)
    assert tuple(p[3].shape) == (N,) and isinstance(p[5], SparseMatrix) and p[5] is adj.matrix
    # node-focused: targets rows = set_mask[output_mask]
    rng = np.random.default_rng(0)
    def node_graph(g):
        om_ = rng.random(g.nodes.shape[0]) < .6
        return GraphObject(g.nodes, g.arcs, rng.normal(size=(int(om_.sum()), 2)), focus='n',
                           set_mask=rng.random(g.nodes.shape[0]) < .7, output_mask=om_)
    ngs = [node_graph(g) for g in gs[:8]]
    sq = MultiGraphSequencer(ngs, 'n', 'sum', 8, shuffle=False, device='cpu')
    x, y, sw = sq[0]
    sm, om_ = x[3].reshape(-1), x[4].reshape(-1)
    assert y.shape[0] == int((sm & om_).sum()) == sw.shape[0]
    sq.set_batch_size(3)
    assert len(sq) == 3
    assert len(sq.copy()) == 3 and 'node-focused' in repr(sq)


def test_sequencer_shuffle_rebuilds():
    rng = np.random.default_rng(1)
    gs = [GraphObject(rng.normal(size=(n, 2)), np.array([[i, (i + 1) % n, 1.] for i in range(n)]), np.ones((1, 1)) * n,
                      focus='g') for n in range(3, 13)]
    seq = MultiGraphSequencer(gs, 'g', 'sum', 4, shuffle=True, device='cpu')
    before = [float(t) for i in range(len(seq)) for t in seq[i][1].reshape(-1)]
    np.random.seed(0)
    seq.on_epoch_end()
    after = [float(t) for i in range(len(seq)) for t in seq[i][1].reshape(-1)]
    assert sorted(before) == sorted(after) and before != after


def test_single_graph_sequencer_q7():
    rng = np.random.default_rng(2)
    n = 20
    g = GraphObject(rng.normal(size=(n, 2)), np.array([[i, (i + 1) % n, 1.] for i in range(n)]), rng.normal(size=(n, 1)),
                    focus='n', set_mask=np.arange(n) < 10)
    seq = SingleGraphSequencer(g, 'n', batch_size=4, shuffle=False, device='cpu')
    assert len(seq) == 3
    x, y, sw = seq[0]
    assert int(x[3].sum()) == 10 and y.shape[0] == 4      # x carries the full set_mask, targets the batch


def test_composite_sequencer_layout():
    rng = np.random.default_rng(3)
    def cg(n):
        tm = np.zeros((n, 2), bool); tm[np.arange(n), rng.integers(0, 2, n)] = True
        return CompositeGraphObject(rng.normal(size=(n, 3)), np.array([[i, (i + 1) % n, 1.] for i in range(n)]),
                                    rng.normal(size=(n, 1)), type_mask=tm, dim_node_label=(3, 2), focus='n')
    seq = CompositeMultiGraphSequencer([cg(5), cg(7)], 'n', 'composite_average', 2, shuffle=False, device='cpu')
    x, y, sw = seq[0]
    assert len(x) == 10 and tuple(x[2].shape) == (2, 1) and tuple(x[3].shape) == (2, 12, 1)
    assert isinstance(x[6], list) and len(x[6]) == 2 and tuple(x[7][2].tolist()) == (12, 12)
    p = CompositeGNNnodeBased.process_inputs(x)
    assert tuple(p[3].shape) == (2, 12) and isinstance(p[6][0], SparseMatrix)


def test_npz_round_trip(tmp_path, mutag_graphs):
    m = GraphObject.merge(mutag_graphs[:4], focus='g', aggregation_mode='average')
    m.save(str(tmp_path / 'g.npz'))
    back = GraphObject.load(str(tmp_path / 'g'), focus='g', aggregation_mode='average')
    assert np.array_equal(back.arcs, m.arcs) and np.array_equal(back.nodes, m.nodes)
    assert np.array_equal(back.NodeGraph.toarray(), m.NodeGraph.toarray())
    assert np.array_equal(back.Adjacency.toarray(), m.Adjacency.toarray())
    gt = GraphTensor.fromGraphObject(m, device='cpu')
    gt.save(str(tmp_path / 't.npz'))
    data = np.load(str(tmp_path / 't.npz'))
    assert data['Adjacency'].shape == (m.arcs.shape[0], 3)          # [value, row, col] triples (graph_class.py:513)
    assert tuple(data['Adjacency_shape']) == m.Adjacency.shape
    back = GraphTensor.load(str(tmp_path / 't'), device='cpu')
    assert np.array_equal(back.Adjacency.indices, gt.Adjacency.indices)
    assert np.array_equal(back.ArcNode.values, gt.ArcNode.values)
    again = GraphObject.fromGraphTensor(back, 'g')
    assert np.array_equal(again.arcs, m.arcs) and np.array_equal(again.NodeGraph.toarray(), m.NodeGraph.toarray())
    m.savetxt(str(tmp_path / 'txt'))
    t = GraphObject.load_txt(str(tmp_path / 'txt'), focus='g', aggregation_mode='average')
    assert np.allclose(t.arcs, m.arcs) and np.allclose(t.NodeGraph.toarray(), m.NodeGraph.toarray())


def test_model_save_load(tmp_path):
    ns = MLP((31,), [20, 14], ['selu', 'tanh'], 'lecun_normal', 'lecun_normal', rng=3, device='cpu')
    no = MLP((14,), [2], 'softmax', 'glorot_normal', 'zeros', rng=4, device='cpu')
    g = GNNgraphBased(ns, no, 0, 5, 0.01)
    g.save(str(tmp_path / 'model'))
    import json
    assert json.load(open(tmp_path / 'model' / 'config.json')) == {'state_vect_dim': 0, 'max_iteration': 5, 'state_threshold': 0.01}
    b = GNNgraphBased.load(str(tmp_path / 'model'))
    assert b.max_iteration == 5 and b.net_state.units == [20, 14] and b.net_state.activations == ['selu', 'tanh']
    assert all(np.array_equal(x, y) for x, y in zip(b.net_state.get_weights(), ns.get_weights()))


def test_split_heavy_rows_reconstructs_the_aggregate():
    """Hub rows are cut into segments that become virtual source rows of a 'light' operator (sparse.split_heavy)."""
    from gnnkeras_amd.sparse import CSRByDestination, split_heavy
    rng = np.random.default_rng(0)
    n = 60
    pairs = np.unique(np.concatenate([np.stack([rng.integers(0, n, 300), rng.integers(0, n, 300)], 1),
                                      np.stack([np.arange(n), np.full(n, 7)], 1),
                                      np.stack([rng.integers(0, n, 40), np.full(40, 9)], 1)]), axis=0)
    c = CSRByDestination.from_coo(pairs[:, 0], pairs[:, 1], rng.normal(size=len(pairs)).astype(np.float32), (n, n), uniform_rows=False)
    same, none = split_heavy(c, threshold=10 ** 6)
    assert same is c and none is None
    light, heavy = split_heavy(c, threshold=20, segment=16)
    deg = np.diff(c.rowptr)
    assert heavy['n_seg'] == int(np.sum(-(-deg[deg > 20] // 16))) and light.n_src == n + heavy['n_seg'] and light.n_dst == n
    assert np.all(heavy['seg_end'] - heavy['seg_beg'] <= 16) and np.all(heavy['seg_end'] > heavy['seg_beg'])
    X = rng.normal(size=(n, 3))
    agg = lambda cc, XX: np.stack([sum((cc.w[e] * XX[cc.src[e]] for e in range(cc.rowptr[j], cc.rowptr[j + 1])), np.zeros(3))
                                   for j in range(cc.n_dst)])
    V = np.stack([sum((c.w[e] * X[c.src[e]] for e in range(b, en)), np.zeros(3)) for b, en in zip(heavy['seg_beg'], heavy['seg_end'])])
    assert np.allclose(agg(light, np.concatenate([X, V])), agg(c, X))
    # row_scale graphs ('average'): weights stay implicit, virtual arcs need none
    cu = CSRByDestination.from_coo(pairs[:, 0], pairs[:, 1], (1.0 / deg[pairs[:, 1]]).astype(np.float32), (n, n))
    lu, hu = split_heavy(cu, threshold=20, segment=16)
    assert cu.w is None and lu.w is None and lu.row_scale is cu.row_scale


def test_merged_batches_for_convergence_groups(mutag_graphs):
    """`MultiGraphSequencer.merged_batches`: several batches as ONE graph (same arrays as merging their graphs in order) plus
    the node offsets a model passes as `groups=`; a range and the list of its members are the same merge; 'normalized'
    aggregation is never merged across batches (its weights depend on the merged graph's arc count)."""
    gl = [g.copy() for g in mutag_graphs[:40]]
    seq = MultiGraphSequencer(gl, 'g', 'average', 8, shuffle=False, device='cpu')
    x, begin = seq.merged_batches([1, 3, 4])
    graphs = gl[8:16] + gl[24:40]
    want = GraphObject.merge(graphs, focus='g', aggregation_mode='average')
    assert begin == [0] + list(np.cumsum([sum(g.nodes.shape[0] for g in gl[8:16]), sum(g.nodes.shape[0] for g in gl[24:32]),
                                          sum(g.nodes.shape[0] for g in gl[32:40])]))
    assert np.array_equal(x[0].numpy(), want.nodes.astype(np.float32)) and np.array_equal(x[1].numpy(), want.arcs.astype(np.float32))
    assert x[7][2][0] == sum(g.nodes.shape[0] for g in graphs) and x[7][2][1] == len(graphs)         # NodeGraph: nodes x graphs
    xr, br = seq.merged_batches(3, 5)
    xl, bl = seq.merged_batches([3, 4])
    assert br == bl and xr is xl                                      # one cache entry
    assert MultiGraphSequencer(gl, 'g', 'normalized', 8, shuffle=False, device='cpu').merged_batches(0, 2) is None
    seq.set_batch_size(10)                                            # rebuilt batches: the cache starts over
    x2, b2 = seq.merged_batches(0, 2)
    assert b2[-1] == sum(g.nodes.shape[0] for g in gl[:20])


def test_single_sequencers_opt_out_of_merged_batches_and_keep_their_config():
    """A single-graph sequencer has nothing to merge (every batch is the whole graph): `merged_batches` answers None instead of
    touching attributes it never had, so predict() / evaluate() take the batch-by-batch path; `assemble` survives copy()."""
    from gnnkeras_amd.Sequencers.GraphSequencers import CompositeSingleGraphSequencer
    from gnnkeras_amd.Sequencers.TransductiveGraphSequencers import TransductiveMultiGraphSequencer, TransductiveSingleGraphSequencer
    rng = np.random.default_rng(2)
    n = 20
    g = GraphObject(rng.normal(size=(n, 2)), np.array([[i, (i + 1) % n, 1.] for i in range(n)]), rng.normal(size=(n, 1)),
                    focus='n', set_mask=np.arange(n) < 10)
    seq = SingleGraphSequencer(g, 'n', batch_size=4, shuffle=False, device='cpu')
    assert seq.merged_batches(0, 1) is None and seq.merged_batches([0, 1]) is None
    tm = np.zeros((n, 2), bool); tm[np.arange(n), rng.integers(0, 2, n)] = True
    cg = CompositeGraphObject(rng.normal(size=(n, 3)), np.array([[i, (i + 1) % n, 1.] for i in range(n)]), rng.normal(size=(n, 1)),
                              type_mask=tm, dim_node_label=(3, 2), focus='n')
    assert CompositeSingleGraphSequencer(cg, 'n', batch_size=4, shuffle=False, device='cpu').merged_batches(0, 1) is None
    np.random.seed(0)
    assert TransductiveSingleGraphSequencer(g, 'n', 0.5, batch_size=4, shuffle=False, device='cpu').merged_batches(0, 1) is None
    # composite batches are not merged across batches either (no convergence groups for composite models)
    tseq = TransductiveMultiGraphSequencer([g.copy(), g.copy()], 'n', 'average', 0.5, batch_size=1, shuffle=False, device='cpu')
    assert tseq.merged_batches(0, 2) is None
    assert tseq.copy().transductive_rate == 0.5                            # get_config / from_config round trip
    ms = MultiGraphSequencer([g.copy(), g.copy()], 'n', 'average', 1, shuffle=False, device='cpu', assemble='host')
    assert ms.get_config()['assemble'] == 'host' and ms.copy().assemble == 'host'
    # the same GraphObject twice in the list: the signature of the device data set counts distinct objects
    twice = MultiGraphSequencer([g, g], 'n', 'average', 1, shuffle=False, device='cpu')
    assert len(twice) == 2 and twice[0][0][0].shape[0] == n


def test_fit_evaluate_predict_reject_unknown_keyword_arguments():
    """Keras would raise on an argument it does not know; silently dropping one (round 2) hid typos such as `callback=`."""
    ns = MLP((9,), [2], 'linear', 'zeros', 'zeros', device='cpu')
    no = MLP((4,), [1], 'linear', 'zeros', 'zeros', device='cpu')
    m = GNNnodeBased(ns, no, 2, 3, 0.01)
    m.compile(optimizer='adam', loss='mse')
    for call in (lambda: m.fit([], epochs=1, callback=[]), lambda: m.evaluate([], bogus=1), lambda: m.predict([], bogus=1)):
        with pytest.raises(TypeError):
            call()
    # Keras arguments that would change the loss / the amount of training are refused unless they sit at their Keras default
    for call in (lambda: m.fit([], epochs=1, class_weight={0: 2.0}), lambda: m.fit([], epochs=1, steps_per_epoch=3),
                 lambda: m.fit([], epochs=1, validation_freq=2), lambda: m.evaluate([], sample_weight=np.ones(3)), lambda: m.predict([], steps=2)):
        with pytest.raises(NotImplementedError):
            call()
    from gnnkeras_amd.Models.GNN import History
    h = m.fit([], epochs=2, verbose=0, workers=1, callbacks=[], class_weight=None, validation_freq=1, shuffle=True)     # no batches: nothing touches the device
    assert isinstance(h, History) and h.history is h and h.epoch == [0, 1]
    seen = []
    class CB:
        def set_model(self, model): seen.append(('model', model is m))
        def on_train_begin(self, logs): seen.append('begin')
        def on_epoch_end(self, epoch, logs):
            seen.append(('epoch', epoch))
            if epoch == 1: m.stop_training = True
        def on_train_end(self, logs): seen.append('end')
    m.fit([], epochs=5, verbose=0, callbacks=[CB()])
    assert seen == [('model', True), 'begin', ('epoch', 0), ('epoch', 1), 'end']


def test_mutag_composite_graphs_are_the_same_graphs_with_one_node_type(mutag_graphs):
    """`load_MUTAG.composite_graphs` (reference load_MUTAG.py:57-60, what starter_composite.py trains on): one node type, the label width
    as the single `dim_node_label` entry, the composite adjacency of that type IS the adjacency."""
    from gnnkeras_amd.load_MUTAG import load_composite_graphs
    from gnnkeras_amd.Sequencers.GraphSequencers import CompositeMultiGraphSequencer
    cgs = load_composite_graphs(limit=7)
    assert len(cgs) == 7
    for c, g in zip(cgs, mutag_graphs):
        assert c.type_mask.shape == (g.nodes.shape[0], 1) and c.type_mask.all() and list(c.DIM_NODE_LABEL) == [g.nodes.shape[1]]
        assert np.array_equal(c.nodes, g.nodes) and np.array_equal(c.arcs, g.arcs) and np.array_equal(c.targets, g.targets)
        assert len(c.CompositeAdjacencies) == 1
        assert np.array_equal(c.CompositeAdjacencies[0].toarray(), c.Adjacency.toarray())
    x, y, sw = CompositeMultiGraphSequencer(cgs, 'g', 'average', 7, shuffle=False)[0]
    assert len(x) == 10 and y.shape == (7, 2)


def test_bench_reports_pmc_traffic_only_for_the_library_sources_it_runs_on(tmp_path):
    """a PMC record of profiles/hbm_traffic.json is this run's traffic only when it was taken on THESE sources of
    csrc/ (`library_source_hash`): otherwise `traffic` is null and `traffic_null_reason` says why."""
    import importlib.util, json, os
    from gnnkeras_amd._native import source_hash
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location('bench_for_test', os.path.join(root, 'bench.py'))
    bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
    here = source_hash()
    assert len(here) == 16 and here == source_hash()
    f = tmp_path / 'traffic.json'
    base = {'kernel': 'k_x<64>', 'nodes': 10, 'arcs': 20, 'state_dim': 64, 'hbm_bytes_per_launch': 123.0, 'bounds': [100.0, 123.0], 'launches': 7}
    f.write_text(json.dumps({'records': [dict(base, library_source_hash='0' * 16)]}))
    rec = bench.traffic_lookup({'traffic': None}, 'k_x<64>', 10, 20, 64, traffic_file=str(f))
    assert rec['traffic'] is None and 'other library sources' in rec['traffic_null_reason'] and rec['library_source_hash'] == here
    f.write_text(json.dumps({'records': [dict(base, library_source_hash='0' * 16), dict(base, library_source_hash=here)]}))
    rec = bench.traffic_lookup({'traffic': None}, 'k_x<64>', 10, 20, 64, traffic_file=str(f))
    assert rec['traffic'] == 123.0 and 'traffic_null_reason' not in rec and here in rec['traffic_source']
    rec = bench.traffic_lookup({'traffic': None}, 'k_y<64>', 10, 20, 64, traffic_file=str(f))
    assert rec['traffic'] is None and rec['traffic_null_reason'] == 'no PMC record for this kernel and workload'
    # rocprofv3 names carry the defaulted trailing template arguments the library's own name leaves out
    f.write_text(json.dumps({'records': [dict(base, kernel='k_x<64, false, 1, false>', library_source_hash=here)]}))
    assert bench.traffic_lookup({'traffic': None}, 'k_x<64,false>', 10, 20, 64, traffic_file=str(f))['traffic'] == 123.0
    assert bench.traffic_lookup({'traffic': None}, 'k_x<64,true>', 10, 20, 64, traffic_file=str(f))['traffic'] is None
    assert bench.traffic_lookup({'traffic': None}, 'k_x<6>', 10, 20, 64, traffic_file=str(f))['traffic'] is None
    rec = bench.traffic_lookup({'traffic': None}, 'k_x<64>', 10, 20, 64, traffic_file=str(tmp_path / 'absent.json'))
    assert rec['traffic'] is None and 'missing' in rec['traffic_null_reason']


def test_graphobject_copy_field_by_field_equals_the_constructor_path(mutag_graphs):
    """`GraphObject.copy()` (reference graph_class.py:141-146 copies through the constructor) takes a field-by-field path when the constructor
    would change nothing; both must give the same object - every array, every operator, the DIM_* fields - and a graph whose ArcNode or arc
    order was changed by hand must fall back to the constructor, which rebuilds them."""
    from gnnkeras_amd.graph_class import GraphObject

    def same(a, b):
        assert set(a.__dict__) == set(b.__dict__)
        for key, x in a.__dict__.items():
            y = b.__dict__[key]
            if hasattr(x, 'tocsr'):
                assert x.shape == y.shape and x.dtype == y.dtype and x.data is not y.data, key
                assert np.array_equal(x.row, y.row) and np.array_equal(x.col, y.col) and np.array_equal(x.data, y.data), key
            elif isinstance(x, np.ndarray):
                assert x is not y and x.dtype == y.dtype and x.shape == y.shape and np.array_equal(x, y), key
            else:
                assert type(x) == type(y) and x == y, key

    for g in mutag_graphs[:60]:
        g = g.copy()
        for mode in ('average', 'sum', 'normalized'):
            g.setAggregation(mode)
            fast = g._copy_fields()
            assert fast is not None
            slow = GraphObject(nodes=g.getNodes(), arcs=g.getArcs(), targets=g.getTargets(), set_mask=g.getSetMask(), output_mask=g.getOutputMask(),
                               sample_weight=g.getSampleWeights(), NodeGraph=g.getNodeGraph(), aggregation_mode=g.aggregation_mode)
            same(fast, slow)
    g = mutag_graphs[3].copy()
    g.ArcNode.data[0] = 0.123                                    # an operator changed by hand: the constructor rebuilds it from the mode
    assert g._copy_fields() is None and g.copy().ArcNode.data[0] != np.float32(0.123)
    g = mutag_graphs[3].copy()
    g.arcs = g.arcs[::-1].copy()                                 # arcs out of order: the constructor sorts them again
    assert g._copy_fields() is None and np.array_equal(g.copy().arcs, mutag_graphs[3].arcs)
    g = mutag_graphs[3].copy()
    g.nodes = np.concatenate([g.nodes, g.nodes], axis=1)         # what LGNN's propagation does: wider labels -> DIM_NODE_LABEL follows the array
    assert g.copy().DIM_NODE_LABEL[0] == g.nodes.shape[1]
