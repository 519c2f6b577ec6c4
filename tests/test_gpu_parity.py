"""Parity tests proper: the HIP path (through the C ABI, libgnnloop.so) against the oracle on identical inputs.

Tolerance (BASELINE.json north_star): float32 node state and output within 1e-5 relative —
`max|a-b| / max|b|` — against the float32 restatement, with both also compared to the float64 restatement.
k (iteration count) must match exactly when pinned by threshold 0."""
import ctypes as C

import numpy as np
import pytest
import torch

from gnnkeras_amd import _native as nat
from gnnkeras_amd import GraphObject, CompositeGraphObject, SparseMatrix
from gnnkeras_amd.Models.MLP import MLP, Sequential, get_inout_dims
from gnnkeras_amd.Models.GNN import GNNnodeBased, GNNarcBased, GNNgraphBased
from gnnkeras_amd.Models.CompositeGNN import CompositeGNNnodeBased, CompositeGNNarcBased, CompositeGNNgraphBased
from gnnkeras_amd.Sequencers.GraphSequencers import MultiGraphSequencer, CompositeMultiGraphSequencer
from gnnkeras_amd.synth import er_graph, er_composite_graph
from oracle import gnn_oracle as O
from oracle.harness import oracle_loop, oracle_composite_loop, rel_err

pytestmark = pytest.mark.gpu
# every way the iteration can run: size-based default, un-fused kernels, and each fused-kernel generation pinned
# (2 phase-alternating, 4 wave-specialised, 5 / 6 whole loop in one launch: one tile per CU / several tiles per workgroup)
PATHS = (0, nat.FLAG_UNFUSED, nat.FLAG_FUSED_GEN2, nat.FLAG_FUSED_GEN4, nat.FLAG_FUSED_GEN5, nat.FLAG_FUSED_GEN6)
TOL = 1e-5
CLS = {'n': GNNnodeBased, 'a': GNNarcBased, 'g': GNNgraphBased}
CCLS = {'n': CompositeGNNnodeBased, 'a': CompositeGNNarcBased, 'g': CompositeGNNgraphBased}


def dev(x):
    return torch.as_tensor(np.asarray(x)).cuda()


def starter_nets(focus, d, L=14, A=3, T=2, hidden_state=None, hidden_out=None, act='selu', scale=1.0, bn=True):
    inp, lay = get_inout_dims('state', L, A, T, focus, d, hidden_units=hidden_state)
    acts = [act] * len(lay)
    ns = MLP(inp[0], lay, acts, 'lecun_normal', 'lecun_normal', rng=0, batch_normalization=bn)
    inp, lay = get_inout_dims('output', L, A, T, focus, d, hidden_units=hidden_out)
    no = MLP(inp[0], lay, ['tanh'] * (len(lay) - 1) + ['softmax'], 'glorot_normal', 'glorot_normal', rng=1,
             batch_normalization=bn)
    if scale != 1.0:
        w = ns.get_weights()
        ns.set_weights([a * scale if a.ndim == 2 else a for a in w])
    return ns, no


def refocus(graphs, focus, rng):
    if focus == 'g': return graphs
    out = []
    for g in graphs:
        n = (g.nodes if focus == 'n' else g.arcs).shape[0]
        om = rng.random(n) < 0.7
        out.append(GraphObject(nodes=g.nodes, arcs=g.arcs, targets=rng.normal(size=(int(om.sum()), 2)), focus=focus,
                               set_mask=rng.random(n) < 0.8, output_mask=om))
    return out


def check(model, x, s0, tol=TOL, pin_k=True, oracle=oracle_loop):
    k64, st64, o64 = oracle(model, x, s0, np.float64)
    k32, st32, o32 = oracle(model, x, s0, np.float32)
    res = {}
    for flags in PATHS:
        model.native_flags = flags
        k, st, o = model.Loop(*model.process_inputs(x), state0=None if s0 is None else dev(s0))
        torch.cuda.synchronize()
        k, st, o = float(k), st.cpu().numpy(), o.cpu().numpy()
        assert st.shape == st32.shape and o.shape == o32.shape
        if pin_k:
            assert k == float(k32) == float(k64), (k, k32, k64)
        assert np.all(np.isfinite(st)) and np.all(np.isfinite(o))
        e = dict(st32=rel_err(st, st32), st64=rel_err(st, st64), o32=rel_err(o, o32), o64=rel_err(o, o64),
                 ref=rel_err(st32, st64))
        assert e['st32'] <= tol and e['st64'] <= tol, (flags, e)
        assert e['o32'] <= tol and e['o64'] <= tol, (flags, e)
        res[flags] = (k, st, o)
    assert rel_err(res[0][1], res[nat.FLAG_UNFUSED][1]) <= tol
    return res[0]


# ----------------------------------------------------------------------------------------------------------------------
# C2: MUTAG batch = first 32 graphs in file order (N=935, E=1922), d=32, max_iteration=50
# ----------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('focus', ['g', 'n', 'a'])
def test_c2_mutag_batch_state_dim_32_k_pinned(mutag_graphs, focus):
    gl = refocus(mutag_graphs[:32], focus, np.random.default_rng(5))
    seq = MultiGraphSequencer(gl, focus, 'average', 32, shuffle=False)
    x, y, sw = seq[0]
    assert x[0].shape[0] == 935 and x[1].shape[0] == 1922
    ns, no = starter_nets(focus, 32)
    model = CLS[focus](ns, no, 32, 50, 0.0)
    s0 = np.random.default_rng(1).normal(0, 0.1, (935, 32)).astype(np.float32)
    k, st, o = check(model, x, s0)
    assert k == 50.0
    if focus == 'g': assert o.shape == (32, 2) and np.allclose(o.sum(1), 1, atol=1e-5)


def test_c2_realistic_threshold_converges_early(mutag_graphs):
    """threshold 0.01 with a contractive state network: k < max_iteration, equal to the oracle's (reported, H3)."""
    seq = MultiGraphSequencer(mutag_graphs[:32], 'g', 'average', 32, shuffle=False)
    x = seq[0][0]
    ns, no = starter_nets('g', 32, scale=0.25)
    model = GNNgraphBased(ns, no, 32, 50, 0.01)
    s0 = np.random.default_rng(1).normal(0, 0.1, (935, 32)).astype(np.float32)
    k, st, o = check(model, x, s0)
    assert 2 <= k < 50


def test_c1_starter_config_state_dim_0(mutag_graphs):
    """starter.py: dim_state=0, max_iter=5, threshold=0.01, 'average', graph focus; several batches incl. the ragged last one."""
    gs = [g.copy() for g in mutag_graphs[:100]]
    for g in gs: g.setAggregation('average')
    seq = MultiGraphSequencer(gs, 'g', 'average', 32, shuffle=False)
    ns, no = starter_nets('g', 0)
    model = GNNgraphBased(ns, no, 0, 5, 0.01)
    for i in range(len(seq)):
        check(model, seq[i][0], None)


@pytest.mark.parametrize('mode', ['sum', 'normalized'])
def test_other_aggregation_modes(mutag_graphs, mode):
    seq = MultiGraphSequencer(mutag_graphs[32:64], 'g', mode, 32, shuffle=False)
    ns, no = starter_nets('g', 16, scale=0.5)
    model = GNNgraphBased(ns, no, 16, 8, 0.0)
    N = seq[0][0][0].shape[0]
    check(model, seq[0][0], np.random.default_rng(2).normal(0, 0.1, (N, 16)).astype(np.float32))


def test_per_arc_weights_path(mutag_graphs):
    """A user-supplied ArcNode with non-uniform values forces the per-arc weight array (w != NULL) kernels."""
    rng = np.random.default_rng(3)
    m = GraphObject.merge(mutag_graphs[:16], 'g', 'sum')
    an = m.getArcNode(); an.data = rng.uniform(0.2, 1.0, len(an.data)).astype(np.float32)
    m = GraphObject(nodes=m.nodes, arcs=m.arcs, targets=m.targets, focus='g', ArcNode=an, NodeGraph=m.NodeGraph)
    seq = MultiGraphSequencer([m], 'g', 'sum', 1, shuffle=False)
    # the sequencer's merge re-derives ArcNode for its mode: put the custom operands back on the batch
    seq.graph_tensors[0].ArcNode = SparseMatrix.from_scipy(m.ArcNode)
    seq.graph_tensors[0].Adjacency = SparseMatrix.from_scipy(m.Adjacency)
    seq._items = [None]
    x = seq[0][0]
    assert x[5].matrix.csr().w is not None
    ns, no = starter_nets('g', 32, scale=0.3)
    model = GNNgraphBased(ns, no, 32, 10, 0.0)
    N = x[0].shape[0]
    check(model, x, rng.normal(0, 0.1, (N, 32)).astype(np.float32))


@pytest.mark.parametrize('act', ['tanh', 'relu', 'sigmoid', 'linear', 'elu', 'softplus'])
def test_multilayer_networks_and_activations(mutag_graphs, act):
    """Hidden layers in both networks take the un-fused general path (segmented MFMA dense chain)."""
    seq = MultiGraphSequencer(mutag_graphs[:32], 'g', 'average', 32, shuffle=False)
    x = seq[0][0]
    ns, no = starter_nets('g', 24, hidden_state=[40, 17], hidden_out=[9], act=act, scale=0.5)
    model = GNNgraphBased(ns, no, 24, 6, 0.0)
    check(model, x, np.random.default_rng(4).normal(0, 0.1, (935, 24)).astype(np.float32))


@pytest.mark.parametrize('d', [1, 3, 14, 20, 33, 64, 100, 130, 200])
def test_odd_state_widths(mutag_graphs, d):
    seq = MultiGraphSequencer(refocus(mutag_graphs[:16], 'n', np.random.default_rng(d)), 'n', 'average', 16, shuffle=False)
    x = seq[0][0]
    ns, no = starter_nets('n', d, scale=0.5, bn=(d % 2 == 0))
    model = GNNnodeBased(ns, no, d, 4, 0.0)
    N = x[0].shape[0]
    check(model, x, np.random.default_rng(d).normal(0, 0.1, (N, d)).astype(np.float32))


# ----------------------------------------------------------------------------------------------------------------------
# known answers and edge cases through the GPU
# ----------------------------------------------------------------------------------------------------------------------
def _toy(focus='n', mode='sum', n=7):
    rng = np.random.default_rng(7)
    nodes = rng.normal(size=(n, 3))
    arcs = np.array([[0, 1, .5, 1.], [1, 0, .25, 2.], [2, 1, 1., 3.], [3, 1, 2., 4.], [1, 2, 3., 5.], [4, 5, 1., 6.],
                     [5, 4, 2., 7.], [3, 2, 7., 8.]])                  # node 6 isolated
    nt = {'n': n, 'a': len(arcs), 'g': 1}[focus]
    return GraphObject(nodes=nodes, arcs=arcs, targets=rng.normal(size=(nt, 2)), focus=focus, aggregation_mode=mode)


def test_zero_weights_converge_at_k2_on_gpu():
    g = _toy()
    seq = MultiGraphSequencer([g], 'n', 'sum', 1, shuffle=False)
    d = 5
    b = np.array([.3, -.2, 1.5, 0., -3.], np.float32)
    ns = Sequential(2 * d + 2 * 3 + 2, [d], ['selu'], batch_normalization=False,
                    weights=[np.zeros((2 * d + 8, d)), b])
    no = Sequential(d + 3, [2], ['linear'], batch_normalization=False, weights=[np.ones((d + 3, 2)), np.zeros(2)])
    for thr in (0.0, 0.5):
        model = GNNnodeBased(ns, no, d, 10, thr)
        k, st, o = check(model, seq[0][0], np.full((7, d), 0.1, np.float32))
        assert k == 2.0
        assert np.allclose(st, np.tile(O.activation('selu', b, np.float32), (7, 1)), atol=1e-6)


def test_max_iteration_zero_and_threshold_huge():
    g = _toy()
    seq = MultiGraphSequencer([g], 'n', 'average', 1, shuffle=False)
    ns, no = starter_nets('n', 8, L=3, A=2)
    s0 = np.random.default_rng(0).normal(0, .1, (7, 8)).astype(np.float32)
    k, st, o = check(GNNnodeBased(ns, no, 8, 0, 0.0), seq[0][0], s0)
    assert k == 0.0 and np.array_equal(st, s0)
    # ||s0 - 1|| > thr * sqrt(d) is false for a huge threshold: the loop never starts (GNN.py:196-214 at k = 0)
    k, st, o = check(GNNnodeBased(ns, no, 8, 10, 100.0), seq[0][0], s0)
    assert k == 0.0 and np.array_equal(st, s0)


def test_empty_mask_no_arcs_single_node():
    ns, no = starter_nets('n', 4, L=3, A=2)
    model = GNNnodeBased(ns, no, 4, 3, 0.0)
    g = _toy()
    g0 = GraphObject(g.nodes, g.arcs, np.zeros((0, 2)), focus='n', set_mask=np.zeros(7, bool), output_mask=np.zeros(7, bool))
    x = MultiGraphSequencer([g0], 'n', 'sum', 1, shuffle=False)[0][0]
    k, st, o = check(model, x, np.zeros((7, 4), np.float32) + .1)
    assert o.shape == (0, 2)
    g1 = GraphObject(np.ones((1, 3)), np.zeros((0, 4)), np.ones((1, 2)), focus='n')
    x = MultiGraphSequencer([g1], 'n', 'average', 1, shuffle=False)[0][0]
    k, st, o = check(model, x, np.full((1, 4), .1, np.float32))
    assert o.shape == (1, 2)


def test_merged_equals_separate_on_gpu(mutag_graphs):
    """Block-diagonal batching invariance ('average'), with k pinned."""
    gs = mutag_graphs[:6]
    ns, no = starter_nets('g', 16)
    model = GNNgraphBased(ns, no, 16, 7, 0.0)
    rng = np.random.default_rng(8)
    s0s = [rng.normal(0, .1, (g.nodes.shape[0], 16)).astype(np.float32) for g in gs]
    big = MultiGraphSequencer(gs, 'g', 'average', 6, shuffle=False)
    k, st, o = model.Loop(*model.process_inputs(big[0][0]), state0=dev(np.concatenate(s0s)))
    off = 0
    small = MultiGraphSequencer(gs, 'g', 'average', 1, shuffle=False)
    for i, g in enumerate(gs):
        ki, sti, oi = model.Loop(*model.process_inputs(small[i][0]), state0=dev(s0s[i]))
        n = g.nodes.shape[0]
        assert float(ki) == float(k)
        assert rel_err(st[off:off + n].cpu().numpy(), sti.cpu().numpy()) <= TOL
        assert rel_err(o[i:i + 1].cpu().numpy(), oi.cpu().numpy()) <= TOL
        off += n


def test_golden_reference_operands_through_gpu(golden, mutag_graphs):
    """Feed the Adjacency / ArcNode / NodeGraph produced by the REFERENCE's own numpy code (golden fixture) straight
    into the device loop and into the oracle."""
    p = 'merge32_average_'
    N, E = golden[p + 'nodes'].shape[0], golden[p + 'arcs'].shape[0]
    trip = lambda key, shape: (golden[p + key][:, :2].astype(np.int64), golden[p + key][:, 2:3].astype(np.float32),
                               np.array(shape))
    x = [dev(golden[p + 'nodes']), dev(golden[p + 'arcs']), torch.tensor([[14]], dtype=torch.int32),
         dev(golden[p + 'set_mask'])[:, None], dev(golden[p + 'output_mask'])[:, None],
         trip('Adjacency', (N, N)), trip('ArcNode', (E, N)), trip('NodeGraph', tuple(golden[p + 'NodeGraph_shape']))]
    ns, no = starter_nets('g', 32)
    model = GNNgraphBased(ns, no, 32, 50, 0.0)
    s0 = np.random.default_rng(1).normal(0, 0.1, (N, 32)).astype(np.float32)
    k, st, o = check(model, x, s0)
    seq = MultiGraphSequencer(mutag_graphs[:32], 'g', 'average', 32, shuffle=False)
    model.native_flags = 0
    k2, st2, o2 = model.Loop(*model.process_inputs(seq[0][0]), state0=dev(s0))
    assert np.array_equal(st2.cpu().numpy(), st) and np.array_equal(o2.cpu().numpy(), o)


def test_bitwise_run_to_run_determinism(mutag_graphs):
    seq = MultiGraphSequencer(mutag_graphs[:32], 'g', 'average', 32, shuffle=False)
    ns, no = starter_nets('g', 32)
    model = GNNgraphBased(ns, no, 32, 20, 0.0)
    s0 = dev(np.random.default_rng(1).normal(0, 0.1, (935, 32)).astype(np.float32))
    inputs = model.process_inputs(seq[0][0])
    a = [t.clone() for t in model.Loop(*inputs, state0=s0)]
    for _ in range(3):
        b = model.Loop(*inputs, state0=s0)
        assert all(torch.equal(p, q) for p, q in zip(a, b))


# ----------------------------------------------------------------------------------------------------------------------
# separately callable pieces of the ABI
# ----------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('F', [1, 3, 14, 32, 70])
def test_gnn_aggregate_matches_adjoint_spmm(F):
    rng = np.random.default_rng(F)
    n_src, n_dst, nnz = 300, 200, 1500
    idx = np.unique(np.stack([rng.integers(0, n_src, nnz), rng.integers(0, n_dst, nnz)], 1), axis=0)
    val = rng.normal(size=len(idx)).astype(np.float32)
    X = rng.normal(size=(n_src, F)).astype(np.float32)
    want = O.sparse_dense_matmul_adjoint(idx, val, (n_src, n_dst), X, np.float32)
    m = SparseMatrix(idx, val, (n_src, n_dst))
    c = m.device_csr('cuda')
    Xd = dev(X)
    out = torch.empty((n_dst, F), dtype=torch.float32, device='cuda')
    csr = nat.make_csr(c)
    nat.check(nat.lib().gnn_aggregate(C.byref(csr), nat.ptr(Xd), F, F, nat.ptr(out), F, nat.current_stream(Xd.device)))
    assert rel_err(out.cpu().numpy(), want) <= TOL


def test_sequential_call_matches_oracle_mlp():
    rng = np.random.default_rng(0)
    net = MLP((37,), [50, 70, 5], ['selu', 'tanh', 'softmax'], 'glorot_normal', 'lecun_normal', rng=0)
    w = net.get_weights()
    w[0:4] = [rng.uniform(.5, 1.5, 37), rng.normal(size=37), rng.normal(size=37), rng.uniform(.5, 2, 37)]
    net.set_weights(w)
    x = rng.normal(size=(333, 37)).astype(np.float32)
    got = net(dev(x)).cpu().numpy()
    spec, ww = net.spec()
    assert rel_err(got, O.mlp_apply(spec, ww, x, False, np.float64)) <= TOL
    assert net(dev(x[:0])).shape == (0, 5)


def test_condition_truth_table_on_gpu():
    ns, no = starter_nets('n', 2, L=3, A=2)
    cases = [([[1., 1.]], [[1., 1.]], 5, 0.0, 0, False), ([[0., 0.]], [[0., 0.]], 5, 0.5, 0, False),
             ([[1., 0.]], [[0., 0.]], 5, 10., 0, True), ([[1.1, 1.]], [[1., 1.]], 5, 0.05, 0, True),
             ([[1.1, 1.]], [[1., 1.]], 5, 0.1, 0, False), ([[9., 9.]], [[1., 1.]], 5, 0.0, 5, False),
             ([[1., 1.], [5., 5.]], [[1., 1.], [1., 1.]], 5, 0.1, 0, True)]
    for s, so, mx, thr, k, want in cases:
        m = GNNnodeBased(ns, no, 2, mx, thr)
        got = bool(m.condition(k, dev(np.array(s, np.float32)), dev(np.array(so, np.float32))))
        assert got is want is O.condition(k, np.array(s, np.float32), np.array(so, np.float32), mx, thr, np.float32)
    m = GNNnodeBased(ns, no, 2, 5, 0.5)
    assert bool(m.condition(0, dev(np.full((3, 2), .1, np.float32)), None)) is True     # state_old = ones


def test_convergence_step_matches_oracle(mutag_graphs):
    seq = MultiGraphSequencer(mutag_graphs[:8], 'g', 'average', 8, shuffle=False)
    x = seq[0][0]
    ns, no = starter_nets('g', 32)
    model = GNNgraphBased(ns, no, 32, 5, 0.0)
    nodes, arcs, _, _, _, adj, an, ng = model.process_inputs(x)
    N = nodes.shape[0]
    s = np.random.default_rng(0).normal(0, .1, (N, 32)).astype(np.float32)
    a = (adj.indices, adj.values, np.array(adj.shape))
    agg_nodes = O.sparse_dense_matmul_adjoint(*a, nodes.cpu().numpy(), np.float64)
    agg_arcs = O.sparse_dense_matmul_adjoint(an.indices, an.values, np.array(an.shape), arcs.cpu().numpy()[:, 2:], np.float64)
    want = O.convergence(s.astype(np.float64), nodes.cpu().numpy().astype(np.float64), a, agg_nodes, agg_arcs,
                         ns.spec(), 32, False, np.float64)
    for flags in PATHS:
        model.native_flags = flags
        k1, new, old, *_ = model.convergence(0, dev(s), None, nodes, adj, None, None, False, arcs=arcs, arcnode=an)
        assert k1 == 1 and rel_err(new.cpu().numpy(), want) <= TOL


# ----------------------------------------------------------------------------------------------------------------------
# composite (heterogeneous) GNN
# ----------------------------------------------------------------------------------------------------------------------
def composite_nets(dims, A, D, T, focus, rng_seed=0, scale=0.5, hidden=None):
    inp, lay = get_inout_dims('state', dims, A, T, focus, D, hidden_units=hidden)
    ns = [MLP(i, lay, 'tanh', 'lecun_normal', 'lecun_normal', rng=rng_seed + t) for t, i in enumerate(inp)]
    for n in ns:
        n.set_weights([a * scale if a.ndim == 2 else a for a in n.get_weights()])
    inp, lay = get_inout_dims('output', dims, A, T, focus, D)
    no = MLP(inp[0], lay, 'softmax', 'glorot_normal', 'glorot_normal', rng=99)
    return ns, no


@pytest.mark.parametrize('focus', ['n', 'a', 'g'])
@pytest.mark.parametrize('mode', ['composite_average', 'average', 'sum'])
def test_composite_small_graphs(focus, mode):
    rng = np.random.default_rng(21)
    dims = (4, 2, 3)

    def cg(n, e):
        pairs = set()
        while len(pairs) < e:
            a, b = rng.integers(0, n, 2)
            if a != b: pairs.add((int(a), int(b)))
        ids = np.array(sorted(pairs), dtype=float)
        arcs = np.concatenate([ids, rng.normal(size=(e, 2))], 1)
        types = rng.integers(0, 3, n); types[:3] = [0, 1, 2]
        tm = np.zeros((n, 3), bool); tm[np.arange(n), types] = True
        nt = {'n': n, 'a': e, 'g': 1}[focus]
        return CompositeGraphObject(nodes=rng.normal(size=(n, 4)), arcs=arcs, targets=rng.normal(size=(nt, 2)),
                                    type_mask=tm, dim_node_label=dims, focus=focus, aggregation_mode=mode)
    gl = [cg(40, 130), cg(25, 60), cg(70, 200)]
    seq = CompositeMultiGraphSequencer(gl, focus, mode, 3, shuffle=False)
    x = seq[0][0]
    ns, no = composite_nets(dims, 2, 6, 2, focus)
    model = CCLS[focus](ns, no, 6, 7, 0.0)
    N = x[0].shape[0]
    check(model, x, rng.normal(0, .1, (N, 6)).astype(np.float32), oracle=oracle_composite_loop)


def test_composite_hidden_layers_and_state_dim_0():
    g = er_composite_graph(3000, 20000, dim_node_label=(5, 5, 5), aggregation_mode='composite_average', seed=3)
    seq = CompositeMultiGraphSequencer([g], 'n', 'composite_average', 1, shuffle=False)
    x = seq[0][0]
    ns, no = composite_nets((5, 5, 5), 3, 12, 2, 'n', hidden=[20])
    check(CompositeGNNnodeBased(ns, no, 12, 4, 0.0), x, np.random.default_rng(0).normal(0, .1, (3000, 12)).astype(np.float32),
          oracle=oracle_composite_loop)
    # state_vect_dim == 0: state0 = nodes (CompositeGNN.py:258), every net maps back to the label width
    inp = [(5 + 2 * 5 + 15 + 3,)] * 3
    ns0 = [MLP(i, [5], 'tanh', 'lecun_normal', 'zeros', rng=t) for t, i in enumerate(inp)]
    no0 = MLP((5,), [2], 'softmax', 'glorot_normal', 'zeros', rng=9)
    check(CompositeGNNnodeBased(ns0, no0, 0, 3, 0.0), x, None, oracle=oracle_composite_loop)


def test_composite_type_mask_must_be_one_hot():
    g = er_composite_graph(50, 200, dim_node_label=(3, 3), seed=1)
    seq = CompositeMultiGraphSequencer([g], 'n', 'sum', 1, shuffle=False)
    x = list(seq[0][0])
    tm = x[3].clone(); tm[:, 0] = True
    x[3] = tm
    ns, no = composite_nets((3, 3), 3, 4, 2, 'n')
    with pytest.raises(ValueError):
        CompositeGNNnodeBased(ns, no, 4, 2, 0.0)(x)


# ----------------------------------------------------------------------------------------------------------------------
# BASELINE full sizes: size-independent properties + oracle on a few iterations (fast scipy path)
# ----------------------------------------------------------------------------------------------------------------------
def c3_case(mode, iters=3, paths=(0,)):
    """C3 (100 k nodes / 1 M arcs, d = 64) for a few iterations against both oracles on the given paths.  (Every path for the 50 iterations the
    bench times: tests/test_gpu_round4.py::test_c3_at_the_timed_depth_every_path - this is what the kernel-variant tests below re-run.)"""
    N, E, d = 100_000, 1_000_000, 64
    g = er_graph(N, E, aggregation_mode=mode)
    x = MultiGraphSequencer([g], 'n', mode, 1, shuffle=False)[0][0]
    ns, no = starter_nets('n', d, scale=1.0 if mode == 'average' else 0.1)
    s0 = np.random.default_rng(1).normal(0, 0.1, (N, d)).astype(np.float32)
    model = GNNnodeBased(ns, no, d, iters, 0.0)
    k32, st32, o32 = oracle_loop(model, x, s0, np.float32, exact_order=False)
    k64, st64, o64 = oracle_loop(model, x, s0, np.float64, exact_order=False)
    for flags in paths:
        model.native_flags = flags
        k, st, o = model.Loop(*model.process_inputs(x), state0=dev(s0))
        assert float(k) == float(iters) == float(k32)
        assert rel_err(st.cpu().numpy(), st32) <= TOL and rel_err(st.cpu().numpy(), st64) <= TOL
        assert rel_err(o.cpu().numpy(), o32) <= TOL and rel_err(o.cpu().numpy(), o64) <= TOL
    return nat.lib().gnn_last_kernel_name().decode()


# (C4 / C5 at full size against the oracle on the default, un-fused and generation-2 paths, run-to-run determinism: the 50-iteration tests
# of tests/test_gpu_round4.py - test_c4_at_the_timed_depth_vs_fp64_oracle, test_c5_at_the_timed_depth_vs_fp64_oracle)


def test_expired_in_launch_wait_is_loud():
    """A lost slot hand-off in the wave-specialised kernel must reach the caller: libgnnloop_spin0.so is the same source
    built with -DGNN_F4_SPIN_MAX=0 (every bounded wait expires at once); the forward must come back with k < 0,
    check_last_k() must raise NativeError and predict() must recover on the kernels without such waits (round 4).  Runs in a
    child process (another build of the library)."""
    import os, subprocess, sys
    lib = os.path.join(nat.CSRC, 'libgnnloop_spin0.so')
    assert os.path.exists(lib), 'build() makes it'
    code = r"""
import numpy as np, torch, sys
from gnnkeras_amd import _native as nat
from gnnkeras_amd.synth import er_graph
from gnnkeras_amd.Models.MLP import MLP, get_inout_dims
from gnnkeras_amd.Models.GNN import GNNnodeBased
from gnnkeras_amd.Sequencers.GraphSequencers import MultiGraphSequencer
assert nat.LIB_PATH.endswith('libgnnloop_spin0.so')
N, E, d = 200_000, 2_000_000, 64
g = er_graph(N, E, aggregation_mode='average')
seq = MultiGraphSequencer([g], 'n', 'average', 1, shuffle=False)
inp, lay = get_inout_dims('state', 14, 3, 2, 'n', d); ns = MLP(inp[0], lay, 'selu', 'lecun_normal', 'lecun_normal', rng=0)
inp, lay = get_inout_dims('output', 14, 3, 2, 'n', d); no = MLP(inp[0], lay, 'softmax', 'glorot_normal', 'glorot_normal', rng=1)
m = GNNnodeBased(ns, no, d, 3, 0.0)
m.native_flags = nat.FLAG_FUSED_GEN4
k, st, o = m.Loop(*m.process_inputs(seq[0][0]), state0=torch.randn(N, d, device='cuda') * 0.1)
torch.cuda.synchronize()
assert float(k) < 0, float(k)
try:
    m.check_last_k()
except nat.NativeError:
    pass
else:
    sys.exit('check_last_k() did not raise')
# predict() / evaluate() recover: the walk is repeated on the phase-alternating kernel (one launch per iteration, no bounded
# hand-offs between waves), with a RuntimeWarning, and gives what that kernel gives when asked for directly
import warnings
with warnings.catch_warnings(record=True) as w:
    warnings.simplefilter('always')
    torch.manual_seed(7); got = m.predict(seq)
assert any(issubclass(x.category, RuntimeWarning) for x in w), [str(x.message) for x in w]
assert m.recovered_walks == 1 and m.native_flags == nat.FLAG_FUSED_GEN4, (getattr(m, 'recovered_walks', None), m.native_flags)
m.native_flags = nat.FLAG_FUSED_GEN2
torch.manual_seed(7); want = m.predict(seq)
assert m.recovered_walks == 1 and np.isfinite(got).all()
assert np.abs(got - want).max() <= 1e-5 * np.abs(want).max(), np.abs(got - want).max()      # (state_0 is drawn at random: same draws, see _with_recovery)
m.native_flags = nat.FLAG_FUSED_GEN4
# the same for the fused kernel of state widths 129 .. 256 (kernel_state_xwide.hpp: its hand-overs use the same bound)
N, E, d = 20_000, 100_000, 160
g = er_graph(N, E, aggregation_mode='average')
seq = MultiGraphSequencer([g], 'n', 'average', 1, shuffle=False)
inp, lay = get_inout_dims('state', 14, 3, 2, 'n', d); ns = MLP(inp[0], lay, 'tanh', 'lecun_normal', 'lecun_normal', rng=0)
inp, lay = get_inout_dims('output', 14, 3, 2, 'n', d); no = MLP(inp[0], lay, 'softmax', 'glorot_normal', 'glorot_normal', rng=1)
m = GNNnodeBased(ns, no, d, 3, 0.0)
k, st, o = m.Loop(*m.process_inputs(seq[0][0]), state0=torch.randn(N, d, device='cuda') * 0.1)
torch.cuda.synchronize()
assert 'k_state_xwide' in nat.lib().gnn_last_kernel_name().decode()
assert float(k) < 0, float(k)
print('LOUD_OK')
"""
    env = dict(os.environ, GNNKERAS_AMD_LIB=lib, PYTHONPATH=os.path.dirname(os.path.dirname(os.path.abspath(nat.HERE))) + os.pathsep + os.environ.get('PYTHONPATH', ''))
    root = os.path.dirname(nat.HERE)
    env['PYTHONPATH'] = root + os.pathsep + os.environ.get('PYTHONPATH', '')
    r = subprocess.run([sys.executable, '-c', code], env=env, capture_output=True, text=True, timeout=600, cwd=root)
    assert r.returncode == 0 and 'LOUD_OK' in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


# ----------------------------------------------------------------------------------------------------------------------
# Keras-style evaluate / predict on top of the loop
# ----------------------------------------------------------------------------------------------------------------------
def test_evaluate_and_predict(mutag_graphs):
    gs = [g.copy() for g in mutag_graphs[:96]]
    for g in gs: g.setAggregation('average')
    seq = MultiGraphSequencer(gs, 'g', 'average', 32, shuffle=False)
    ns, no = starter_nets('g', 0)
    model = GNNgraphBased(ns, no, 0, 5, 0.01)
    model.compile(optimizer=None, loss='categorical_crossentropy', metrics=['accuracy'])
    pred = model.predict(seq)
    assert pred.shape == (96, 2)
    outs, ys = [], []
    for i in range(len(seq)):
        x, y, sw = seq[i]
        outs.append(oracle_loop(model, x, None, np.float64)[2]); ys.append(y.cpu().numpy())
    want, y = np.concatenate(outs), np.concatenate(ys)
    assert rel_err(pred, want) <= TOL
    res = model.evaluate(seq, return_dict=True)
    loss = float(np.mean(-np.sum(y * np.log(np.clip(want, 1e-7, 1 - 1e-7)), axis=1)))
    acc = float(np.mean(want.argmax(1) == y.argmax(1)))
    assert abs(res['loss'] - loss) < 1e-5 and abs(res['accuracy'] - acc) < 1e-6


# ----------------------------------------------------------------------------------------------------------------------
# node-range sharded loop (multi-GPU path) with the native kernels: R shards emulated on ONE device, the all-gather
# replaced by explicit slice copies between the shards' full buffers (what RCCL does across GPUs)
# ----------------------------------------------------------------------------------------------------------------------
def _run_shards_on_one_gpu(model, g, s0, R, overlap=False):
    from gnnkeras_amd.distributed import ShardedLoop
    shards = [ShardedLoop(model, g, r, R, 'cuda', overlap=overlap) for r in range(R)]
    if overlap: assert all(sl.overlap for sl in shards), 'the library refused the own-range / halo split for this shard'
    for sl in shards:
        sl._load_state0(s0 if s0 is not None else sl.plan_nodes_as_state())
        sl._setup()
        sl._initial_flags()
    n = shards[0].plan.rows_per_slice * shards[0].SP
    for it in range(model.max_iteration):
        for sl in shards:
            if overlap: sl._partial(it); sl._iteration_split(it)      # phase A reads own rows only, phase B the halo
            else: sl._iteration(it)
        for r, src in enumerate(shards):                     # "all-gather": slice r of rank r's buffer -> everyone
            piece = src.buf[(it + 1) & 1].view(-1)[r * n:(r + 1) * n]
            for dst in shards:
                if dst is not src: dst.buf[(it + 1) & 1].view(-1)[r * n:(r + 1) * n].copy_(piece)
    outs = [sl._output() for sl in shards]
    torch.cuda.synchronize()
    ks = [float(o[0]) for o in outs]
    return ks, np.concatenate([o[1].cpu().numpy() for o in outs]), np.concatenate([o[2].cpu().numpy() for o in outs])


_SHARD_CASES = {}


def _sharded_native_case(threshold):
    """Graph, model, state_0 and the float64 oracle of test_sharded_native_kernels_match_oracle: the same for every shard count."""
    if threshold not in _SHARD_CASES:
        rng = np.random.default_rng(0)
        N, d = 5003, 64
        g = er_graph(N, 40000, seed=7)
        om = rng.random(N) < 0.7
        g = GraphObject(g.nodes, g.arcs, rng.normal(size=(int(om.sum()), 2)), focus='n', set_mask=rng.random(N) < 0.8,
                        output_mask=om, aggregation_mode='average')
        ns, no = starter_nets('n', d, scale=0.3)
        model = GNNnodeBased(ns, no, d, 12, threshold)
        s0 = rng.normal(0, 0.1, (N, d)).astype(np.float32)
        x = MultiGraphSequencer([g], 'n', 'average', 1, shuffle=False)[0][0]
        _SHARD_CASES[threshold] = (g, model, s0, oracle_loop(model, x, s0, np.float64))
    return _SHARD_CASES[threshold]


@pytest.mark.parametrize('R', [1, 2, 3, 8])
@pytest.mark.parametrize('threshold', [0.0, 0.02])
def test_sharded_native_kernels_match_oracle(R, threshold):
    g, model, s0, (k64, st64, o64) = _sharded_native_case(threshold)
    if threshold > 0: assert 1 < k64 < 12
    for flags in PATHS:
        model.native_flags = flags
        ks, st, o = _run_shards_on_one_gpu(model, g, s0, R)
        assert all(k == float(k64) for k in ks), (ks, k64)
        assert rel_err(st, st64) <= TOL and rel_err(o, o64) <= TOL


def test_sharded_overlap_split_wide_state():
    """The own-range / halo split on the wide kernel (d = 128)."""
    rng = np.random.default_rng(3)
    N, d = 20_003, 128
    g = er_graph(N, 160_000, seed=7, aggregation_mode='average')
    ns, no = starter_nets('n', d, scale=0.25)
    model = GNNnodeBased(ns, no, d, 5, 0.0)
    s0 = rng.normal(0, 0.1, (N, d)).astype(np.float32)
    x = MultiGraphSequencer([g], 'n', 'average', 1, shuffle=False)[0][0]
    k64, st64, o64 = oracle_loop(model, x, s0, np.float64, exact_order=False)
    for R in (1, 4):
        ks, st, o = _run_shards_on_one_gpu(model, g, s0, R, overlap=True)
        assert all(k == float(k64) for k in ks) and rel_err(st, st64) <= TOL and rel_err(o, o64) <= TOL


@pytest.mark.parametrize('R', [1, 2, 8])
@pytest.mark.parametrize('mode,threshold', [('average', 0.0), ('average', 0.02), ('sum', 0.0)])
def test_sharded_overlap_split_matches_oracle(R, mode, threshold):
    """Own-range / halo split of the iteration (gnn_shard_partial + gnn_shard_iteration_split, the overlap path of
    distributed.py) on emulated shards: same oracle, same 1e-5 bar, k identical."""
    key = ('overlap', mode, threshold)
    if key not in _SHARD_CASES:           # (graph, model and the float64 oracle are the same for every shard count)
        rng = np.random.default_rng(0)
        N, d = 40_003, 64
        g = er_graph(N, 400_000, seed=7, aggregation_mode=mode)
        ns, no = starter_nets('n', d, scale=0.3 if mode == 'average' else 0.03)
        model = GNNnodeBased(ns, no, d, 6, threshold)
        s0 = rng.normal(0, 0.1, (N, d)).astype(np.float32)
        x = MultiGraphSequencer([g], 'n', mode, 1, shuffle=False)[0][0]
        _SHARD_CASES[key] = (g, model, s0, oracle_loop(model, x, s0, np.float64, exact_order=False))
    g, model, s0, (k64, st64, o64) = _SHARD_CASES[key]
    model.native_flags = 0
    ks, st, o = _run_shards_on_one_gpu(model, g, s0, R, overlap=True)
    assert all(k == float(k64) for k in ks), (ks, k64)
    assert rel_err(st, st64) <= TOL and rel_err(o, o64) <= TOL
    ks2, st2, o2 = _run_shards_on_one_gpu(model, g, s0, R, overlap=False)
    assert ks2 == ks and rel_err(st, st2) <= TOL


def test_sharded_overlap_per_arc_weights_and_composite():
    """The split with per-arc weights (w != NULL) and with per-type state networks (C5 shape)."""
    rng = np.random.default_rng(1)
    N, d = 36_000, 32
    g = er_graph(N, 300_000, seed=9, aggregation_mode='sum')
    an = g.getArcNode(); an.data = rng.uniform(0.01, 0.1, len(an.data)).astype(np.float32)
    g = GraphObject(g.nodes, g.arcs, g.targets, focus='n', ArcNode=an)
    ns, no = starter_nets('n', d, scale=0.3)
    model = GNNnodeBased(ns, no, d, 4, 0.0)
    s0 = rng.normal(0, 0.1, (N, d)).astype(np.float32)
    ks, st, o = _run_shards_on_one_gpu(model, g, s0, 1, overlap=False)
    ks2, st2, o2 = _run_shards_on_one_gpu(model, g, s0, 1, overlap=True)      # one shard: every arc is own-range
    ks4, st4, o4 = _run_shards_on_one_gpu(model, g, s0, 4, overlap=True)
    assert ks == ks2 == [4.0] and ks4 == [4.0] * 4
    assert rel_err(st2, st) <= TOL and rel_err(st4, st) <= TOL and rel_err(o4, o) <= TOL
    dims = (5, 3, 4)
    gc = er_composite_graph(N, 300_000, dim_node_label=dims, aggregation_mode='composite_average', seed=11)
    nsc, noc = composite_nets(dims, 3, d, 2, 'n')
    mc = CompositeGNNnodeBased(nsc, noc, d, 4, 0.0)
    xc = CompositeMultiGraphSequencer([gc], 'n', 'composite_average', 1, shuffle=False)[0][0]
    k64, st64, o64 = oracle_composite_loop(mc, xc, s0, np.float64, exact_order=False)
    ksc, stc, oc = _run_shards_on_one_gpu(mc, gc, s0, 4, overlap=True)
    assert all(k == float(k64) for k in ksc) and rel_err(stc, st64) <= TOL and rel_err(oc, o64) <= TOL


def test_sharded_state_dim_0_and_per_arc_weights():
    rng = np.random.default_rng(1)
    N = 3000
    g = er_graph(N, 20000, seed=9, aggregation_mode='sum')
    an = g.getArcNode(); an.data = rng.uniform(0.05, 0.3, len(an.data)).astype(np.float32)
    g = GraphObject(g.nodes, g.arcs, g.targets, focus='n', ArcNode=an)
    ns, no = starter_nets('n', 0)
    model = GNNnodeBased(ns, no, 0, 5, 0.0)
    seq = MultiGraphSequencer([g], 'n', 'sum', 1, shuffle=False)
    seq.graph_tensors[0].ArcNode = SparseMatrix.from_scipy(g.ArcNode)
    seq.graph_tensors[0].Adjacency = SparseMatrix.from_scipy(g.Adjacency)
    seq._items = [None]
    k64, st64, o64 = oracle_loop(model, seq[0][0], None, np.float64)
    ks, st, o = _run_shards_on_one_gpu(model, g, None, 4)
    assert all(k == 5.0 for k in ks) and rel_err(st, st64) <= TOL and rel_err(o, o64) <= TOL


@pytest.mark.parametrize('R', [1, 4])
@pytest.mark.parametrize('mode', ['average', 'composite_average'])
def test_sharded_composite_native_kernels_match_oracle(R, mode):
    """BASELINE config C5 shape (3 node types, per-type state networks) on emulated node-range shards."""
    rng = np.random.default_rng(4)
    N, dims, d = 4001, (5, 3, 4), 32
    g = er_composite_graph(N, 30000, dim_node_label=dims, aggregation_mode=mode, seed=11)
    ns, no = composite_nets(dims, 3, d, 2, 'n')
    model = CompositeGNNnodeBased(ns, no, d, 8, 0.0)
    s0 = rng.normal(0, 0.1, (N, d)).astype(np.float32)
    x = CompositeMultiGraphSequencer([g], 'n', mode, 1, shuffle=False)[0][0]
    k64, st64, o64 = oracle_composite_loop(model, x, s0, np.float64)
    for flags in PATHS:
        model.native_flags = flags
        ks, st, o = _run_shards_on_one_gpu(model, g, s0, R)
        assert all(k == float(k64) for k in ks)
        assert rel_err(st, st64) <= TOL and rel_err(o, o64) <= TOL


def _run_halo_shards_on_one_gpu(model, g, s0, R):
    """R compacted-halo shards on one device; the all-to-all is emulated by copying every packed segment into the
    receiver's halo region (what RCCL does pair-wise across GPUs)."""
    from gnnkeras_amd.distributed import HaloShardedLoop
    shards = [HaloShardedLoop(model, g, r, R, 'cuda') for r in range(R)]
    for sl in shards:
        sl._load_state0(s0 if s0 is not None else sl._graph_nodes_full)
        sl._setup()
        sl._initial_flags()
    for it in range(model.max_iteration):
        nxt = (it + 1) & 1
        for sl in shards: sl._iteration(it)
        for sl in shards: sl._pack(sl.buf[nxt])
        for p_ in shards:
            for r_ in shards:
                if r_ is p_: continue
                cnt = r_.plan.send_counts[p_.rank]
                off = sum(r_.plan.send_counts[:p_.rank])
                start = p_.plan.seg_start[r_.rank]
                assert cnt == p_.plan.recv_rows[r_.rank]
                p_.buf[nxt][start:start + cnt].copy_(r_.sendbuf[off:off + cnt])
        for sl in shards: sl.gates[nxt].copy_(sl._flag_words(sl.buf[nxt]))
    outs = [sl._output() for sl in shards]
    torch.cuda.synchronize()
    return [float(o[0]) for o in outs], np.concatenate([o[1].cpu().numpy() for o in outs]), \
        np.concatenate([o[2].cpu().numpy() for o in outs])


@pytest.mark.parametrize('R', [1, 3, 8])
@pytest.mark.parametrize('threshold', [0.0, 0.02])
def test_halo_sharded_native_kernels_match_oracle(R, threshold):
    rng = np.random.default_rng(0)
    N, d = 5003, 64
    g = er_graph(N, 40000, seed=7)
    om = rng.random(N) < 0.7
    g = GraphObject(g.nodes, g.arcs, rng.normal(size=(int(om.sum()), 2)), focus='n', set_mask=rng.random(N) < 0.8,
                    output_mask=om, aggregation_mode='average')
    ns, no = starter_nets('n', d, scale=0.3)
    model = GNNnodeBased(ns, no, d, 12, threshold)
    s0 = rng.normal(0, 0.1, (N, d)).astype(np.float32)
    x = MultiGraphSequencer([g], 'n', 'average', 1, shuffle=False)[0][0]
    k64, st64, o64 = oracle_loop(model, x, s0, np.float64)
    for flags in PATHS:
        model.native_flags = flags
        ks, st, o = _run_halo_shards_on_one_gpu(model, g, s0, R)
        assert all(k == float(k64) for k in ks), (ks, k64)
        assert rel_err(st, st64) <= TOL and rel_err(o, o64) <= TOL


def test_halo_sharded_mutag_batch_by_graph_and_composite(mutag_graphs):
    """A block-diagonal MUTAG batch sharded near graph boundaries exchanges (almost) nothing but flags; composite too."""
    from gnnkeras_amd.distributed import HaloShardPlan, make_sharded_loop
    m = GraphObject.merge(mutag_graphs[:32], 'n', 'average')
    g = GraphObject(m.nodes, m.arcs, np.zeros((m.nodes.shape[0], 2)), focus='n', aggregation_mode='average')
    plan = HaloShardPlan(g, 1, 4)
    assert sum(len(h) for h in plan.halo) < 0.15 * plan.n_local           # a graph may straddle a cut, nothing more
    assert type(make_sharded_loop(GNNnodeBased(*starter_nets('n', 8), 8, 3, 0.0), g, 0, 4, 'cuda')).__name__ == 'HaloShardedLoop'
    ns, no = starter_nets('n', 32)
    model = GNNnodeBased(ns, no, 32, 10, 0.0)
    s0 = np.random.default_rng(2).normal(0, 0.1, (g.nodes.shape[0], 32)).astype(np.float32)
    x = MultiGraphSequencer([g], 'n', 'average', 1, shuffle=False)[0][0]
    k64, st64, o64 = oracle_loop(model, x, s0, np.float64)
    ks, st, o = _run_halo_shards_on_one_gpu(model, g, s0, 4)
    assert all(k == 10.0 for k in ks) and rel_err(st, st64) <= TOL and rel_err(o, o64) <= TOL
    # composite (C5 shape)
    dims = (5, 3, 4)
    cg = er_composite_graph(3001, 20000, dim_node_label=dims, aggregation_mode='composite_average', seed=5)
    cns, cno = composite_nets(dims, 3, 16, 2, 'n')
    cmodel = CompositeGNNnodeBased(cns, cno, 16, 6, 0.0)
    cs0 = np.random.default_rng(3).normal(0, 0.1, (3001, 16)).astype(np.float32)
    cx = CompositeMultiGraphSequencer([cg], 'n', 'composite_average', 1, shuffle=False)[0][0]
    ck, cst, co = oracle_composite_loop(cmodel, cx, cs0, np.float64)
    ks, st, o = _run_halo_shards_on_one_gpu(cmodel, cg, cs0, 3)
    assert all(k == float(ck) for k in ks) and rel_err(st, cst) <= TOL and rel_err(o, co) <= TOL


# ----------------------------------------------------------------------------------------------------------------------
# hub nodes: rows with more than sparse.HEAVY_THRESHOLD in-arcs are summed by a whole-workgroup pre-pass
# ----------------------------------------------------------------------------------------------------------------------
def _hub_graph(rng, n, e, hubs, mode, weights=False):
    from gnnkeras_amd.synth import er_arcs
    ids = er_arcs(n, e, seed=3)
    extra = [np.stack([rng.choice(n - 1, deg, replace=False) + (1 if h == 0 else 0) * 0, np.full(deg, h)], 1) for h, deg in hubs]
    extra = [x[x[:, 0] != x[0, 1]] for x in extra]
    ids = np.unique(np.concatenate([ids] + extra), axis=0)
    arcs = np.concatenate([ids.astype(np.float64), np.eye(3)[rng.integers(0, 3, len(ids))]], axis=1)
    nodes = np.eye(14)[rng.integers(0, 14, n)]
    g = GraphObject(nodes, arcs, np.zeros((n, 2)), focus='n', aggregation_mode=mode)
    if weights:
        an = g.getArcNode(); an.data = rng.uniform(0.001, 0.01, len(an.data)).astype(np.float32)
        g = GraphObject(nodes, arcs, np.zeros((n, 2)), focus='n', ArcNode=an)
    return g


@pytest.mark.parametrize('mode,weights,d', [('average', False, 64), ('average', False, 32), ('sum', True, 64), ('average', False, 0),
                                            ('average', False, 128), ('sum', True, 96), ('average', False, 200)])
def test_hub_rows_use_the_segment_prepass(mode, weights, d):
    from gnnkeras_amd import sparse
    rng = np.random.default_rng(17)
    g = _hub_graph(rng, 20000, 150000, [(5, 700), (1234, 3000), (19999, 9000)], mode, weights)
    seq = MultiGraphSequencer([g], 'n', mode, 1, shuffle=False)
    if weights:
        seq.graph_tensors[0].ArcNode = SparseMatrix.from_scipy(g.ArcNode)
        seq.graph_tensors[0].Adjacency = SparseMatrix.from_scipy(g.Adjacency)
        seq._items = [None]
    x = seq[0][0]
    dc = x[5].matrix.device_csr('cuda')
    assert dc['max_degree'] > sparse.HEAVY_THRESHOLD and dc['heavy'] is not None
    assert dc['heavy']['n_seg'] == 1 + 2 + 5 and dc['light']['n_src'] == 20000 + 8     # ceil(deg / 2048) segments per hub
    ns, no = starter_nets('n', d, scale=0.5)
    model = GNNnodeBased(ns, no, d, 6, 0.0)
    s0 = rng.normal(0, 0.1, (20000, d)).astype(np.float32) if d else None
    k64, st64, o64 = oracle_loop(model, x, s0, np.float64, exact_order=False)
    for flags in PATHS:
        model.native_flags = flags
        k, st, o = model.Loop(*model.process_inputs(x), state0=None if s0 is None else dev(s0))
        assert float(k) == float(k64)
        assert rel_err(st.cpu().numpy(), st64) <= TOL and rel_err(o.cpu().numpy(), o64) <= TOL
    if d:        # the standalone step (convergence(): gnn_state_step skips the pointer validation) takes the same hub path
        model.native_flags = 0
        step = model.convergence(0, dev(s0), None, x[0], x[5], None, None, False, arcs=x[1], arcnode=x[6])[1]
        one = GNNnodeBased(ns, no, d, 1, 0.0)
        st1 = oracle_loop(one, x, s0, np.float64, exact_order=False)[1]
        assert rel_err(step.cpu().numpy(), st1) <= TOL


@pytest.mark.parametrize('d,mode,weights', [(128, 'average', False), (96, 'sum', False), (65, 'average', False), (128, 'sum', True), (100, 'normalized', False)])
def test_wide_state_runs_fused(d, mode, weights):
    """State widths 65 .. 128 (any `state_vect_dim` is legal in the reference, GNN.py:26-28) run the wide fused kernel
    (kernel_state_wide.hpp: one workgroup per CU, W1 = 128 KB in LDS): against the oracle and against the un-fused kernels,
    with early exit, masks and a ragged last tile."""
    rng = np.random.default_rng(23)
    N = 20_011
    g = er_graph(N, 8 * N, seed=5, aggregation_mode=mode)
    if weights:
        an = g.getArcNode(); an.data = rng.uniform(0.01, 0.12, len(an.data)).astype(np.float32)
        g = GraphObject(g.nodes, g.arcs, g.targets, focus='n', ArcNode=an)
    om = rng.random(N) < 0.7
    g = GraphObject(g.nodes, g.arcs, rng.normal(size=(int(om.sum()), 2)), focus='n', set_mask=rng.random(N) < 0.8, output_mask=om,
                    aggregation_mode=mode, **({'ArcNode': g.ArcNode} if weights else {}))
    ns, no = starter_nets('n', d, scale=0.25 if mode != 'sum' else 0.03)
    seq = MultiGraphSequencer([g], 'n', mode, 1, shuffle=False)
    if weights:
        seq.graph_tensors[0].ArcNode = SparseMatrix.from_scipy(g.ArcNode)
        seq.graph_tensors[0].Adjacency = SparseMatrix.from_scipy(g.Adjacency)
        seq._items = [None]
    x = seq[0][0]
    s0 = rng.normal(0, 0.1, (N, d)).astype(np.float32)
    for thr, iters in ((0.0, 4), (0.05, 30)):
        model = GNNnodeBased(ns, no, d, iters, thr)
        k64, st64, o64 = oracle_loop(model, x, s0, np.float64, exact_order=False)
        if thr > 0: assert 1 < k64 < iters, k64
        inputs = model.process_inputs(x)
        for flags in (0, nat.FLAG_UNFUSED, nat.FLAG_FUSED_GEN4):
            model.native_flags = flags
            k, st, o = model.Loop(*inputs, state0=dev(s0))
            assert float(k) == float(k64), (flags, float(k), k64)
            assert rel_err(st.cpu().numpy(), st64) <= TOL and rel_err(o.cpu().numpy(), o64) <= TOL, (flags, rel_err(st.cpu().numpy(), st64))
            if flags == 0: assert nat.lib().gnn_last_kernel_name().decode().startswith('k_state_wide'), nat.lib().gnn_last_kernel_name()
        assert model.check_last_k() == float(k64)


def test_wide_state_small_graphs_and_composite(mutag_graphs):
    """The wide kernel on a MUTAG batch (59 tiles: most workgroups idle) and on a 3-type composite graph, d = 128 / 80."""
    seq = MultiGraphSequencer(mutag_graphs[:32], 'g', 'average', 32, shuffle=False)
    x = seq[0][0]
    ns, no = starter_nets('g', 128, scale=0.25)
    model = GNNgraphBased(ns, no, 128, 10, 0.0)
    s0 = np.random.default_rng(1).normal(0, 0.1, (x[0].shape[0], 128)).astype(np.float32)
    check(model, x, s0)
    rng = np.random.default_rng(4)
    N, dims, d = 6001, (5, 3, 4), 80
    g = er_composite_graph(N, 40000, dim_node_label=dims, aggregation_mode='composite_average', seed=11)
    nsc, noc = composite_nets(dims, 3, d, 2, 'n')
    mc = CompositeGNNnodeBased(nsc, noc, d, 6, 0.0)
    xc = CompositeMultiGraphSequencer([g], 'n', 'composite_average', 1, shuffle=False)[0][0]
    check(mc, xc, rng.normal(0, 0.1, (N, d)).astype(np.float32), oracle=oracle_composite_loop)


@pytest.mark.parametrize('d,hidden,mode', [(64, 48, 'average'), (32, 32, 'sum'), (64, 64, 'normalized')])
def test_two_layer_state_network_runs_fused_at_scale(d, hidden, mode):
    """A state network with one hidden layer (reference MLP(hidden_units=...), MLP.py:82-140) on a graph large enough
    for the wave-specialised kernel: the second Dense runs inside the matrix waves; every path must agree with the oracle."""
    rng = np.random.default_rng(11)
    N = 40000
    g = er_graph(N, 6 * N, seed=3, aggregation_mode=mode)
    ns, no = starter_nets('n', d, hidden_state=[hidden], act='tanh', scale=0.3)
    thr = 0.02 if mode == 'average' else 0.0
    model = GNNnodeBased(ns, no, d, 8, thr)
    s0 = rng.normal(0, 0.1, (N, d)).astype(np.float32)
    x = MultiGraphSequencer([g], 'n', mode, 1, shuffle=False)[0][0]
    k64, st64, o64 = oracle_loop(model, x, s0, np.float64)
    inputs = model.process_inputs(x)
    for flags in (0, nat.FLAG_UNFUSED, nat.FLAG_FUSED_GEN4):
        model.native_flags = flags
        k, st, o = model.Loop(*inputs, state0=torch.from_numpy(s0).cuda())
        assert float(k) == float(k64), (flags, float(k), k64)
        assert rel_err(st.cpu().numpy(), st64) <= TOL and rel_err(o.cpu().numpy(), o64) <= TOL, flags


def test_predict_and_evaluate_with_concurrent_batches_match_one_stream(mutag_graphs):
    """predict() / evaluate() spread the batches of a sequencer over side HIP streams; results must equal the
    one-stream walk bit for bit (explicit per-batch state_0 through a seeded generator is not available in call(), so
    state_vect_dim = 0 makes the forward deterministic)."""
    gl = [g.copy() for g in mutag_graphs[:32 * 12]]
    seq = MultiGraphSequencer(gl, 'g', 'average', 32, shuffle=False)
    ns, no = starter_nets('g', 0)
    model = GNNgraphBased(ns, no, 0, 20, 0.001)
    model.compile(optimizer='adam', loss='categorical_crossentropy', metrics=['accuracy'])
    model.inference_streams = 1
    p1 = model.predict(seq); e1 = model.evaluate(seq, return_dict=True)
    model.inference_streams = 8
    assert model._round_width(seq, torch.device('cuda', 0)) == 8
    for _ in range(3):
        p8 = model.predict(seq); e8 = model.evaluate(seq, return_dict=True)
        assert np.array_equal(p1, p8)
        assert abs(e1['loss'] - e8['loss']) <= 1e-6 and abs(e1['accuracy'] - e8['accuracy']) <= 1e-6


# ----------------------------------------------------------------------------------------------------------------------
# small graphs: one set-up launch (kernels_setup.hpp) + the whole-loop kernel, which also writes the caller's state
# ----------------------------------------------------------------------------------------------------------------------
def _last_kernel():
    return nat.lib().gnn_last_kernel_name().decode()


def _random_graph(rng, n, e, L, A, focus='n', mode='average'):
    nodes = rng.normal(size=(n, L))
    ends = rng.integers(0, n, size=(e, 2))
    ends = np.unique(ends[ends[:, 0] != ends[:, 1]], axis=0) if e else np.zeros((0, 2), int)
    arcs = np.concatenate([ends, rng.normal(size=(len(ends), A))], axis=1)
    nt = {'n': n, 'a': len(arcs), 'g': 1}[focus]
    return GraphObject(nodes=nodes, arcs=arcs, targets=rng.normal(size=(nt, 2)), focus=focus, aggregation_mode=mode)


@pytest.mark.parametrize('n', [1, 63, 64, 65, 700])
@pytest.mark.parametrize('mode', ['sum', 'average'])
def test_whole_loop_setup_tile_edges(n, mode):
    """Tile boundaries of the 64-node set-up / loop tiles, with and without BatchNormalization, every path."""
    rng = np.random.default_rng(n)
    g = _random_graph(rng, n, 4 * n, 5, 2, mode=mode)
    x = MultiGraphSequencer([g], 'n', mode, 1, shuffle=False)[0][0]
    for bn in (True, False):
        ns, no = starter_nets('n', 32, L=5, A=2, scale=0.4, bn=bn)
        model = GNNnodeBased(ns, no, 32, 6, 0.0)
        check(model, x, rng.normal(0, 0.1, (n, 32)).astype(np.float32))
        model.native_flags = 0
        model.Loop(*model.process_inputs(x), state0=dev(rng.normal(0, 0.1, (n, 32)).astype(np.float32)))
        assert _last_kernel().startswith('k_state_small'), _last_kernel()


def test_whole_loop_setup_state_dim_0_wide_labels():
    """state_dim = 0 with 20 label columns: the state IS the label matrix (padded to 32), the first layer has no label /
    neighbour-label segments (GNN.py:222-231), and the set-up kernel evaluates state_0's predicate on the labels."""
    rng = np.random.default_rng(20)
    g = _random_graph(rng, 300, 1500, 20, 3, focus='g')
    x = MultiGraphSequencer([g], 'g', 'average', 1, shuffle=False)[0][0]
    ns, no = starter_nets('g', 0, L=20, A=3, scale=0.4)
    for thr in (0.0, 0.01):
        model = GNNgraphBased(ns, no, 0, 7, thr)
        check(model, x, None)
    model.native_flags = 0
    model.Loop(*model.process_inputs(x))
    assert _last_kernel().startswith('k_state_small'), _last_kernel()


def test_whole_loop_never_started_returns_state_0():
    """A threshold no node passes at k = 0 (GNN.py:196-214): the loop kernel runs no iteration and must still deliver
    state_0 to the caller's buffer, and the output network's answer on it."""
    rng = np.random.default_rng(3)
    g = _random_graph(rng, 200, 900, 4, 2)
    x = MultiGraphSequencer([g], 'n', 'sum', 1, shuffle=False)[0][0]
    ns, no = starter_nets('n', 64, L=4, A=2)
    s0 = rng.normal(0, .1, (200, 64)).astype(np.float32)
    k, st, o = check(GNNnodeBased(ns, no, 64, 10, 100.0), x, s0)
    assert k == 0.0 and np.array_equal(st, s0)
    # converged after the first iterations with a realistic threshold: the state of the LAST iteration, not of max_iteration
    ns, no = starter_nets('n', 64, L=4, A=2, scale=0.05)
    k, st, o = check(GNNnodeBased(ns, no, 64, 30, 0.05), x, s0)
    assert 1 <= k < 30


@pytest.mark.parametrize('n,d,mode,thr', [(20000, 64, 'average', 0.0), (50000, 32, 'sum', 0.0), (30001, 64, 'average', 0.02), (70000, 40, 'normalized', 0.0)])
def test_mid_size_whole_loop_kernel(n, d, mode, thr):
    """Graphs between the small whole-loop kernel's range and the large-graph kernels: the whole loop in one launch with
    several 64-node tiles per workgroup and a two-level grid barrier between iterations (kernel_state_mid.hpp).  The
    automatic choice picks it up to 36 000 nodes (pinned beyond), and it must agree with the oracle and with one launch
    per iteration."""
    rng = np.random.default_rng(n)
    g = er_graph(n, 8 * n, seed=5, aggregation_mode=mode)
    ns, no = starter_nets('n', d, scale=0.3)
    model = GNNnodeBased(ns, no, d, 9, thr)
    s0 = rng.normal(0, 0.1, (n, d)).astype(np.float32)
    x = MultiGraphSequencer([g], 'n', mode, 1, shuffle=False)[0][0]
    k64, st64, o64 = oracle_loop(model, x, s0, np.float64)
    inputs = model.process_inputs(x)
    res = {}
    for flags in (0, nat.FLAG_FUSED_GEN6, nat.FLAG_FUSED_GEN2, nat.FLAG_UNFUSED):
        model.native_flags = flags
        k, st, o = model.Loop(*inputs, state0=torch.from_numpy(s0).cuda())
        assert float(k) == float(k64), (flags, float(k), k64)
        assert rel_err(st.cpu().numpy(), st64) <= TOL and rel_err(o.cpu().numpy(), o64) <= TOL, flags
        if flags == nat.FLAG_FUSED_GEN6 or (flags == 0 and n <= 36000): assert _last_kernel().startswith('k_state_mid'), _last_kernel()
        res[flags] = st
    # bit-stable from run to run (fixed summation order inside a tile, barrier-separated iterations)
    model.native_flags = 0
    k, st, o = model.Loop(*inputs, state0=torch.from_numpy(s0).cuda())
    assert torch.equal(st, res[0])


def test_mid_size_whole_loop_weighted_and_composite():
    """Per-arc weights (4-wave shape) and a 3-type composite graph (one launch serves every type's tiles)."""
    rng = np.random.default_rng(8)
    n, d = 25000, 32
    g = er_graph(n, 6 * n, seed=2, aggregation_mode='sum')
    an = g.getArcNode(); an.data = rng.uniform(0.2, 1.0, len(an.data)).astype(np.float32)
    g = GraphObject(nodes=g.nodes, arcs=g.arcs, targets=g.targets, focus='n', ArcNode=an)
    seq = MultiGraphSequencer([g], 'n', 'sum', 1, shuffle=False)
    seq.graph_tensors[0].ArcNode = SparseMatrix.from_scipy(g.ArcNode)
    seq.graph_tensors[0].Adjacency = SparseMatrix.from_scipy(g.Adjacency)
    seq._items = [None]
    x = seq[0][0]
    assert x[5].matrix.csr().w is not None
    ns, no = starter_nets('n', d, scale=0.2)
    model = GNNnodeBased(ns, no, d, 6, 0.0)
    s0 = rng.normal(0, 0.1, (n, d)).astype(np.float32)
    check(model, x, s0)
    model.native_flags = 0
    model.Loop(*model.process_inputs(x), state0=dev(s0))
    assert _last_kernel().startswith('k_state_mid<32,true'), _last_kernel()
    N, dims = 30000, (5, 3, 4)
    gc = er_composite_graph(N, 200000, dim_node_label=dims, aggregation_mode='composite_average', seed=11)
    nsc, noc = composite_nets(dims, 3, 64, 2, 'n')
    mc = CompositeGNNnodeBased(nsc, noc, 64, 6, 0.0)
    xc = CompositeMultiGraphSequencer([gc], 'n', 'composite_average', 1, shuffle=False)[0][0]
    check(mc, xc, rng.normal(0, 0.1, (N, 64)).astype(np.float32), oracle=oracle_composite_loop)
    mc.native_flags = 0
    mc.Loop(*mc.process_inputs(xc), state0=dev(rng.normal(0, 0.1, (N, 64)).astype(np.float32)))
    assert _last_kernel().startswith('k_state_mid'), _last_kernel()


# ----------------------------------------------------------------------------------------------------------------------
# convergence groups: several batches merged into one call, each with its own `while` (include/gnnloop.h group_node_begin)
# ----------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('focus,d,thr,hidden', [('g', 32, 0.01, None), ('n', 64, 0.02, None), ('a', 32, 0.0, None), ('g', 32, 0.01, [24]),
                                                ('g', 0, 0.01, None)])
def test_convergence_groups_equal_batch_by_batch_calls(mutag_graphs, focus, d, thr, hidden):
    """Eight MUTAG batches as eight independent loops of ONE launch: per batch the iteration count, the state and the
    output must equal - bit for bit, the tiles and summation orders are the same - what a call on that batch alone
    gives, including batches that converge at different k."""
    gl = refocus(mutag_graphs[:8 * 20], focus, np.random.default_rng(3))
    seq = MultiGraphSequencer(gl, focus, 'average', 20, shuffle=False)
    ns, no = starter_nets(focus, d, scale=0.22, hidden_state=hidden, act='tanh' if hidden else 'selu')
    model = CLS[focus](ns, no, d, 30, thr)
    rng = np.random.default_rng(9)
    parts = []
    for i in range(len(seq)):
        x = seq[i][0]
        # different scales: different k per batch (state_vect_dim = 0: the state starts from the labels, no state_0 to pass)
        s0 = rng.normal(0, 0.1 * (1 + i), (x[0].shape[0], d)).astype(np.float32) if d else None
        k, st, o = model.Loop(*model.process_inputs(x), state0=None if s0 is None else dev(s0))
        parts.append((float(k), st, o, s0))
    x, begin = seq.merged_batches(0, len(seq))
    assert begin[-1] == x[0].shape[0] and len(begin) == len(seq) + 1
    from gnnkeras_amd import ops
    assert ops.loop_groups_supported(begin[-1], x[0].shape[1], x[1].shape[1] - 2, ns, no, d, 30, nat.FOCUS[focus], 0, 0, begin)
    s0_all = dev(np.concatenate([p[3] for p in parts])) if d else None
    # the form that spreads every group over several CUs (pinned): bit for bit the batch-by-batch results
    model.native_flags = nat.FLAG_FUSED_GEN5
    k, st, o = model.Loop(*model.process_inputs(x), state0=s0_all, groups=begin)
    assert _last_kernel().startswith('k_state_small'), _last_kernel()
    assert k.shape == (len(seq),)
    assert [float(v) for v in k.cpu()] == [p[0] for p in parts]
    if thr > 0 and d: assert len({p[0] for p in parts}) > 1, 'the batches were meant to stop at different iterations'
    assert torch.equal(st, torch.cat([p[1] for p in parts]))
    assert torch.equal(o, torch.cat([p[2] for p in parts]))
    assert np.array_equal(model.check_last_k(), np.array([p[0] for p in parts], np.float32))
    # the automatic choice: one CU per group with its state in LDS where a group fits (width <= 32, one-layer network) - other
    # summation orders, so within the parity tolerance instead of bit-equal; same iteration counts
    model.native_flags = 0
    k2, st2, o2 = model.Loop(*model.process_inputs(x), state0=s0_all, groups=begin)
    lds = d <= 32 and hidden is None
    assert _last_kernel().startswith('k_state_lds' if lds else 'k_state_small'), _last_kernel()
    assert [float(v) for v in k2.cpu()] == [p[0] for p in parts]
    assert rel_err(st2.cpu().numpy(), st.cpu().numpy()) <= TOL and rel_err(o2.cpu().numpy(), o.cpu().numpy()) <= TOL
    k, st, o = k2, st2, o2
    # ... and against the oracle on one of the batches
    i = 3
    k64, st64, o64 = oracle_loop(model, seq[i][0], parts[i][3], np.float64)
    assert float(k[i]) == float(k64) and rel_err(st[begin[i]:begin[i + 1]].cpu().numpy(), st64) <= TOL


def test_many_groups_one_cu_each_with_the_state_in_lds(mutag_graphs):
    """The 136 MUTAG batches (4 337 graphs, 131 k nodes) the way predict() plans them: every batch whose state fits the LDS of one
    CU in ONE launch, one workgroup per batch (kernel_state_lds.hpp) - all of them at 16-wide states, all but the largest at
    32-wide - and the rest spread over several CUs each.  Iteration counts, states and outputs against the batch-by-batch calls,
    two batches against the oracle."""
    seq = MultiGraphSequencer(mutag_graphs, 'g', 'average', 32, shuffle=False)
    for d, K_it in ((32, 30), (0, 5)):
        ns, no = starter_nets('g', d, scale=0.22)
        model = GNNgraphBased(ns, no, d, K_it, 0.01)
        rng = np.random.default_rng(1)
        plan = model._group_plan(seq, torch.device('cuda', 0))
        assert plan is not None and sorted(b for bs in plan for b in bs) == list(range(len(seq)))
        assert len(plan[0]) >= (len(seq) if d == 0 else 100), [len(bs) for bs in plan]
        s0s = [rng.normal(0, 0.1, (seq[i][0][0].shape[0], d)).astype(np.float32) if d else None for i in range(len(seq))]
        ks_single = {}
        for li, bs in enumerate(plan):
            if len(bs) < 2: continue
            x, begin = seq.merged_batches(bs)
            kw = {}
            if bs.parts:      # batches above one CU's LDS: cut along graph boundaries, the parts share the convergence flag (group sets)
                fine, kw['group_sets'] = bs.groups_and_sets({b: begin[i + 1] - begin[i] for i, b in enumerate(bs)})
                assert len(fine) > len(begin)
            else: fine = begin
            k, st, o = model.Loop(*model.process_inputs(x), state0=dev(np.concatenate([s0s[b] for b in bs])) if d else None, groups=fine, **kw)
            assert _last_kernel().startswith('k_state_lds' if bs.resident else 'k_state_small'), (li, _last_kernel())
            assert k.shape == (len(bs),)
            r0 = 0
            for j, b in enumerate(bs):
                kb, stb, ob = model.Loop(*model.process_inputs(seq[b][0]), state0=None if s0s[b] is None else dev(s0s[b]))
                ks_single[b] = float(kb)
                assert float(k[j]) == float(kb), (b, float(k[j]), float(kb))
                assert rel_err(st[begin[j]:begin[j + 1]].cpu().numpy(), stb.cpu().numpy()) <= TOL, b
                assert rel_err(o[r0:r0 + ob.shape[0]].cpu().numpy(), ob.cpu().numpy()) <= TOL, b
                r0 += ob.shape[0]
            if li == 0:
                for j in (0, len(bs) - 1):
                    k64, st64, o64 = oracle_loop(model, seq[bs[j]][0], s0s[bs[j]], np.float64)
                    assert float(k[j]) == float(k64) and rel_err(st[begin[j]:begin[j + 1]].cpu().numpy(), st64) <= TOL
        if d == 32:           # every batch is resident now: the 14 batches above 1 123 nodes as two coupled groups each, ONE launch
            assert len(plan) == 1 and plan[0].resident and len(plan[0].parts) >= 10, [(len(bs), bs.resident, len(bs.parts)) for bs in plan]


def test_predict_and_evaluate_group_batches(mutag_graphs):
    """predict() / evaluate() with grouped launches against the batch-by-batch walk: same numbers (state_vect_dim = 0 keeps
    the forward deterministic; 20 label columns put the state on the 32-wide kernels the groups need)."""
    rng = np.random.default_rng(4)
    gl = []
    for i in range(150):
        n = int(rng.integers(5, 40))
        g = _random_graph(rng, n, 3 * n, 20, 3, focus='g')
        gl.append(GraphObject(nodes=g.nodes, arcs=g.arcs, targets=np.eye(2)[rng.integers(0, 2, 1)], focus='g'))
    seq = MultiGraphSequencer(gl, 'g', 'average', 16, shuffle=False)
    ns, no = starter_nets('g', 0, L=20, A=3, scale=0.3)
    model = GNNgraphBased(ns, no, 0, 25, 0.005)
    model.compile(optimizer='adam', loss='categorical_crossentropy', metrics=['accuracy'])
    dev_ = torch.device('cuda', 0)
    plan = model._group_plan(seq, dev_)
    assert plan is not None and len(plan[0]) > 1, plan
    model.group_batches = True
    p1 = model.predict(seq); e1 = model.evaluate(seq, return_dict=True)
    model.group_batches = False
    p0 = model.predict(seq); e0 = model.evaluate(seq, return_dict=True)
    assert p1.shape == (150, 2) and rel_err(p1, p0) <= TOL             # (grouped: one CU per batch, other summation orders)
    assert abs(e1['loss'] - e0['loss']) <= 1e-5 and abs(e1['accuracy'] - e0['accuracy']) <= 1e-6
    # 'normalized' divides by the merged graph's arc count: never grouped
    seqn = MultiGraphSequencer(gl, 'g', 'normalized', 16, shuffle=False)
    assert seqn.merged_batches(0, 2) is None and model._group_plan(seqn, dev_) is None


def test_groups_are_refused_where_unsupported(mutag_graphs):
    seq = MultiGraphSequencer(mutag_graphs[:64], 'g', 'average', 32, shuffle=False)
    ns, no = starter_nets('g', 100)                            # state width 100: the wide kernel, no whole-loop launch
    model = GNNgraphBased(ns, no, 100, 5, 0.01)
    assert model._group_plan(seq, torch.device('cuda', 0)) is None
    x, begin = seq.merged_batches(0, 2)
    with pytest.raises(RuntimeError, match='groups'):
        model.Loop(*model.process_inputs(x), groups=begin)
    assert np.array_equal(model.predict(seq).shape, (64, 2))


def test_starter_configuration_runs_grouped(mutag_graphs):
    """The reference's own default (starter.py: state_vect_dim = 0, max_iteration = 5, threshold 0.01, batches of 32): the state
    is the 14 label columns (16-wide kernels); predict() / evaluate() group the batches and must reproduce the batch-by-batch
    walk exactly."""
    gl = [g.copy() for g in mutag_graphs[:32 * 20]]
    seq = MultiGraphSequencer(gl, 'g', 'average', 32, shuffle=False)
    ns, no = starter_nets('g', 0)
    model = GNNgraphBased(ns, no, 0, 5, 0.01)
    model.compile(optimizer='adam', loss='categorical_crossentropy', metrics=['accuracy'])
    plan = model._group_plan(seq, torch.device('cuda', 0))
    assert plan is not None and len(plan) < len(seq), plan
    p1 = model.predict(seq); e1 = model.evaluate(seq, return_dict=True)
    assert _last_kernel().startswith('k_state_lds<16'), _last_kernel()
    model.group_batches = False
    p0 = model.predict(seq); e0 = model.evaluate(seq, return_dict=True)
    assert rel_err(p1, p0) <= TOL and abs(e1['loss'] - e0['loss']) <= 1e-5 and abs(e1['accuracy'] - e0['accuracy']) <= 1e-6


@pytest.mark.parametrize('d,state_dim0,mode', [(64, False, 'average'), (32, False, 'sum'), (20, True, 'average')])
def test_constant_inputs_on_the_matrix_cores_variant(d, state_dim0, mode):
    """From ~200 k nodes the wave-specialised kernel no longer reads the per-node constant C (4 H bytes per node and
    iteration) but the node's constant inputs [labels | aggregated labels | aggregated arcs | 1] (128 bytes) and multiplies
    them with their folded weights on the matrix cores (k_state_fused4<.., XC = true>): against the oracle, the un-fused
    kernels and the C form of the same kernel (phase-alternating kernel pinned)."""
    rng = np.random.default_rng(d)
    N = 210000
    g = er_graph(N, 5 * N, seed=9, aggregation_mode=mode, dim_node_label=20 if state_dim0 else 14)
    sd = 0 if state_dim0 else d
    ns, no = starter_nets('n', sd, L=20 if state_dim0 else 14, scale=0.3)
    model = GNNnodeBased(ns, no, sd, 4, 0.0)
    s0 = None if state_dim0 else rng.normal(0, 0.1, (N, d)).astype(np.float32)
    x = MultiGraphSequencer([g], 'n', mode, 1, shuffle=False)[0][0]
    k64, st64, o64 = oracle_loop(model, x, s0, np.float64, exact_order=False)
    inputs = model.process_inputs(x)
    for flags in (0, nat.FLAG_UNFUSED, nat.FLAG_FUSED_GEN2):
        model.native_flags = flags
        k, st, o = model.Loop(*inputs, state0=None if s0 is None else torch.from_numpy(s0).cuda())
        assert float(k) == float(k64), (flags, float(k), k64)
        assert rel_err(st.cpu().numpy(), st64) <= TOL and rel_err(o.cpu().numpy(), o64) <= TOL, (flags, rel_err(st.cpu().numpy(), st64))
        if flags == 0: assert _last_kernel().endswith(',true>') and _last_kernel().startswith('k_state_fused4'), _last_kernel()


def test_constant_inputs_variant_at_every_size(mutag_graphs, monkeypatch):
    """GNN_XC_MIN_NODES=0 (read at every call) puts every homogeneous one-layer model that reaches the wave-specialised kernel on its XC
    form (the constant inputs multiplied on the matrix cores instead of the per-node constant C read) - small graphs pinned to that
    kernel, shards, the overlapped shard iteration (INIT + XC), hub rows - on one representative configuration per kernel instance
    (padded widths 16 / 32 / 64, with and without the own-range split) of the parity tests that cover those paths."""
    monkeypatch.setenv('GNN_XC_MIN_NODES', '0')
    test_c2_mutag_batch_state_dim_32_k_pinned(mutag_graphs, 'g')
    test_c2_mutag_batch_state_dim_32_k_pinned(mutag_graphs, 'a')
    for d in (14, 33, 64): test_odd_state_widths(mutag_graphs, d)
    test_other_aggregation_modes(mutag_graphs, 'sum')
    test_sharded_native_kernels_match_oracle(3, 0.02)
    test_sharded_overlap_split_matches_oracle(2, 'average', 0.0)
    test_sharded_overlap_split_matches_oracle(8, 'sum', 0.0)
    test_hub_rows_use_the_segment_prepass('average', False, 64)
    test_hub_rows_use_the_segment_prepass('average', False, 32)
    assert ',true>' in c3_case('average').replace(' ', ''), 'C3 did not run the XC form'
