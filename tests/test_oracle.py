"""Pins for the oracle (oracle/gnn_oracle.py). The reference holds no golden vectors for the Loop and TensorFlow is not
available (PARITY UNPINNED), so the restatement is pinned by
  (1) hand-computable known answers, (2) the `condition` truth table, (3) invariances of the algorithm,
  (4) an independent torch-CPU implementation of the Keras op semantics (batch_norm eps 1e-3, selu, softmax, AᵀX)."""
import numpy as np
import pytest
import torch

from oracle import gnn_oracle as O
from oracle.harness import rel_err


def coo_triple(rows, cols, vals, shape):
    idx, v = O.sparse_reorder(np.stack([rows, cols], 1), np.asarray(vals, dtype=np.float64))
    return idx, v, np.array(shape)


def toy_graph(mode='sum'):
    # 0->1, 1->2, 2->0, 0->2, 3 isolated
    src, dst = np.array([0, 0, 1, 2]), np.array([1, 2, 2, 0])
    N, E = 4, 4
    indeg = np.bincount(dst, minlength=N)
    w = np.ones(E) if mode == 'sum' else 1.0 / indeg[dst]
    adjacency = coo_triple(src, dst, w, (N, N))
    arcnode = coo_triple(np.arange(E), dst, w, (E, N))
    nodegraph = coo_triple(np.arange(N), np.zeros(N, int), np.full(N, 1 / N), (N, 1))
    nodes = np.array([[1., 0.], [0., 1.], [1., 1.], [2., 0.]])
    arcs = np.concatenate([np.stack([src, dst], 1), np.array([[.5], [1.], [2.], [3.]])], axis=1).astype(float)
    return nodes, arcs, adjacency, arcnode, nodegraph


def dense_net(in_dim, out_dim, act='linear', bn=False, W=None, b=None):
    w = []
    if bn: w += [np.ones(in_dim), np.zeros(in_dim), np.zeros(in_dim), np.ones(in_dim)]
    w += [np.zeros((in_dim, out_dim)) if W is None else W, np.zeros(out_dim) if b is None else b]
    return {'batch_normalization': bn, 'activations': [act]}, w


def test_sparse_matmul_adjoint_known_answer():
    nodes, arcs, adjacency, arcnode, _ = toy_graph('sum')
    agg = O.sparse_dense_matmul_adjoint(*adjacency, nodes, np.float64)
    assert np.array_equal(agg, np.array([[1., 1.], [1., 0.], [1., 1.], [0., 0.]]))      # 2->0 ; 0->1 ; 0,1->2 ; none
    agg_arcs = O.sparse_dense_matmul_adjoint(*arcnode, arcs[:, 2:], np.float64)
    assert np.array_equal(agg_arcs, np.array([[3.], [.5], [3.], [0.]]))
    fast = O.sparse_dense_matmul_adjoint(*adjacency, nodes, np.float64, exact_order=False)
    assert np.allclose(agg, fast)


def test_zero_weights_converge_at_k2():
    """W = 0 => state_1 = act(b) for every node, state_2 == state_1 => the loop stops at k = 2 for any threshold."""
    nodes, arcs, adjacency, arcnode, nodegraph = toy_graph()
    d = 3
    b = np.array([0.3, -0.2, 1.5])
    ns = dense_net(2 * d + 2 * 2 + 1, d, 'selu', b=b)
    no = dense_net(d + 2, 2, 'linear', W=np.ones((d + 2, 2)))
    for thr in [0.0, 0.5]:
        k, state, out = O.loop(nodes, arcs, [2], np.ones(4, bool), np.ones(4, bool), adjacency, arcnode, nodegraph,
                               net_state=ns, net_output=no, state_vect_dim=d, max_iteration=10, state_threshold=thr,
                               state0=np.full((4, d), 0.1), dtype=np.float64)
        assert k == 2
        expect = O.activation('selu', b, np.float64)
        assert np.allclose(state, np.tile(expect, (4, 1)))
        assert np.allclose(out[:, 0], expect.sum() + nodes.sum(1))


def test_linear_fixed_point_closed_form():
    """d = 1, linear activation, 'average': s_{t+1} = a s + c A_avg^T s + bias  — compare with the dense recurrence."""
    nodes, arcs, adjacency, arcnode, nodegraph = toy_graph('average')
    a, c, bias = 0.3, 0.4, 0.2
    # input layout [state(1) | nodes(2) | agg_state(1) | agg_nodes(2) | agg_arcs(1)]  (GNN.py:222-231)
    W = np.zeros((7, 1)); W[0, 0] = a; W[3, 0] = c
    ns = dense_net(7, 1, 'linear', W=W, b=np.array([bias]))
    no = dense_net(3, 1, 'linear', W=np.array([[1.], [0.], [0.]]))
    A = np.zeros((4, 4))
    for (i, j), v in zip(adjacency[0], adjacency[1]): A[i, j] = v
    s = np.full((4, 1), 0.1)
    k, state, out = O.loop(nodes, arcs, [2], np.ones(4, bool), np.ones(4, bool), adjacency, arcnode, nodegraph,
                           net_state=ns, net_output=no, state_vect_dim=1, max_iteration=7, state_threshold=0.0,
                           state0=s, dtype=np.float64)
    for _ in range(7): s = a * s + c * (A.T @ s) + bias
    assert k == 7 and np.allclose(state, s) and np.allclose(out, s)


def test_condition_truth_table():
    f = lambda k, s, so, mx, thr: O.condition(k, np.array(s, float), np.array(so, float), mx, thr, np.float64)
    assert f(0, [[1., 1.]], [[1., 1.]], 5, 0.0) is False            # zero distance, strict '>'  => converged
    assert f(0, [[0., 0.]], [[0., 0.]], 5, 0.5) is False            # zero norm & zero distance => converged
    assert f(0, [[1., 0.]], [[0., 0.]], 5, 10.) is True             # zero norm, positive distance
    assert f(0, [[1.1, 1.]], [[1., 1.]], 5, 0.05) is True           # 0.1 > 0.05*sqrt(2)
    assert f(0, [[1.1, 1.]], [[1., 1.]], 5, 0.1) is False           # 0.1 > 0.1414 false
    assert f(5, [[9., 9.]], [[1., 1.]], 5, 0.0) is False            # k < max_iteration fails
    assert f(0, [[1., 1.], [5., 5.]], [[1., 1.], [1., 1.]], 5, 0.1) is True   # reduce_any over nodes


def test_max_iteration_zero_returns_state0():
    nodes, arcs, adjacency, arcnode, nodegraph = toy_graph()
    d = 2
    rng = np.random.default_rng(0)
    ns = dense_net(2 * d + 5, d, 'tanh', W=rng.normal(size=(2 * d + 5, d)))
    no = dense_net(d + 2, 2, 'softmax', W=rng.normal(size=(d + 2, 2)))
    s0 = rng.normal(size=(4, d))
    k, state, out = O.loop(nodes, arcs, [2], np.ones(4, bool), np.ones(4, bool), adjacency, arcnode, nodegraph,
                           net_state=ns, net_output=no, state_vect_dim=d, max_iteration=0, state_threshold=0.0,
                           state0=s0, dtype=np.float64)
    assert k == 0 and np.array_equal(state, s0) and np.allclose(out.sum(1), 1.0)


def _random_graph(rng, n, e, A=2):
    pairs = set()
    while len(pairs) < e:
        i, j = rng.integers(0, n, 2)
        if i != j: pairs.add((int(i), int(j)))
    ids = np.array(sorted(pairs))
    arcs = np.concatenate([ids, rng.normal(size=(e, A))], axis=1)
    return ids, arcs


def _operands(ids, n, mode='average'):
    e = len(ids)
    indeg = np.bincount(ids[:, 1], minlength=n)
    w = np.ones(e) if mode == 'sum' else 1.0 / indeg[ids[:, 1]]
    return coo_triple(ids[:, 0], ids[:, 1], w, (n, n)), coo_triple(np.arange(e), ids[:, 1], w, (e, n))


def _nets(rng, L, A, d, T, bn=True):
    in_s, in_o = 2 * d + 2 * L + A, d + L
    def net(i, o, act):
        w = []
        if bn: w += [rng.uniform(.5, 1.5, i), rng.normal(size=i) * .1, rng.normal(size=i) * .1, rng.uniform(.5, 1.5, i)]
        w += [rng.normal(size=(i, o)) / np.sqrt(i), rng.normal(size=o) * .1]
        return {'batch_normalization': bn, 'activations': [act]}, w
    return net(in_s, d, 'selu'), net(in_o, T, 'softmax')


def test_permutation_equivariance_and_merge_invariance():
    rng = np.random.default_rng(3)
    n, e, L, A, d, T = 30, 90, 3, 2, 4, 2
    ids, arcs = _random_graph(rng, n, e, A)
    nodes = rng.normal(size=(n, L)); s0 = rng.normal(size=(n, d)) * .1
    ns, no = _nets(rng, L, A, d, T)
    adj, an = _operands(ids, n)
    ng = coo_triple(np.arange(n), np.zeros(n, int), np.full(n, 1 / n), (n, 1))
    kw = dict(net_state=ns, net_output=no, state_vect_dim=d, max_iteration=6, state_threshold=0.0, dtype=np.float64)
    k, st, out = O.loop(nodes, arcs, [L], np.ones(n, bool), np.ones(n, bool), adj, an, ng, state0=s0, **kw)
    # relabel nodes with a permutation: states / outputs are permuted the same way
    perm = rng.permutation(n); inv = np.argsort(perm)                  # new id of old node i is inv[i]
    ids2 = inv[ids]; order = np.lexsort((ids2[:, 1], ids2[:, 0]))
    arcs2 = np.concatenate([ids2, arcs[:, 2:]], axis=1)[order]
    adj2, an2 = _operands(ids2[order], n)
    k2, st2, out2 = O.loop(nodes[perm], arcs2, [L], np.ones(n, bool), np.ones(n, bool), adj2, an2, ng,
                           state0=s0[perm], **kw)
    assert k2 == k and np.allclose(st2, st[perm], atol=1e-12) and np.allclose(out2, out[perm], atol=1e-12)
    # block-diagonal merge of two copies == running each separately ('average' is batching invariant)
    idsm = np.concatenate([ids, ids + n]); arcsm = np.concatenate([arcs, arcs]); arcsm[:, :2] = idsm
    adjm, anm = _operands(idsm, 2 * n)
    ngm = coo_triple(np.arange(2 * n), np.repeat([0, 1], n), np.full(2 * n, 1 / n), (2 * n, 2))
    km, stm, outm = O.loop(np.concatenate([nodes, nodes]), arcsm, [L], np.ones(2 * n, bool), np.ones(2 * n, bool), adjm,
                           anm, ngm, state0=np.concatenate([s0, s0]), **kw)
    assert km == k and np.allclose(stm[:n], st) and np.allclose(stm[n:], st) and np.allclose(outm[:n], out)


@pytest.mark.parametrize('training', [False, True])
def test_mlp_apply_against_independent_torch(training):
    """Keras op semantics restated in the oracle vs torch's own batch_norm / linear / selu / softmax."""
    rng = np.random.default_rng(5)
    x = rng.normal(size=(50, 7))
    gamma, beta, mean, var = rng.uniform(.5, 1.5, 7), rng.normal(size=7), rng.normal(size=7), rng.uniform(.5, 2, 7)
    W1, b1, W2, b2 = rng.normal(size=(7, 5)), rng.normal(size=5), rng.normal(size=(5, 3)), rng.normal(size=3)
    spec = {'batch_normalization': True, 'activations': ['selu', 'softmax']}
    got = O.mlp_apply(spec, [gamma, beta, mean, var, W1, b1, W2, b2], x, training, np.float64)
    t = lambda a: torch.tensor(a, dtype=torch.float64)
    h = torch.nn.functional.batch_norm(t(x), t(mean).clone(), t(var).clone(), t(gamma), t(beta), training=training,
                                       momentum=0.01, eps=1e-3)
    h = torch.selu(h @ t(W1) + t(b1))
    ref = torch.softmax(h @ t(W2) + t(b2), dim=-1).numpy()
    assert rel_err(got, ref) < 1e-12
    got32 = O.mlp_apply(spec, [gamma, beta, mean, var, W1, b1, W2, b2], x, training, np.float32)
    assert got32.dtype == np.float32 and rel_err(got32, ref) < 1e-5


def test_full_loop_against_independent_torch():
    """The whole op sequence (adjoint SpMM via torch.sparse, concat, BN, Dense, selu, predicate) re-implemented with
    torch CPU ops, independent of the numpy restatement."""
    rng = np.random.default_rng(9)
    n, e, L, A, d, T = 40, 120, 3, 2, 5, 2
    ids, arcs = _random_graph(rng, n, e, A)
    nodes = rng.normal(size=(n, L)); s0 = rng.normal(size=(n, d)) * .1
    ns, no = _nets(rng, L, A, d, T)
    adj, an = _operands(ids, n)
    ng = coo_triple(np.arange(n), np.zeros(n, int), np.full(n, 1 / n), (n, 1))
    thr, max_it = 1e-3, 30
    k, st, out = O.loop(nodes, arcs, [L], np.ones(n, bool), np.ones(n, bool), adj, an, ng, net_state=ns, net_output=no,
                        state_vect_dim=d, max_iteration=max_it, state_threshold=thr, state0=s0, dtype=np.float64,
                        focus='g')
    t = lambda a: torch.tensor(np.asarray(a), dtype=torch.float64)
    At = torch.sparse_coo_tensor(torch.tensor(adj[0].T[[1, 0]]), t(adj[1]), (n, n))          # A^T
    ANt = torch.sparse_coo_tensor(torch.tensor(an[0].T[[1, 0]]), t(an[1]), (n, e))
    def mlp(net, x):
        g, b, m, v, W, bb = [t(w) for w in net[1]]
        x = torch.nn.functional.batch_norm(x, m, v, g, b, training=False, eps=1e-3)
        x = x @ W + bb
        return torch.selu(x) if net[0]['activations'][0] == 'selu' else torch.softmax(x, -1)
    X, lab = t(nodes), t(arcs[:, 2:])
    agg_arcs, agg_nodes = torch.sparse.mm(ANt, lab), torch.sparse.mm(At, X)
    s, so, kk = t(s0), torch.ones(n, d, dtype=torch.float64), 0
    while bool(((s - so).pow(2).sum(1).sqrt() > thr * so.pow(2).sum(1).sqrt()).any()) and kk < max_it:
        s, so, kk = mlp(ns, torch.cat([s, X, torch.sparse.mm(At, s), agg_nodes, agg_arcs], 1)), s, kk + 1
    o = mlp(no, torch.cat([s, X], 1)).mean(0, keepdim=True)
    assert kk == k and 0 < k < max_it
    assert rel_err(st, s.numpy()) < 1e-10 and rel_err(out, o.numpy()) < 1e-10


def test_composite_equals_homogeneous_when_one_type():
    """T = 1 composite with d_0 = L reduces to the homogeneous input [labels|state|agg_state|agg_labels|agg_arcs] up to
    the column order of the first layer, and its filters drop the label concat (CompositeGNN.py:237-239)."""
    rng = np.random.default_rng(11)
    n, e, L, A, d, T = 25, 70, 3, 2, 4, 2
    ids, arcs = _random_graph(rng, n, e, A)
    nodes = rng.normal(size=(n, L)); s0 = rng.normal(size=(n, d)) * .1
    adj, an = _operands(ids, n)
    ng = coo_triple(np.arange(n), np.zeros(n, int), np.full(n, 1 / n), (n, 1))
    Wh = rng.normal(size=(2 * d + 2 * L + A, d)) / 4; bh = rng.normal(size=d) * .1
    # homogeneous rows: [state d | nodes L | agg_state d | agg_nodes L | agg_arcs A]; composite: [nodes L | state d | agg_state d | agg_nodes L | agg_arcs A]
    perm = np.r_[d:d + L, 0:d, d + L:2 * d + L, 2 * d + L:2 * d + 2 * L + A]
    ns_h = ({'batch_normalization': False, 'activations': ['tanh']}, [Wh, bh])
    ns_c = ({'batch_normalization': False, 'activations': ['tanh']}, [Wh[perm], bh])
    Wo = rng.normal(size=(d, T))
    no_c = ({'batch_normalization': False, 'activations': ['linear']}, [Wo, np.zeros(T)])
    no_h = ({'batch_normalization': False, 'activations': ['linear']}, [np.concatenate([Wo, np.zeros((L, T))]), np.zeros(T)])
    kw = dict(state_vect_dim=d, max_iteration=5, state_threshold=0.0, state0=s0, dtype=np.float64)
    kh, sh, oh = O.loop(nodes, arcs, [L], np.ones(n, bool), np.ones(n, bool), adj, an, ng, net_state=ns_h,
                        net_output=no_h, **kw)
    kc, sc, oc = O.composite_loop(nodes, arcs, [L], np.ones((1, n), bool), np.ones(n, bool), np.ones(n, bool), [adj],
                                  adj, an, ng, net_state=[ns_c], net_output=no_c, **kw)
    assert kh == kc and np.allclose(sh, sc) and np.allclose(oh, oc)


def test_train_step_oracle_with_checkpointed_iterations_is_the_same_oracle():
    """oracle/torch_train.py `checkpoint_iterations=True` (what lets the float64 autograd oracle of a million-node train step fit a host:
    tests/test_gpu_round5.py) recomputes each iteration in the backward pass instead of keeping its intermediates - the same float64
    operations in the same order: loss, every gradient, the moving statistics (one update per call, not two) and the kink counts are
    bit-identical to the plain run; and the kink counter sees a pre-activation planted at a relu kink."""
    from gnnkeras_amd.synth import er_graph
    from gnnkeras_amd.Models.MLP import MLP, get_inout_dims
    from oracle import torch_train
    N, d = 1500, 8
    g = er_graph(N, 6 * N, seed=3, aggregation_mode='average')
    rng = np.random.default_rng(0)
    inp, lay = get_inout_dims('state', 14, 3, 2, 'n', d)
    ns = MLP(inp[0], lay, 'relu', 'lecun_normal', 'lecun_normal', rng=0, batch_normalization=True, device='cpu')
    inp, lay = get_inout_dims('output', 14, 3, 2, 'n', d)
    no = MLP(inp[0], lay, 'softmax', 'glorot_normal', 'glorot_normal', rng=1, batch_normalization=True, device='cpu')
    trip = lambda m: (np.stack([m.tocoo().row, m.tocoo().col], 1), m.tocoo().data, m.shape)
    s0 = np.abs(rng.normal(0, 0.1, (N, d))).astype(np.float32)
    kw = dict(net_state=ns.spec(), net_output=no.spec(), state_vect_dim=d, max_iteration=4, state_threshold=0.0, focus='n', state0=s0,
              y=g.targets, sample_weight=None, loss='categorical_crossentropy')
    args = (g.nodes, g.arcs, trip(g.Adjacency), trip(g.ArcNode), None, np.ones(N, bool))
    a = torch_train.train_step(*args, **kw)
    b = torch_train.train_step(*args, checkpoint_iterations=True, **kw)
    assert a['k'] == b['k'] == 4 and a['loss'] == b['loss']
    for x_, y_ in zip(a['grads_state'] + a['grads_output'] + list(a['moving_state']) + list(a['moving_output']),
                      b['grads_state'] + b['grads_output'] + list(b['moving_state']) + list(b['moving_output'])):
        assert np.array_equal(x_, y_)
    assert all(np.array_equal(p, q) for p, q in zip(a['kinks_state'], b['kinks_state']))
    # a bias that puts unit 0's pre-activation of row 0 exactly on the kink in the first call: counted once
    spec = ns.spec()
    w = [np.array(t, dtype=np.float64) for t in spec[1]]
    net = torch_train.Net(spec[0], w)
    x0 = torch.tensor(rng.normal(0, 1, (5, w[-2].shape[0])), dtype=torch.float64)
    z = ((x0 - x0.mean(0)) / torch.sqrt(((x0 - x0.mean(0)) ** 2).mean(0) + 1e-3) * net.gamma + net.beta) @ net.W[0] + net.b[0]
    with torch.no_grad(): net.b[0][0] -= z[0, 0]
    net(x0)
    assert net.kinks[0][0] >= 1


def test_composite_train_step_oracle_with_checkpointed_iterations_is_the_same_oracle():
    """`composite_train_step(checkpoint_iterations=True)` - what lets the float64 oracle of a C5-size heterogeneous train step
    (tests/test_gpu_round6.py) fit a host - is bit for bit the plain run: loss, every network's gradients, one moving-average update per
    call and network, the kink counts."""
    from gnnkeras_amd.synth import er_composite_graph
    from gnnkeras_amd.Models.MLP import MLP, get_inout_dims
    from oracle import torch_train
    N, d, dims = 900, 8, (5, 3, 2)
    g = er_composite_graph(N, 5 * N, dim_node_label=dims, seed=5, aggregation_mode='average')
    rng = np.random.default_rng(0)
    inp, lay = get_inout_dims('state', list(dims), 3, 2, 'n', d)
    ns = [MLP(i, lay, 'relu', 'lecun_normal', 'lecun_normal', rng=t, batch_normalization=True, device='cpu') for t, i in enumerate(inp)]
    inp, lay = get_inout_dims('output', list(dims), 3, 2, 'n', d)
    no = MLP(inp[0], lay, 'softmax', 'glorot_normal', 'glorot_normal', rng=9, batch_normalization=True, device='cpu')
    trip = lambda m: (np.stack([m.tocoo().row, m.tocoo().col], 1), m.tocoo().data, m.shape)
    s0 = np.abs(rng.normal(0, 0.1, (N, d))).astype(np.float32)
    kw = dict(net_state=[n.spec() for n in ns], net_output=no.spec(), state_vect_dim=d, max_iteration=3, state_threshold=0.0, focus='n',
              state0=s0, y=g.targets, sample_weight=None, loss='categorical_crossentropy')
    args = (g.nodes, g.arcs, list(dims), g.type_mask.T, [trip(c) for c in g.CompositeAdjacencies], trip(g.Adjacency), trip(g.ArcNode), None,
            np.ones(N, bool))
    a = torch_train.composite_train_step(*args, **kw)
    b = torch_train.composite_train_step(*args, checkpoint_iterations=True, **kw)
    assert a['k'] == b['k'] == 3 and a['loss'] == b['loss']
    flat = lambda r: [t for gs in r['grads_state'] for t in gs] + r['grads_output'] + [t for mv in r['moving_state'] for t in mv] + list(r['moving_output'])
    for x_, y_ in zip(flat(a), flat(b)): assert np.array_equal(x_, y_)
    for ka, kb in zip(a['kinks_state'], b['kinks_state']):
        assert all(np.array_equal(p, q) for p, q in zip(ka, kb))
