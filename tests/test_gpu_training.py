"""train_step / fit on the device vs the torch-autograd restatement of the reference's train_step
(oracle/torch_train.py, float64).  Gradients are checked per trainable variable against PER-TENSOR bars (`BARS`: BatchNormalization
gamma / beta, Dense kernel / bias - max|a-b| relative to the tensor's own largest entry, or to the largest gradient entry of the step
for tensors that are tiny themselves), widened to `KINK_BAR` only for the output units the float64 oracle saw within 1e-6 of an
activation kink (relu / selu: act' jumps there, and a float32 pre-activation may sit on the other side - `grad_rows`)."""
import numpy as np
import pytest
import torch

from gnnkeras_amd import GraphObject
from gnnkeras_amd.Models.MLP import MLP, get_inout_dims
from gnnkeras_amd.Models.GNN import GNNnodeBased, GNNarcBased, GNNgraphBased
from gnnkeras_amd.Models.training import Adam, SGD
from gnnkeras_amd.Sequencers.GraphSequencers import MultiGraphSequencer
from oracle import torch_train
from oracle.harness import rel_err, _np, _triple

pytestmark = pytest.mark.gpu
CLS = {'n': GNNnodeBased, 'a': GNNarcBased, 'g': GNNgraphBased}
# Per-tensor bars of train_step's gradients (round 5).  Measured on MI355X over the 184 comparisons of the training tests in which the
# float64 oracle saw NO pre-activation near an activation kink (profiles/r05_train_tensor_errors.txt: 18 .. 1 000 000 rows, both
# orchestrations, all three foci, fuzzed configurations): worst gamma 5e-7, beta 1.9e-6, kernel 4.9e-6, bias 4.3e-6 - the bars keep 4 x.
BARS = {'gamma': 2e-5, 'beta': 2e-5, 'kernel': 2e-5, 'bias': 2e-5}
KINK_UNIT = 16.0               # what ONE pre-activation on the other side of a relu / selu kink may move a network's gradient entries, in units of
                               # (largest gradient entry of the network) / (rows of the network call).  Measured: 0.12 (125 rows), 1.3 (5 297
                               # rows), 11 (40 000 rows, profiles/r04_notes.txt 7); n such elements add up like sqrt(n) (independent signs)
Y_PRED_BAR = 1e-5              # training-mode predictions (worst measured 3.3e-6)


def tensor_kinds(bn, n_layers):
    return (['gamma', 'beta'] if bn else []) + [k for l in range(n_layers) for k in (f'kernel{l}', f'bias{l}')]


def grad_rows(net_name, got, ref, bn, kinks, scale, M):
    """One row per trainable tensor: its error relative to its own largest entry and to the network's largest gradient entry (`scale`),
    the bar that applies and whether it holds.  An entry passes when |got - ref| <= bar x max(own largest, scale) + kink allowance.
    `kinks`: per Dense layer, per output unit, the float64 oracle's count of pre-activations within 1e-6 of a relu / selu kink (None: no
    allowance).  A float32 implementation may put such an element on the other side of zero, where act' differs by the jump: that is one
    row's contribution with the wrong factor - not an arithmetic error of the kernels.  It moves its unit's kernel column and bias entry
    directly and, through dZ . W^T and the loop's earlier iterations, every other entry of the network a little: the allowance is
    KINK_UNIT x sqrt(n) x scale / M for n such elements in the network's calls of this step - and ZERO when the oracle saw none, which is
    the case in four comparisons out of five."""
    rows = []
    kinds = tensor_kinds(bn, (len(ref) - (2 if bn else 0)) // 2)
    total_kinks = 0 if kinks is None else int(sum(int(np.sum(kq)) for kq in kinks))
    allowance = KINK_UNIT * float(np.sqrt(total_kinks)) * scale / max(M, 1)
    for kind, g, r in zip(kinds, got, ref):
        g = g.detach().cpu().numpy() if hasattr(g, 'detach') else np.asarray(g)
        diff = np.abs(g - r)
        own = max(float(np.max(np.abs(r))), 1e-30)
        bar = BARS[kind.rstrip('0123456789')]
        ok = bool(np.all(diff <= bar * max(own, scale) + allowance))
        rows.append(dict(net=net_name, tensor=kind, err_own=float(np.max(diff)) / own, err_scale=float(np.max(diff)) / max(scale, 1e-30),
                         kink_elements=total_kinks, kink_allowance_rel_scale=allowance / max(scale, 1e-30), bar=bar, ok=ok))
    return rows


def log_rows(tag, rows, **extra):
    """Append the per-tensor errors to $GNN_TEST_ERRLOG (one JSON line per comparison) - how BARS were chosen and are re-checked."""
    import json, os
    path = os.environ.get('GNN_TEST_ERRLOG')
    if path:
        with open(path, 'a') as f: f.write(json.dumps(dict(tag=tag, rows=rows, **extra)) + '\n')


def check_network_grads(tag, networks, **extra):
    """`networks`: [(name, _NetGrads of the device step, the oracle's gradients of that network, its kink counts or None, rows of its calls)].
    Every tensor of every network against `BARS` relative to max(its own largest entry, the largest gradient entry of ITS network) - the
    rule of check_step, for models with more than two networks (heterogeneous models: one state network per node type; layered models: a
    state and an output network per layer)."""
    import os
    rows = []
    for name, ng_, ref, kinks, M in networks:
        got = ng_.gradients()
        assert len(got) == len(ref), (name, len(got), len(ref))
        scale = max(max(float(np.max(np.abs(t_))) for t_ in ref), 1e-30)
        rows += grad_rows(name, got, ref, ng_.bn, kinks, scale, M)
    log_rows(tag or os.environ.get('PYTEST_CURRENT_TEST', ''), rows, **extra)
    bad = [r_ for r_ in rows if not r_['ok']]
    assert not bad, bad
    return rows


def split_by_network(holders, flat_ref):
    """The oracle's flat gradient list of several networks cut at the networks' tensor counts."""
    out, pos = [], 0
    for h in holders:
        n = len(h.gradients())
        out.append(flat_ref[pos:pos + n]); pos += n
    assert pos == len(flat_ref), (pos, len(flat_ref))
    return out


def nets(focus, d, bn, hidden_state=None, hidden_out=None, act='selu', out_act='softmax', scale=0.5):
    inp, lay = get_inout_dims('state', 14, 3, 2, focus, d, hidden_units=hidden_state)
    ns = MLP(inp[0], lay, act, 'lecun_normal', 'lecun_normal', rng=0, batch_normalization=bn)
    ns.set_weights([a * scale if a.ndim == 2 else a for a in ns.get_weights()])
    inp, lay = get_inout_dims('output', 14, 3, 2, focus, d, hidden_units=hidden_out)
    no = MLP(inp[0], lay, ['tanh'] * (len(lay) - 1) + [out_act], 'glorot_normal', 'glorot_normal', rng=1,
             batch_normalization=bn)
    if bn:      # non-trivial gamma / beta so their gradients are exercised
        rng = np.random.default_rng(3)
        for n in (ns, no):
            w = n.get_weights()
            w[0] = rng.uniform(0.7, 1.3, w[0].shape).astype(np.float32); w[1] = rng.normal(0, 0.2, w[1].shape).astype(np.float32)
            n.set_weights(w)
    return ns, no


def refocus(graphs, focus, rng):
    if focus == 'g': return graphs
    out = []
    for g in graphs:
        n = (g.nodes if focus == 'n' else g.arcs).shape[0]
        om = rng.random(n) < 0.7
        t = np.zeros((int(om.sum()), 2)); t[np.arange(len(t)), rng.integers(0, 2, len(t))] = 1
        out.append(GraphObject(nodes=g.nodes, arcs=g.arcs, targets=t, focus=focus, set_mask=rng.random(n) < 0.8,
                               output_mask=om, sample_weight=rng.uniform(0.5, 1.5, len(t))))
    return out


def oracle_step(model, x, y, sw, s0, loss, avg=False, dtype=torch.float64, seed=None, checkpoint_iterations=False):
    nodes, arcs, _, sm, om, adj, an, ng = x
    mask = np.logical_and(_np(sm).reshape(-1), _np(om).reshape(-1))
    return torch_train.train_step(_np(nodes), _np(arcs), _triple(adj), _triple(an), _triple(ng), mask,
                                  net_state=model.net_state.spec(), net_output=model.net_output.spec(),
                                  state_vect_dim=model.state_vect_dim, max_iteration=model.max_iteration,
                                  state_threshold=model.state_threshold, focus=model._focus, state0=s0, y=_np(y),
                                  sample_weight=_np(sw), loss=loss, average_st_grads=avg, dtype=dtype, seed=seed,
                                  checkpoint_iterations=checkpoint_iterations)


# ---- float64 oracle results shared between tests and PREFETCHED on worker threads ----------------------------------------------------------
# The large-graph training tests spend most of their time in the float64 autograd oracle (seconds of host work each, the GPU idle).  A test
# that passes `oracle_key` to check_step (its inputs are seeded: the key names them) gets its oracle result from a cache - the alternative-
# kernel tests re-run configurations - and tests/conftest.py starts the tests marked `@prefetch_oracle` on a small thread pool at collection
# time in ORACLE-ONLY mode (the test body runs up to its check_step, which computes and stores the oracle result and stops), so that the
# real test finds it ready.
import threading

_ORACLE_CACHE, _ORACLE_LOCKS, _ORACLE_GUARD = {}, {}, threading.Lock()
_ORACLE_ONLY = threading.local()


class _OracleDone(Exception):
    pass


def cached_oracle(key, fn):
    """fn() once per key, whichever thread asks first; the others wait for it."""
    with _ORACLE_GUARD: lock = _ORACLE_LOCKS.setdefault(key, threading.Lock())
    with lock:
        if key not in _ORACLE_CACHE: _ORACLE_CACHE[key] = fn()
    return _ORACLE_CACHE[key]


def prefetch_oracle(fn):
    """Marks a parametrised test without fixtures whose body reaches check_step(.., oracle_key=..): conftest runs it ahead in oracle-only mode."""
    fn._prefetch_oracle = True
    return fn


def run_oracle_only(fn, kwargs):
    _ORACLE_ONLY.on = True
    try: fn(**kwargs)
    except _OracleDone: pass
    except BaseException: pass                 # (the real test will show it)
    finally: _ORACLE_ONLY.on = False


def check_step(model, x, y, sw, s0, loss='categorical_crossentropy', avg=False, native=None, seed=None, oracle_key=None):
    from gnnkeras_amd.Models.training import LoopTrainer
    if not getattr(_ORACLE_ONLY, 'on', False): model.compile(optimizer=SGD(0.0), loss=loss, average_st_grads=avg)
    if oracle_key is not None: want = cached_oracle(oracle_key, lambda: oracle_step(model, x, y, sw, s0, loss, avg, seed=seed))
    else: want = oracle_step(model, x, y, sw, s0, loss, avg, seed=seed)
    if getattr(_ORACLE_ONLY, 'on', False): raise _OracleDone()
    before = [w.copy() for w in model.net_state.get_weights() + model.net_output.get_weights()]
    if native is None:      # both orchestrations: the in-library step (gnn_train_step) and the building blocks driven from Python
        moving = [w.copy() for w in model.net_state.get_weights() + model.net_output.get_weights()]
        check_step(model, x, y, sw, s0, loss, avg, native=False, seed=seed, oracle_key=oracle_key)
        for net, n0 in ((model.net_state, 0), (model.net_output, len(model.net_state.get_weights()))):
            net.set_weights(moving[n0:n0 + len(net.get_weights())])               # the first pass moved the BN moving statistics
        return check_step(model, x, y, sw, s0, loss, avg, native=True, seed=seed, oracle_key=oracle_key)
    tr = LoopTrainer(model)
    tr.use_native_step = native
    res = tr.train_step(x, y, sw, state0=None if s0 is None else torch.from_numpy(s0).cuda(), apply=False, seed=seed)
    assert res['k'] == want['k']
    assert abs(float(res['loss']) - want['loss']) <= 1e-5 * max(1.0, abs(want['loss']))
    # training mode: BatchNormalization on the statistics of a small batch multiplies rounding by 1/sigma of thin columns
    e_pred = rel_err(res['y_pred'].cpu().numpy(), want['y_pred'])
    rows = []
    for name, ng_, got, ref, kinks in [('state', tr.gs, tr.gs.gradients(), want['grads_state'], want.get('kinks_state')),
                                       ('output', tr.go, tr.go.gradients(), want['grads_output'], want.get('kinks_output'))]:
        assert len(got) == len(ref)
        scale = max(float(np.max(np.abs(x_))) for x_ in ref)
        rows += grad_rows(name, got, ref, ng_.bn, kinks, scale, int(x[0].shape[0]) if name == 'state' else int(want['y_pred'].shape[0]))
    import os
    log_rows(os.environ.get('PYTEST_CURRENT_TEST', ''), rows, native=bool(native), y_pred=e_pred, n_nodes=int(x[0].shape[0]), k=int(res['k']))
    assert e_pred <= Y_PRED_BAR, e_pred
    bad = [r_ for r_ in rows if not r_['ok']]
    assert not bad, bad
    # moving statistics: k updates for the state network, one for the output network
    for net, key in [(model.net_state, 'moving_state'), (model.net_output, 'moving_output')]:
        if net.batch_normalization:
            w = net.get_weights()
            assert rel_err(w[2], want[key][0]) <= 1e-5 and rel_err(w[3], want[key][1]) <= 1e-5
    # trainable weights untouched with apply=False
    after = model.net_state.get_weights() + model.net_output.get_weights()
    tr_idx = [i for i in range(len(after))]
    return res, want


@pytest.mark.parametrize('focus', ['g', 'n', 'a'])
@pytest.mark.parametrize('bn', [False, True])
def test_gradients_single_layer(mutag_graphs, focus, bn):
    rng = np.random.default_rng(5)
    gl = refocus([g.copy() for g in mutag_graphs[:16]], focus, rng)
    seq = MultiGraphSequencer(gl, focus, 'average', 16, shuffle=False)
    x, y, sw = seq[0]
    d = 16
    ns, no = nets(focus, d, bn)
    model = CLS[focus](ns, no, d, 6, 0.0)
    s0 = rng.normal(0, 0.1, (x[0].shape[0], d)).astype(np.float32)
    check_step(model, x, y, sw, s0)


@pytest.mark.parametrize('bn', [False, True])
def test_gradients_hidden_layers_early_exit_and_average(mutag_graphs, bn):
    rng = np.random.default_rng(6)
    seq = MultiGraphSequencer(mutag_graphs[:24], 'g', 'average', 24, shuffle=False)
    x, y, sw = seq[0]
    d = 12
    ns, no = nets('g', d, bn, hidden_state=[20, 9], hidden_out=[7], act='tanh', scale=0.25)
    model = GNNgraphBased(ns, no, d, 30, 0.02)
    s0 = rng.normal(0, 0.1, (x[0].shape[0], d)).astype(np.float32)
    res, want = check_step(model, x, y, sw, s0, avg=True)
    assert 1 < want['k'] < 30                                       # the loop really stops early; grads / k


def test_gradients_state_dim_0_mse_sum_aggregation(mutag_graphs):
    rng = np.random.default_rng(7)
    seq = MultiGraphSequencer(mutag_graphs[:16], 'g', 'sum', 16, shuffle=False)
    x, y, sw = seq[0]
    ns, no = nets('g', 0, True, out_act='sigmoid', scale=0.2)
    model = GNNgraphBased(ns, no, 0, 4, 0.0)
    check_step(model, x, y, sw, None, loss='mse')


def test_gradients_with_weight_regularizers(mutag_graphs):
    """`MLP(kernel_regularizer=, bias_regularizer=)` (reference MLP.py:12-15, :48-49; train_step adds `self.losses`,
    GNN.py:286): the penalty enters the loss once per variable and its gradient every trainable Dense variable, before the
    1/k of `average_st_grads`."""
    from gnnkeras_amd.Models.MLP import l1, l2, l1_l2
    rng = np.random.default_rng(9)
    seq = MultiGraphSequencer(mutag_graphs[:16], 'g', 'average', 16, shuffle=False)
    x, y, sw = seq[0]
    d = 8
    inp, lay = get_inout_dims('state', 14, 3, 2, 'g', d, hidden_units=[10])
    ns = MLP(inp[0], lay, 'tanh', 'lecun_normal', 'lecun_normal', kernel_regularizer=[l2(0.02), l1_l2(0.003, 0.01)],
             bias_regularizer=[None, l1(0.05)], rng=0)
    ns.set_weights([a * 0.3 if a.ndim == 2 else a + 0.1 for a in ns.get_weights()])
    inp, lay = get_inout_dims('output', 14, 3, 2, 'g', d)
    no = MLP(inp[0], lay, 'softmax', 'glorot_normal', 'glorot_normal', kernel_regularizer='l2', rng=1)
    model = GNNgraphBased(ns, no, d, 5, 0.0)
    s0 = rng.normal(0, 0.1, (x[0].shape[0], d)).astype(np.float32)
    res, want = check_step(model, x, y, sw, s0, avg=True)
    assert want['loss'] > 0.05          # the penalty is not negligible next to the cross-entropy (~0.7)
    with pytest.raises(ValueError):
        MLP(inp[0], lay, 'softmax', 'glorot_normal', 'glorot_normal', kernel_regularizer=lambda w: w.sum())


@pytest.mark.parametrize('focus,bn,alpha', [('g', True, False), ('n', False, False), ('a', True, True), ('n', True, False)])
def test_dropout_in_front_of_the_first_dense(mutag_graphs, focus, bn, alpha):
    """Position 0 (reference MLP.py:60-71: Dropout inserted in front of Dense 0, BatchNormalization in front of both): the
    normalised input is materialised through an identity Dense carrying the batch statistics, dropped out, and Dense 0 runs plain -
    in the state network (fresh masks every iteration) and in the output network; k, loss, predictions, every gradient (gamma /
    beta through the masks included) and the moving statistics against torch autograd fed the same masks."""
    rng = np.random.default_rng(21)
    gl = refocus([g.copy() for g in mutag_graphs[:10]], focus, rng)
    seq = MultiGraphSequencer(gl, focus, 'average', 10, shuffle=False)
    x, y, sw = seq[0]
    d = 5
    inp, lay = get_inout_dims('state', 14, 3, 2, focus, d, hidden_units=[8])
    ns = MLP(inp[0], lay, ['selu' if alpha else 'tanh', 'tanh'], 'lecun_normal', 'lecun_normal', dropout_rate=[0.2, 0.1], dropout_pos=[0, 1],
             alphadropout=alpha, batch_normalization=bn, rng=0)
    inp, lay = get_inout_dims('output', 14, 3, 2, focus, d)
    no = MLP(inp[0], lay, 'softmax', 'glorot_normal', 'glorot_normal', dropout_rate=0.15, dropout_pos=0, alphadropout=alpha,
             batch_normalization=bn, rng=1)
    if bn:
        r2 = np.random.default_rng(3)
        for n in (ns, no):
            w = n.get_weights()
            w[0] = r2.uniform(0.7, 1.3, w[0].shape).astype(np.float32); w[1] = r2.normal(0, 0.2, w[1].shape).astype(np.float32)
            n.set_weights(w)
    model = CLS[focus](ns, no, d, 3, 0.0)
    s0 = rng.normal(0, 0.1, (x[0].shape[0], d)).astype(np.float32)
    res, want = check_step(model, x, y, sw, s0, seed=5)
    assert res['k'] == 3
    model.compile(optimizer='adam', loss='categorical_crossentropy')
    model.fit(seq, epochs=1, verbose=0)
    out = model(x)                                                  # inference ignores Dropout layers, as Keras does
    assert out.shape[1] == 2


def test_dropout_kernel_draws_the_documented_mask():
    """gnn_dropout against the numpy restatement of its counter hash: same keep mask, Keras Dropout / AlphaDropout values,
    the backward of both, and in-place operation."""
    from gnnkeras_amd import _native as nat
    from oracle.torch_train import dropout_keep_mask, mix32
    rng = np.random.default_rng(0)
    for M, H, rate in ((935, 32, 0.2), (1, 7, 0.5), (4099, 3, 0.05)):
        x = rng.normal(size=(M, H)).astype(np.float32)
        key = mix32(11, 3, M, H)
        keep = dropout_keep_mask(key, M, H, rate)
        assert abs(keep.mean() - (1 - rate)) < 0.05 + 2.0 / np.sqrt(M * H)
        xd = torch.from_numpy(x).cuda()
        for alpha in (0, 1):
            y = torch.empty_like(xd)
            nat.check(nat.lib().gnn_dropout(nat.ptr(xd), H, nat.ptr(y), H, M, H, rate, key, alpha, 0, None))
            g = xd.clone()
            nat.check(nat.lib().gnn_dropout(nat.ptr(g), H, nat.ptr(g), H, M, H, rate, key, alpha, 1, None))     # in place
            r = float(np.float32(rate))
            if alpha:
                ap = -1.6732632423543772 * 1.0507009873554805
                a = ((1 - r) * (1 + r * ap ** 2)) ** -0.5
                want, wantg = a * np.where(keep, x, ap) + (-a * ap * r), np.where(keep, x * a, 0.0)
            else:
                want, wantg = np.where(keep, x / (1 - r), 0.0), np.where(keep, x / (1 - r), 0.0)
            assert np.allclose(y.cpu().numpy(), want, rtol=1e-6, atol=1e-6)
            assert np.allclose(g.cpu().numpy(), wantg, rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize('focus,alpha', [('g', False), ('n', False), ('a', True), ('g', True)])
def test_dropout_gradients_match_autograd_with_the_same_masks(mutag_graphs, focus, alpha):
    """Dropout / AlphaDropout behind the state network's Dense layers (fresh masks every iteration of the loop) and inside the
    output network (reference MLP.py:60-66): forward values, loss and every gradient against torch autograd fed the same
    counter-hash masks; the backward sweep regenerates each iteration's masks from its key."""
    rng = np.random.default_rng(12)
    gl = refocus([g.copy() for g in mutag_graphs[:12]], focus, rng)
    seq = MultiGraphSequencer(gl, focus, 'average', 12, shuffle=False)
    x, y, sw = seq[0]
    d = 6
    inp, lay = get_inout_dims('state', 14, 3, 2, focus, d, hidden_units=[9])
    ns = MLP(inp[0], lay, ['tanh', 'selu' if alpha else 'tanh'], 'lecun_normal', 'lecun_normal', dropout_rate=[0.25, 0.1],
             dropout_pos=[1, 2], alphadropout=alpha, rng=0)
    inp, lay = get_inout_dims('output', 14, 3, 2, focus, d, hidden_units=[7])
    no = MLP(inp[0], lay, ['selu' if alpha else 'relu', 'softmax'], 'glorot_normal', 'glorot_normal', dropout_rate=0.3, dropout_pos=1,
             alphadropout=alpha, rng=1)
    model = CLS[focus](ns, no, d, 4, 0.0)
    s0 = rng.normal(0, 0.1, (x[0].shape[0], d)).astype(np.float32)
    res, want = check_step(model, x, y, sw, s0, seed=77)
    assert res['k'] == 4
    # another seed: other masks, other loss (the masks really are applied)
    from gnnkeras_amd.Models.training import LoopTrainer
    other = LoopTrainer(model).train_step(x, y, sw, state0=torch.from_numpy(s0).cuda(), apply=False, seed=78)
    assert abs(float(other['loss']) - float(res['loss'])) > 1e-6
    # fit() runs (masks from the model's step counter), inference ignores the Dropout layers
    model.compile(optimizer='adam', loss='categorical_crossentropy')
    model.fit(seq, epochs=1, verbose=0)
    k, st, o = model.Loop(*model.process_inputs(x), state0=torch.from_numpy(s0).cuda())
    k2, st2, o2 = model.Loop(*model.process_inputs(x), state0=torch.from_numpy(s0).cuda())
    assert torch.equal(o, o2)


@pytest.mark.parametrize('focus', ['g', 'n'])
def test_call_training_true_returns_k_state_out_on_batch_statistics(mutag_graphs, focus):
    """`gnn(x, training=True)` outside train_step (reference GNN.py:165-177 returns `(k, state, out)`): BatchNormalization
    networks — the starter default, MLP.py:14 — run on BATCH statistics; against the oracle's training=True forward."""
    from oracle.harness import oracle_loop
    rng = np.random.default_rng(12)
    gl = refocus([g.copy() for g in mutag_graphs[:16]], focus, rng)
    seq = MultiGraphSequencer(gl, focus, 'average', 16, shuffle=False)
    x = seq[0][0]
    d = 16
    ns, no = nets(focus, d, True, scale=0.3)
    model = CLS[focus](ns, no, d, 8, 0.02)
    s0 = rng.normal(0, 0.1, (x[0].shape[0], d)).astype(np.float32)
    before = ns.get_weights()[2].copy()
    k64, st64, o64 = oracle_loop(model, x, s0, np.float64, training=True)
    k, st, o = model.Loop(*model.process_inputs(x), training=True, state0=torch.from_numpy(s0).cuda())
    assert float(k) == float(k64) and 1 < float(k) <= 8
    assert rel_err(st.cpu().numpy(), st64) <= 5e-5 and rel_err(o.cpu().numpy(), o64) <= 5e-5
    assert not np.array_equal(ns.get_weights()[2], before)            # the moving mean moved: training mode really ran
    ki, sti, oi = model.Loop(*model.process_inputs(x), training=False, state0=torch.from_numpy(s0).cuda())
    assert rel_err(oi.cpu().numpy(), o64) > 1e-3                      # and inference statistics give a different answer


def test_adam_step_matches_reference_formula(mutag_graphs):
    rng = np.random.default_rng(8)
    seq = MultiGraphSequencer(mutag_graphs[:16], 'g', 'average', 16, shuffle=False)
    x, y, sw = seq[0]
    d = 8
    ns, no = nets('g', d, True)
    model = GNNgraphBased(ns, no, d, 5, 0.0)
    model.compile(optimizer=Adam(learning_rate=0.01), loss='categorical_crossentropy', metrics=['accuracy'])
    s0 = rng.normal(0, 0.1, (x[0].shape[0], d)).astype(np.float32)
    tv = lambda net: [w for i, w in enumerate(net.get_weights()) if not (net.batch_normalization and i in (2, 3))]
    params = tv(ns) + tv(no)
    m = [np.zeros_like(p, dtype=np.float64) for p in params]; v = [np.zeros_like(p, dtype=np.float64) for p in params]
    params = [p.astype(np.float64) for p in params]
    for step in (1, 2, 3):
        want = oracle_step(model, x, y, sw, s0, 'categorical_crossentropy')
        grads = want['grads_state'] + want['grads_output']
        upd = [torch_train.adam_update(p, g, mi, vi, step, lr=0.01) for p, g, mi, vi in zip(params, grads, m, v)]
        params, m, v = [u[0] for u in upd], [u[1] for u in upd], [u[2] for u in upd]
        logs = model.train_step((x, y, sw), state0=torch.from_numpy(s0).cuda())
        assert abs(float(logs['loss']) - want['loss']) < 1e-4
        got = tv(model.net_state) + tv(model.net_output)
        for g_, p_ in zip(got, params):
            assert np.max(np.abs(g_ - p_)) <= 2e-4 * max(1.0, np.max(np.abs(p_)))
        assert 0.0 <= float(logs['accuracy']) <= 1.0


@pytest.mark.parametrize('bn', [True, False])
def test_fit_reduces_loss_starter_config(mutag_graphs, bn):
    """BASELINE config C1 plumbing: starter.py hyper-parameters (dim_state 0, max_iter 5, threshold 0.01, 'average',
    graph focus, [BN +] Dense selu / softmax, Adam lr 0.01, categorical cross-entropy), fit + evaluate on the device.
    With BatchNormalization the inference-mode (moving statistics, momentum 0.99) validation loss lags the training
    loss for the first few hundred updates exactly as in Keras, so improvement of the validation loss is asserted for
    the BN-free variant only."""
    gs = [g.copy() for g in mutag_graphs[:320]]
    for g in gs: g.setAggregation('average')
    tr = MultiGraphSequencer(gs[:256], 'g', 'average', 32, shuffle=True)
    va = MultiGraphSequencer(gs[256:], 'g', 'average', 32, shuffle=False)
    inp, lay = get_inout_dims('state', 14, 3, 2, 'g', 0)
    ns = MLP(inp[0], lay, 'selu', 'lecun_normal', 'lecun_normal', rng=0, batch_normalization=bn)
    inp, lay = get_inout_dims('output', 14, 3, 2, 'g', 0)
    no = MLP(inp[0], lay, 'softmax', 'glorot_normal', 'glorot_normal', rng=1, batch_normalization=bn)
    gnn = GNNgraphBased(ns, no, 0, 5, 0.01)
    gnn.compile(optimizer=Adam(learning_rate=0.01), loss='categorical_crossentropy', average_st_grads=False,
                metrics=['accuracy'], run_eagerly=True)
    np.random.seed(0)
    before = gnn.evaluate(va, return_dict=True)
    hist = gnn.fit(tr, epochs=6, validation_data=va, verbose=0)
    after = gnn.evaluate(va, return_dict=True)
    assert len(hist['loss']) == 6 and hist['loss'][-1] < hist['loss'][0]
    assert np.isfinite(hist['val_loss']).all() and np.isfinite(hist['accuracy']).all()
    if not bn:
        assert after['loss'] < before['loss']


# ----------------------------------------------------------------------------------------------------------------------
# composite (heterogeneous) models: one state network per node type (reference CompositeGNN.py:275-304)
# ----------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('native', [False, True])
@pytest.mark.parametrize('focus', ['n', 'g', 'a'])
@pytest.mark.parametrize('bn', [False, True])
def test_composite_gradients(focus, bn, native):
    """Both orchestrations against torch autograd in float64: the device building blocks driven from Python (`native=False`) and the
    whole step inside the library (`gnn_train_step` with `composite`, csrc/train_composite.hpp; node, graph and - round 6 - arc focus:
    CompositeGNN.py:315-327)."""
    from gnnkeras_amd import CompositeGraphObject
    from gnnkeras_amd.Models.CompositeGNN import CompositeGNNnodeBased, CompositeGNNarcBased, CompositeGNNgraphBased
    from gnnkeras_amd.Models.training import LoopTrainer
    from gnnkeras_amd.Sequencers.GraphSequencers import CompositeMultiGraphSequencer
    CC = {'n': CompositeGNNnodeBased, 'a': CompositeGNNarcBased, 'g': CompositeGNNgraphBased}[focus]
    rng = np.random.default_rng(31)
    dims, A, D, T = (4, 2, 3), 2, 10, 2

    def cg(n, e):
        pairs = set()
        while len(pairs) < e:
            a, b = rng.integers(0, n, 2)
            if a != b: pairs.add((int(a), int(b)))
        ids = np.array(sorted(pairs), dtype=float)
        arcs = np.concatenate([ids, rng.normal(size=(e, A))], 1)
        types = rng.integers(0, 3, n); types[:3] = [0, 1, 2]
        tm = np.zeros((n, 3), bool); tm[np.arange(n), types] = True
        nt = {'n': n, 'a': e, 'g': 1}[focus]
        om = rng.random(nt) < 0.8 if focus != 'g' else np.ones(n, bool)
        tg = np.zeros((int(om.sum()) if focus != 'g' else 1, T)); tg[np.arange(len(tg)), rng.integers(0, T, len(tg))] = 1
        kw = dict(output_mask=om) if focus != 'g' else {}
        return CompositeGraphObject(nodes=rng.normal(size=(n, 4)), arcs=arcs, targets=tg, type_mask=tm, dim_node_label=dims,
                                    focus=focus, aggregation_mode='composite_average', **kw)
    seq = CompositeMultiGraphSequencer([cg(60, 200), cg(35, 90), cg(80, 260)], focus, 'composite_average', 3, shuffle=False)
    x, y, sw = seq[0]
    inp, lay = get_inout_dims('state', dims, A, T, focus, D, hidden_units=[12] if bn else None)
    ns = [MLP(i, lay, 'tanh', 'lecun_normal', 'lecun_normal', rng=t, batch_normalization=bn) for t, i in enumerate(inp)]
    for n in ns: n.set_weights([a * 0.5 if a.ndim == 2 else a for a in n.get_weights()])
    inp, lay = get_inout_dims('output', dims, A, T, focus, D)
    no = MLP(inp[0], lay, 'softmax', 'glorot_normal', 'glorot_normal', rng=9, batch_normalization=bn)
    model = CC(ns, no, D, 5, 0.0)
    model.compile(optimizer=SGD(0.0), loss='categorical_crossentropy')
    N = x[0].shape[0]
    s0 = rng.normal(0, 0.1, (N, D)).astype(np.float32)
    nodes, arcs, dnl, tmask, sm, om_, cas, adj, an, ng = x
    mask = np.logical_and(_np(sm).reshape(-1), _np(om_).reshape(-1))
    want = torch_train.composite_train_step(
        _np(nodes), _np(arcs), _np(dnl).reshape(-1), _np(tmask).reshape(3, -1), [_triple(c) for c in cas], _triple(adj),
        _triple(an), _triple(ng), mask, net_state=[n.spec() for n in ns], net_output=no.spec(), state_vect_dim=D,
        max_iteration=5, state_threshold=0.0, focus=focus, state0=s0, y=_np(y), sample_weight=_np(sw),
        loss='categorical_crossentropy')
    tr = LoopTrainer(model)
    tr.use_native_step = native
    assert tr._native_step_applies(y) == native
    res = tr.train_step(x, y, sw, state0=torch.from_numpy(s0).cuda(), apply=False)
    assert res['k'] == want['k'] == 5
    assert abs(float(res['loss']) - want['loss']) <= 1e-5 * max(1.0, abs(want['loss']))
    e_pred = rel_err(res['y_pred'].cpu().numpy(), want['y_pred'])
    assert e_pred <= Y_PRED_BAR, e_pred
    rows_t = _np(tmask).reshape(3, -1).sum(1)
    check_network_grads('', [(f'state{t}', g_, want['grads_state'][t], want['kinks_state'][t], int(rows_t[t])) for t, g_ in enumerate(tr.gs)]
                        + [('output', tr.go, want['grads_output'], want['kinks_output'], int(want['y_pred'].shape[0]))], native=bool(native), y_pred=e_pred)
    if bn:
        for n, mv in zip(ns, want['moving_state']):
            w = n.get_weights()
            assert rel_err(w[2], mv[0]) <= 1e-5 and rel_err(w[3], mv[1]) <= 1e-5


def test_composite_fit_runs():
    from gnnkeras_amd.Models.CompositeGNN import CompositeGNNnodeBased
    from gnnkeras_amd.Sequencers.GraphSequencers import CompositeMultiGraphSequencer
    from gnnkeras_amd.synth import er_composite_graph
    dims = (5, 3, 4)
    gs_ = [er_composite_graph(300 + 10 * i, 2000, dim_node_label=dims, seed=i) for i in range(6)]
    seq = CompositeMultiGraphSequencer(gs_, 'n', 'average', 2, shuffle=True)
    inp, lay = get_inout_dims('state', dims, 3, 2, 'n', 8)
    ns = [MLP(i, lay, 'selu', 'lecun_normal', 'lecun_normal', rng=t) for t, i in enumerate(inp)]
    inp, lay = get_inout_dims('output', dims, 3, 2, 'n', 8)
    no = MLP(inp[0], lay, 'softmax', 'glorot_normal', 'glorot_normal', rng=9)
    model = CompositeGNNnodeBased(ns, no, 8, 4, 0.01)
    model.compile(optimizer=Adam(0.01), loss='categorical_crossentropy', metrics=['accuracy'])
    np.random.seed(1)
    hist = model.fit(seq, epochs=5, verbose=0)
    assert hist['loss'][-1] < hist['loss'][0] and np.isfinite(hist['accuracy']).all()


# ----------------------------------------------------------------------------------------------------------------------
# Layered GNN (reference GNN/Models/LGNN.py): forward, joint training modes with gradients chained through
# update_graph, serial fit
# ----------------------------------------------------------------------------------------------------------------------
def lgnn_stack(focus, d, n_layers, get_state, get_output, bn, max_it=4, T=2):
    from gnnkeras_amd.Models.LGNN import LGNN
    gnns = []
    for i in range(n_layers):
        inp, lay = get_inout_dims('state', 14, 3, T, focus, d, layer=i, get_state=get_state, get_output=get_output)
        ns = MLP(inp[0], lay, 'tanh', 'lecun_normal', 'lecun_normal', rng=10 + i, batch_normalization=bn)
        ns.set_weights([a * 0.5 if a.ndim == 2 else a for a in ns.get_weights()])
        inp, lay = get_inout_dims('output', 14, 3, T, focus, d, layer=i, get_state=get_state, get_output=get_output)
        no = MLP(inp[0], lay, 'softmax', 'glorot_normal', 'glorot_normal', rng=20 + i, batch_normalization=bn)
        gnns.append(CLS[focus](ns, no, d, max_it, 0.0))
    return LGNN(gnns, get_state, get_output)


@pytest.mark.parametrize('focus,d,get_state,get_output', [('g', 8, True, True), ('n', 8, True, False), ('n', 0, True, True),
                                                          ('g', 8, False, True), ('a', 6, True, False)])
@pytest.mark.parametrize('mode', ['parallel', 'residual'])
def test_lgnn_joint_training_gradients(mutag_graphs, focus, d, get_state, get_output, mode):
    rng = np.random.default_rng(12)
    gl = refocus([g.copy() for g in mutag_graphs[:12]], focus, rng)
    seq = MultiGraphSequencer(gl, focus, 'average', 12, shuffle=False)
    x, y, sw = seq[0]
    bn = d != 0
    lg = lgnn_stack(focus, d, 3, get_state, get_output, bn)
    lg.compile(optimizer=SGD(0.0), loss='categorical_crossentropy', training_mode=mode, average_st_grads=True, metrics=['accuracy'])
    N = x[0].shape[0]
    s0s = [rng.normal(0, 0.1, (N, d)).astype(np.float32) if d else None for _ in range(3)]
    nodes, arcs, _, sm, om, adj, an, ng = x
    mask = np.logical_and(_np(sm).reshape(-1), _np(om).reshape(-1))
    layers = [dict(net_state=g.net_state.spec(), net_output=g.net_output.spec(), state_vect_dim=d, max_iteration=4,
                   state_threshold=0.0) for g in lg.gnns]
    want = torch_train.lgnn_train_step(_np(nodes), _np(arcs), _triple(adj), _triple(an), _triple(ng), mask, layers=layers,
                                       get_state=get_state, get_output=get_output, focus=focus, state0s=s0s, y=_np(y),
                                       sample_weight=_np(sw), loss='categorical_crossentropy', training_mode=mode,
                                       average_st_grads=True)
    logs = lg.train_step((x, y, sw), state0=[None if s is None else torch.from_numpy(s).cuda() for s in s0s], apply=False)
    assert logs['k'] == want['k']
    assert abs(float(logs['loss']) - want['loss']) <= 1e-5 * max(1.0, abs(want['loss']))
    nets_ = []
    for li, tp in enumerate(lg._last_tapes):       # (tanh / softmax networks: no activation kinks)
        nets_ += [(f'layer{li}.state', tp.gs[0], want['grads'][li][0], None, N), (f'layer{li}.output', tp.go, want['grads'][li][1], None, N)]
    check_network_grads('', nets_)
    # eval-mode forward of the stack agrees with the layers' own Loop chain and has the reference's list layout
    K, states, outs = lg.Loop(*lg.process_inputs(x), state0=[None if s is None else torch.from_numpy(s).cuda() for s in s0s])
    assert len(K) == len(states) == len(outs) == 3 and outs[-1].shape == lg(x).shape


@pytest.mark.parametrize('T,get_state', [(4, True), (2, False), (3, False)])
@pytest.mark.parametrize('mode', ['parallel', 'residual'])
def test_lgnn_arc_focused_joint_training_with_get_output(mutag_graphs, T, get_state, mode):
    """Arc-focused stack with `get_output` (reference LGNN.py:252-287 + the arc branch of update_graph, :203-209): layer i's
    per-arc output is PREPENDED to the arcs matrix, so output columns 2.. become arc-label columns of layer i + 1 (columns 0 / 1
    land where the ids were and are never read: with T = 2 no gradient flows back through the output at all).  The gradient
    reaches layer i through the ArcNode scatter-add of every iteration and through the output network's arc-label segment."""
    rng = np.random.default_rng(31)
    gl = []
    for g in mutag_graphs[:10]:
        n = g.arcs.shape[0]
        om = rng.random(n) < 0.7
        t = np.zeros((int(om.sum()), T)); t[np.arange(len(t)), rng.integers(0, T, len(t))] = 1
        gl.append(GraphObject(nodes=g.nodes, arcs=g.arcs, targets=t, focus='a', set_mask=rng.random(n) < 0.8, output_mask=om,
                              sample_weight=rng.uniform(0.5, 1.5, len(t))))
    seq = MultiGraphSequencer(gl, 'a', 'average', 10, shuffle=False)
    x, y, sw = seq[0]
    d = 6
    lg = lgnn_stack('a', d, 3, get_state, True, True, T=T)
    lg.compile(optimizer=SGD(0.0), loss='categorical_crossentropy', training_mode=mode, average_st_grads=False)
    N = x[0].shape[0]
    s0s = [rng.normal(0, 0.1, (N, d)).astype(np.float32) for _ in range(3)]
    nodes, arcs, _, sm, om, adj, an, ng = x
    mask = np.logical_and(_np(sm).reshape(-1), _np(om).reshape(-1))
    layers = [dict(net_state=g.net_state.spec(), net_output=g.net_output.spec(), state_vect_dim=d, max_iteration=4,
                   state_threshold=0.0) for g in lg.gnns]
    want = torch_train.lgnn_train_step(_np(nodes), _np(arcs), _triple(adj), _triple(an), _triple(ng), mask, layers=layers,
                                       get_state=get_state, get_output=True, focus='a', state0s=s0s, y=_np(y),
                                       sample_weight=_np(sw), loss='categorical_crossentropy', training_mode=mode)
    logs = lg.train_step((x, y, sw), state0=[torch.from_numpy(s).cuda() for s in s0s], apply=False)
    assert logs['k'] == want['k']
    assert abs(float(logs['loss']) - want['loss']) <= 1e-5 * max(1.0, abs(want['loss']))
    nets_ = []
    for li, tp in enumerate(lg._last_tapes):
        nets_ += [(f'layer{li}.state', tp.gs[0], want['grads'][li][0], None, N), (f'layer{li}.output', tp.go, want['grads'][li][1], None, N)]
    check_network_grads('', nets_)
    if T > 2:       # the chain through the arc labels carries a real gradient: layer 0's output net sees more than its own loss
        assert float(np.max(np.abs(want['grads'][0][1][-1]))) > 0


def test_lgnn_serial_fit_and_persistence(mutag_graphs, tmp_path):
    from gnnkeras_amd.Models.LGNN import LGNN
    gs = [g.copy() for g in mutag_graphs[:96]]
    for g in gs: g.setAggregation('average')
    tr = MultiGraphSequencer(gs[:64], 'g', 'average', 16, shuffle=True)
    va = MultiGraphSequencer(gs[64:], 'g', 'average', 16, shuffle=False)
    lg = lgnn_stack('g', 0, 3, True, True, bn=False, max_it=3)          # starter.py: 3 layers, get_state, get_output
    lg.compile(optimizer=Adam(0.01), loss='categorical_crossentropy', training_mode='serial', average_st_grads=True,
               metrics=['accuracy'])
    np.random.seed(2)
    hists = lg.fit(tr, epochs=2, validation_data=va, verbose=0)
    assert len(hists) == 3 and all(len(h['loss']) == 2 and np.isfinite(h['val_loss']).all() for h in hists)
    res = lg.evaluate(va, return_dict=True)
    assert np.isfinite(res['loss']) and 0.0 <= res['accuracy'] <= 1.0
    lg.save(str(tmp_path / 'lgnn'))
    back = LGNN.load(str(tmp_path / 'lgnn'))
    assert back.LAYERS == 3 and back.get_state and back.get_output
    back.compile(optimizer=Adam(0.01), loss='categorical_crossentropy', training_mode='serial', metrics=['accuracy'])
    assert np.allclose(back.predict(va), lg.predict(va), atol=1e-6)
    assert 'layers=3' in repr(lg) and lg.copy().LAYERS == 3


def test_composite_lgnn_forward_and_serial_fit():
    from gnnkeras_amd.Models.CompositeGNN import CompositeGNNnodeBased
    from gnnkeras_amd.Models.CompositeLGNN import CompositeLGNN
    from gnnkeras_amd.Sequencers.GraphSequencers import CompositeMultiGraphSequencer
    from gnnkeras_amd.synth import er_composite_graph
    dims, D = (5, 3, 4), 6
    gs_ = [er_composite_graph(200 + 10 * i, 1500, dim_node_label=dims, seed=i) for i in range(6)]
    seq = CompositeMultiGraphSequencer(gs_, 'n', 'average', 2, shuffle=False)
    gnns = []
    for layer in range(2):
        inp, lay = get_inout_dims('state', dims, 3, 2, 'n', D, layer=layer, get_state=True, get_output=True)
        ns = [MLP(i, lay, 'tanh', 'lecun_normal', 'lecun_normal', rng=30 + t + 10 * layer) for t, i in enumerate(inp)]
        inp, lay = get_inout_dims('output', dims, 3, 2, 'n', D, layer=layer, get_state=True, get_output=True)
        no = MLP(inp[0], lay, 'softmax', 'glorot_normal', 'glorot_normal', rng=50 + layer)
        gnns.append(CompositeGNNnodeBased(ns, no, D, 3, 0.0))
    lg = CompositeLGNN(gnns, True, True)
    lg.compile(optimizer=Adam(0.01), loss='categorical_crossentropy', training_mode='serial', metrics=['accuracy'])
    x = seq[0][0]
    K, states, outs = lg.Loop(*lg.process_inputs(x), seed=0)
    assert len(outs) == 2 and outs[1].shape == outs[0].shape == (x[0].shape[0], 2)
    # layer 2 really consumes [state | out | labels]: its first-type state net input is d_t + (D + T) wider than layer 1's
    assert gnns[1].net_state[0].input_dim - gnns[0].net_state[0].input_dim == 4 * (D + 2)
    hists = lg.fit(seq, epochs=2, verbose=0)
    assert len(hists) == 2 and all(np.isfinite(h['loss']).all() for h in hists)
    assert 'CompositeLGNN' in repr(lg)


@pytest.mark.parametrize('mode', ['parallel', 'residual'])
@pytest.mark.parametrize('focus,D', [('n', 6), ('g', 5)])
def test_composite_lgnn_joint_training_gradients(mode, focus, D):
    from gnnkeras_amd import CompositeGraphObject
    from gnnkeras_amd.Models.CompositeGNN import CompositeGNNnodeBased, CompositeGNNgraphBased
    from gnnkeras_amd.Models.CompositeLGNN import CompositeLGNN
    from gnnkeras_amd.Sequencers.GraphSequencers import CompositeMultiGraphSequencer
    CC = {'n': CompositeGNNnodeBased, 'g': CompositeGNNgraphBased}[focus]
    rng = np.random.default_rng(41)
    dims, A, T = (4, 2, 3), 2, 2

    def cg(n, e):
        pairs = set()
        while len(pairs) < e:
            a, b = rng.integers(0, n, 2)
            if a != b: pairs.add((int(a), int(b)))
        ids = np.array(sorted(pairs), dtype=float)
        types = rng.integers(0, 3, n); types[:3] = [0, 1, 2]
        tm = np.zeros((n, 3), bool); tm[np.arange(n), types] = True
        om = rng.random(n) < 0.8 if focus == 'n' else np.ones(n, bool)
        tg = np.zeros((int(om.sum()) if focus == 'n' else 1, T)); tg[np.arange(len(tg)), rng.integers(0, T, len(tg))] = 1
        return CompositeGraphObject(nodes=rng.normal(size=(n, 4)), arcs=np.concatenate([ids, rng.normal(size=(e, A))], 1),
                                    targets=tg, type_mask=tm, dim_node_label=dims, focus=focus,
                                    aggregation_mode='composite_average', **(dict(output_mask=om) if focus == 'n' else {}))
    seq = CompositeMultiGraphSequencer([cg(50, 160), cg(40, 120)], focus, 'composite_average', 2, shuffle=False)
    x, y, sw = seq[0]
    gnns, layers = [], []
    for layer in range(2):
        inp, lay = get_inout_dims('state', dims, A, T, focus, D, layer=layer, get_state=True, get_output=True)
        ns = [MLP(i, lay, 'tanh', 'lecun_normal', 'lecun_normal', rng=60 + t + 10 * layer, batch_normalization=True)
              for t, i in enumerate(inp)]
        for n in ns: n.set_weights([a * 0.5 if a.ndim == 2 else a for a in n.get_weights()])
        inp, lay = get_inout_dims('output', dims, A, T, focus, D, layer=layer, get_state=True, get_output=True)
        no = MLP(inp[0], lay, 'softmax', 'glorot_normal', 'glorot_normal', rng=80 + layer)
        gnns.append(CC(ns, no, D, 3, 0.0))
        layers.append(dict(net_state=[n.spec() for n in ns], net_output=no.spec(), state_vect_dim=D, max_iteration=3,
                           state_threshold=0.0))
    lg = CompositeLGNN(gnns, True, True)
    lg.compile(optimizer=SGD(0.0), loss='categorical_crossentropy', training_mode=mode, average_st_grads=False)
    N = x[0].shape[0]
    s0s = [rng.normal(0, 0.1, (N, D)).astype(np.float32) if D else None for _ in range(2)]
    nodes, arcs, dnl, tmask, sm, om_, cas, adj, an, ng = x
    mask = np.logical_and(_np(sm).reshape(-1), _np(om_).reshape(-1))
    want = torch_train.composite_lgnn_train_step(
        _np(nodes), _np(arcs), _np(dnl).reshape(-1), _np(tmask).reshape(3, -1), [_triple(c) for c in cas], _triple(adj),
        _triple(an), _triple(ng), mask, layers=layers, get_state=True, get_output=True, focus=focus, state0s=s0s, y=_np(y),
        sample_weight=_np(sw), loss='categorical_crossentropy', training_mode=mode)
    logs = lg.train_step((x, y, sw), state0=[None if s is None else torch.from_numpy(s).cuda() for s in s0s], apply=False)
    assert logs['k'] == want['k']
    assert abs(float(logs['loss']) - want['loss']) <= 1e-5 * max(1.0, abs(want['loss']))
    rows_t = _np(tmask).reshape(3, -1).sum(1)
    nets_ = []
    for li, tp in enumerate(lg._last_tapes):
        per_type = split_by_network(tp.gs, want['grads'][li][0])
        nets_ += [(f'layer{li}.state{t}', g_, per_type[t], None, int(rows_t[t])) for t, g_ in enumerate(tp.gs)]
        nets_ += [(f'layer{li}.output', tp.go, want['grads'][li][1], None, N)]
    check_network_grads('', nets_)


def test_inference_after_training_matches_oracle(mutag_graphs):
    """After a few optimisation steps every weight and BatchNormalization moving statistic is non-trivial: the
    inference-mode device loop (moving statistics folded into the fused kernel) must still equal the oracle."""
    from oracle.harness import oracle_loop
    gs = [g.copy() for g in mutag_graphs[:64]]
    for g in gs: g.setAggregation('average')
    seq = MultiGraphSequencer(gs, 'g', 'average', 32, shuffle=False)
    for d, it in [(0, 5), (16, 8)]:
        ns, no = nets('g', d, True, scale=1.0)
        model = GNNgraphBased(ns, no, d, it, 0.01)
        model.compile(optimizer=Adam(0.01), loss='categorical_crossentropy', metrics=['accuracy'])
        for _ in range(3):
            for i in range(len(seq)): model.train_step(seq[i], seed=1)
        w = model.net_state.get_weights()
        assert np.abs(w[2]).max() > 1e-3 and np.abs(w[3] - 1).max() > 1e-3       # moving stats really moved
        x = seq[1][0]
        s0 = np.random.default_rng(0).normal(0, .1, (x[0].shape[0], d)).astype(np.float32) if d else None
        k64, st64, o64 = oracle_loop(model, x, s0, np.float64)
        k, st, o = model.Loop(*model.process_inputs(x), state0=None if s0 is None else torch.from_numpy(s0).cuda())
        assert float(k) == float(k64)
        assert rel_err(st.cpu().numpy(), st64) <= 1e-5 and rel_err(o.cpu().numpy(), o64) <= 1e-5


@pytest.mark.parametrize('M,widths,H,act', [(70001, (64, 14, 64, 14, 3), 64, 'selu'), (131072, (64,), 64, 'linear'),
                                            (65537, (20, 33), 48, 'tanh'), (100000, (64, 64, 64), 37, 'relu')])
def test_dense_entry_large_batches(M, widths, H, act):
    """gnn_dense (the segmented Dense of every un-fused path: GNN.py:231-234 concat + Dense without the concat) on batches
    in the throughput regime (k_segdense<1>, > 16 384 rows): row-index lists, scattered output rows, ragged last tile, H < 64,
    against float64."""
    import ctypes as C
    from gnnkeras_amd import _native as nat
    from gnnkeras_amd.Models.training import _Prim
    rng = np.random.default_rng(M % 97)
    dev = torch.device('cuda', 0)
    p = _Prim(dev)
    K = sum(widths)
    W = torch.from_numpy(rng.normal(0, 0.3, (K, H)).astype(np.float32)).to(dev)
    b = torch.from_numpy(rng.normal(0, 0.3, H).astype(np.float32)).to(dev)
    segs, cols = [], []
    for i, w in enumerate(widths):
        rows = M + (1000 if i % 2 else 0)
        x = torch.from_numpy(rng.normal(0, 1, (rows, w + (i % 3))).astype(np.float32)).to(dev)[:, :w]     # leading dimension > width
        ridx = torch.from_numpy(rng.permutation(rows)[:M].astype(np.int32)).to(dev) if i % 2 else None
        segs.append((x, ridx))
        cols.append(x.double() if ridx is None else x.double()[ridx.long()])
    out_idx = torch.from_numpy(rng.permutation(M + 50)[:M].astype(np.int32)).to(dev) if len(widths) > 2 else None
    Y = torch.zeros((M + 50 if out_idx is not None else M, H), dtype=torch.float32, device=dev)
    p.dense(segs, W, H, b, nat.ACTIVATIONS[act], Y, out_rowidx=out_idx)
    z = torch.cat(cols, dim=1) @ W.double() + b.double()
    want = {'selu': torch.nn.functional.selu, 'linear': lambda t: t, 'tanh': torch.tanh, 'relu': torch.relu}[act](z)
    got = Y if out_idx is None else Y[out_idx.long()]
    assert float((got.double() - want).abs().max() / want.abs().max()) <= 1e-5
