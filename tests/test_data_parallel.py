"""Data-parallel training / batch-parallel inference on CPU: world_size-2 `gloo` processes run the product's orchestration
(`gnnkeras_amd/data_parallel.py` + the DP hooks of `Models/training.py`: shards of whole graphs, exact combination of the
BatchNormalization statistics, all-reduced convergence flag, P / q and gradient sums, loss normalisation) with the device
primitives replaced by a NumPy stand-in, and must reproduce the SINGLE-PROCESS step on the whole batch.  The stand-in is test
code: it restates what each `gnn_*` training primitive computes (include/gnnloop.h) in float64 NumPy; the product calls
libgnnloop.so and has no CPU path.  The single-process stand-in run is itself checked against the torch-autograd oracle
(oracle/torch_train.py), so the chain is: autograd oracle == stand-in (1 process) == stand-in (2 processes, sharded)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from gnnkeras_amd import GraphObject
from gnnkeras_amd import _native as nat
from gnnkeras_amd.Models.MLP import MLP, get_inout_dims, BN_EPSILON
from gnnkeras_amd.Models.GNN import GNNgraphBased, GNNnodeBased
from gnnkeras_amd.Models import training as TR
from gnnkeras_amd.Sequencers.GraphSequencers import MultiGraphSequencer
from oracle import torch_train
from oracle.harness import _np, _triple, rel_err

ACT_NAME = {v: k for k, v in nat.ACTIVATIONS.items() if k is not None}
SELU_SCALE, SELU_ALPHA = 1.0507009873554805, 1.6732632423543772


def _act(a, z):
    n = ACT_NAME[a]
    if n == 'linear': return z
    if n == 'relu': return np.maximum(z, 0)
    if n == 'selu': return SELU_SCALE * np.where(z > 0, z, SELU_ALPHA * (np.exp(np.minimum(z, 0)) - 1))
    if n == 'tanh': return np.tanh(z)
    if n == 'sigmoid': return 1 / (1 + np.exp(-z))
    if n == 'softmax':
        e = np.exp(z - z.max(1, keepdims=True)); return e / e.sum(1, keepdims=True)
    raise NotImplementedError(n)


def _act_grad(a, G, Y):
    n = ACT_NAME[a]
    if n == 'linear': return G
    if n == 'relu': return G * (Y > 0)
    if n == 'selu': return G * np.where(Y > 0, SELU_SCALE, Y + SELU_SCALE * SELU_ALPHA)
    if n == 'tanh': return G * (1 - Y * Y)
    if n == 'sigmoid': return G * Y * (1 - Y)
    if n == 'softmax': return Y * (G - (G * Y).sum(1, keepdims=True))
    raise NotImplementedError(n)


class NumpyPrim:
    """Stand-in for `training._Prim`: same methods, CPU torch tensors in / out, float64 NumPy arithmetic inside."""

    def __init__(self, device):
        self.dev = torch.device('cpu')

    def stream(self): return None
    def new(self, *shape): return torch.zeros(shape, dtype=torch.float32)
    def zeros(self, *shape): return torch.zeros(shape, dtype=torch.float32)

    @staticmethod
    def _rows(x, ridx, M=None):
        a = x.detach().numpy().astype(np.float64)
        if ridx is not None: return a[ridx.numpy().astype(np.int64)]
        return a if M is None else a[:M]

    @staticmethod
    def _put(out, arr):
        out.copy_(torch.from_numpy(np.ascontiguousarray(arr)).to(torch.float32))

    def dense(self, segs, W, H, bias, act, Y, wrows=None, out_rowidx=None, center=None):
        Wn = W.detach().numpy().astype(np.float64)
        cen = None if center is None else center.detach().numpy().astype(np.float64)
        z, off = 0.0, 0
        for i, (x, ridx) in enumerate(segs):
            w = x.shape[1]
            r0 = off if wrows is None else wrows[i]
            z = z + (self._rows(x, ridx) - (0.0 if cen is None else cen[r0:r0 + w])) @ Wn[r0:r0 + w, :H]
            off += w
        if bias is not None: z = z + bias.detach().numpy().astype(np.float64)[:H]
        y = _act(act, z)
        if out_rowidx is None: self._put(Y[:, :H], y)
        else: Y[out_rowidx.long(), :H] = torch.from_numpy(y).to(torch.float32)
        return Y

    def aggregate(self, csr, X, F, out):
        rp, src = csr['rowptr'].numpy().astype(np.int64), csr['src'].numpy().astype(np.int64)
        dst = np.repeat(np.arange(len(rp) - 1), np.diff(rp))
        w = np.ones(len(src)) if csr['w'] is None else csr['w'].numpy().astype(np.float64)
        if csr['row_scale'] is not None: w = w * csr['row_scale'].numpy().astype(np.float64)[dst]
        res = np.zeros((len(rp) - 1, F))
        np.add.at(res, dst, w[:, None] * X.detach().numpy().astype(np.float64)[src, :F])
        self._put(out[:, :F], res)
        return out

    def fold(self, W, b, bn, mean, var, Wf, bf, centred=False):
        Wn, bn_ = W.numpy().astype(np.float64), b.numpy().astype(np.float64)
        if bn is None:
            self._put(Wf, Wn); self._put(bf, bn_); return
        a = bn[0].numpy().astype(np.float64) / np.sqrt(var.numpy().astype(np.float64) + BN_EPSILON)
        c = bn[1].numpy().astype(np.float64) - (0.0 if centred else mean.numpy().astype(np.float64) * a)
        self._put(Wf, a[:, None] * Wn); self._put(bf, bn_ + c @ Wn)

    def colstats(self, x, ridx, M, mean, var):
        r = self._rows(x, ridx, M)
        self._put(mean, r.mean(0)); self._put(var, r.var(0))

    def dense_grad(self, x, ridx, dZ, M, P, q, accumulate, center=None):
        X, D = self._rows(x, ridx, M), dZ.numpy().astype(np.float64)[:M]
        if center is not None: X = X - center.detach().numpy().astype(np.float64)
        Pn, qn = X.T @ D, D.sum(0)
        self._put(P, Pn + (P.numpy() if accumulate else 0))
        if q is not None: self._put(q, qn + (q.numpy() if accumulate else 0))

    def act_grad(self, G, Y, dZ, act):
        self._put(dZ, _act_grad(act, G.numpy().astype(np.float64), Y.numpy().astype(np.float64)))
        return dZ

    def first_layer_param_grads(self, P, q, W, bn, mean, var, M, dW, db, dgamma, dbeta, m1, m2, accumulate, centered=False):
        Pn, qn, Wn = P.numpy().astype(np.float64), q.numpy().astype(np.float64), W.numpy().astype(np.float64)
        K = Wn.shape[0]
        a, c, rstd, mu = np.ones(K), np.zeros(K), np.ones(K), np.zeros(K)
        if bn is not None:
            rstd = 1 / np.sqrt(var.numpy().astype(np.float64) + BN_EPSILON); mu = mean.numpy().astype(np.float64)
            a = bn[0].numpy().astype(np.float64) * rstd; c = bn[1].numpy().astype(np.float64) - (0.0 if centered else mu * a)
        add = lambda t, v: self._put(t, v + (t.numpy() if accumulate else 0))
        add(dW, a[:, None] * Pn + c[:, None] * qn[None, :]); add(db, qn)
        if bn is not None:
            S1, S2 = Wn @ qn, (Wn * Pn).sum(1)
            dg = rstd * (S2 - (0.0 if centered else mu * S1))
            add(dgamma, dg); add(dbeta, S1)
            self._put(m1, S1 / M); self._put(m2, dg / M)

    def bn_input_grad(self, dy, x, ridx, M, k0, bn, mean, var, m1, m2, dx):
        v = dy.numpy().astype(np.float64)
        if bn is not None:
            w = v.shape[1]
            sl = slice(k0, k0 + w)
            rstd = 1 / np.sqrt(var.numpy().astype(np.float64)[sl] + BN_EPSILON)
            xhat = (self._rows(x, ridx, M) - mean.numpy().astype(np.float64)[sl]) * rstd
            v = bn[0].numpy().astype(np.float64)[sl] * rstd * (v - m1.numpy().astype(np.float64)[sl] - xhat * m2.numpy().astype(np.float64)[sl])
        self._put(dx, v)

    def scatter_add_rows(self, D, idx, G):
        g = G.numpy().astype(np.float64)
        np.add.at(g, idx.numpy().astype(np.int64), D.numpy().astype(np.float64))
        self._put(G, g)

    def axpby(self, a, x, b, y, out):
        self._put(out, a * x.numpy().astype(np.float64) + b * y.numpy().astype(np.float64))

    def converged_gated(self, state, state_old, thr, gate, flag, k_dev, k_val):
        if gate is not None and int(gate[0]) == 0: return
        s = state.numpy().astype(np.float64)
        o = np.ones_like(s) if state_old is None else state_old.numpy().astype(np.float64)
        if np.any(np.sqrt(((s - o) ** 2).sum(1)) > thr * np.sqrt((o ** 2).sum(1))): flag[0] = 1
        if k_dev is not None: k_dev.fill_(float(k_val))

    def loss_grad(self, kind, y, y_pred, sw, dpred, loss_rows):
        yt = torch.from_numpy(y.numpy().astype(np.float64))
        p = torch.from_numpy(y_pred.numpy().astype(np.float64)).requires_grad_()
        w = torch.ones(yt.shape[0], dtype=torch.float64) if sw is None else torch.from_numpy(sw.numpy().astype(np.float64))
        name = {0: 'cce', 2: 'mse'}[kind]
        eps = 1e-7
        if name == 'cce':
            pn = p / p.sum(-1, keepdim=True)
            rows = -(yt * torch.log(pn.clamp(eps, 1 - eps))).sum(-1) * w
        else:
            rows = ((p - yt) ** 2).mean(-1) * w
        (rows.sum() / yt.shape[0]).backward()
        self._put(dpred, p.grad.numpy()); self._put(loss_rows[:yt.shape[0]], rows.detach().numpy())


def _graphs(rng, n_graphs, L=4, A=2, focus='g'):
    out = []
    for _ in range(n_graphs):
        n = int(rng.integers(4, 9))
        src = np.concatenate([np.arange(n), rng.integers(0, n, 2 * n)]); dst = np.concatenate([(np.arange(n) + 1) % n, rng.integers(0, n, 2 * n)])
        ids = np.unique(np.stack([src, dst], 1)[src != dst], axis=0)
        arcs = np.concatenate([ids, rng.normal(size=(len(ids), A)).round(2)], axis=1)
        if focus == 'g':
            out.append(GraphObject(rng.normal(size=(n, L)), arcs, np.eye(2)[rng.integers(0, 2, 1)], focus='g', sample_weight=rng.uniform(0.5, 1.5)))
        else:
            om = np.ones(n, bool); om[rng.integers(0, n)] = False
            out.append(GraphObject(rng.normal(size=(n, L)), arcs, np.eye(2)[rng.integers(0, 2, int(om.sum()))], focus='n', output_mask=om,
                                   sample_weight=rng.uniform(0.5, 1.5, int(om.sum()))))
    return out


def _problem(focus='g', hidden=None, thr=0.0):
    rng = np.random.default_rng(7)
    L, A, d = 4, 2, 5
    if focus == 'g0': focus, d = 'g', 0                     # state_vect_dim = 0: the state is the label matrix, no random state_0
    gl = _graphs(rng, 7, L, A, focus)
    inp, lay = get_inout_dims('state', L, A, 2, focus, d, hidden_units=hidden)
    ns = MLP(inp[0], lay, 'tanh', 'lecun_normal', 'lecun_normal', rng=0, device='cpu')
    ns.set_weights([w * 0.6 if w.ndim == 2 else (w + 0.1 * rng.normal(size=w.shape)).astype(np.float32) for w in ns.get_weights()])
    inp, lay = get_inout_dims('output', L, A, 2, focus, d, hidden_units=[6])
    no = MLP(inp[0], lay, ['tanh', 'softmax'], 'glorot_normal', 'glorot_normal', rng=1, device='cpu')
    model = (GNNgraphBased if focus == 'g' else GNNnodeBased)(ns, no, d, 4, thr)
    # inference on CPU (predict / evaluate in these tests): the oracle stands in for the HIP loop, as NumpyPrim does for training
    from oracle.harness import oracle_loop
    model.call = lambda x, training=False, **kw: torch.from_numpy(np.asarray(oracle_loop(model, x, None, np.float32)[2], dtype=np.float32))
    model.compile(optimizer=_SGD(0.05), loss='categorical_crossentropy', metrics=['accuracy'])
    seq = MultiGraphSequencer(gl, focus, 'average', 7, shuffle=False, device='cpu')
    N = seq[0][0][0].shape[0]
    s0 = rng.normal(0, 0.1, (N, d)).astype(np.float32) if d else None
    return model, seq, s0, gl


class _SGD:
    """Host optimizer for the CPU tests (the product's Adam / SGD are device kernels)."""
    def __init__(self, lr): self.lr = lr
    def apply_gradients(self, gv):
        for g, v in gv: v.sub_(self.lr * g)


@pytest.fixture(autouse=True)
def _cpu_stand_in(monkeypatch):
    monkeypatch.setattr(TR.LoopTrainer, 'prim_cls', NumpyPrim)
    monkeypatch.setattr(TR.LoopTrainer, 'use_native_step', False)
    monkeypatch.setattr(nat, 'require_device', lambda t, name: None)


def _single_step(model, seq, s0, apply=False):
    tr = TR.LoopTrainer(model)
    x, y, sw = seq[0]
    res = tr.train_step(x, y, sw, state0=None if s0 is None else torch.from_numpy(s0), apply=apply)
    grads = [g.numpy().copy() for g in tr.gs.gradients() + tr.go.gradients()]
    return res, grads


@pytest.mark.parametrize('focus,hidden,thr', [('g', None, 0.0), ('n', [7], 0.0), ('g', None, 1.0)])
def test_stand_in_step_matches_the_autograd_oracle(focus, hidden, thr):
    """The NumPy stand-in driven by the product's orchestration == torch autograd (float64) on the same batch."""
    model, seq, s0, _ = _problem(focus, hidden, thr)
    x, y, sw = seq[0]
    nodes, arcs, _, sm, om, adj, an, ng = x
    mask = np.logical_and(_np(sm).reshape(-1), _np(om).reshape(-1))
    want = torch_train.train_step(_np(nodes), _np(arcs), _triple(adj), _triple(an), _triple(ng), mask, net_state=model.net_state.spec(),
                                  net_output=model.net_output.spec(), state_vect_dim=5, max_iteration=4, state_threshold=thr, focus=focus,
                                  state0=s0, y=_np(y), sample_weight=_np(sw), loss='categorical_crossentropy')
    res, grads = _single_step(model, seq, s0)
    if thr > 0: assert 0 < want['k'] < 4
    assert res['k'] == want['k'] and abs(float(res['loss']) - want['loss']) < 1e-6
    for g, r in zip(grads, want['grads_state'] + want['grads_output']):
        assert np.max(np.abs(g - r)) <= 2e-6 * max(1.0, np.max(np.abs(r)))


def _dp_worker(rank, world, port, focus, hidden, thr, mode, q):
    os.environ['MASTER_ADDR'], os.environ['MASTER_PORT'] = '127.0.0.1', str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        TR.LoopTrainer.prim_cls = NumpyPrim
        TR.LoopTrainer.use_native_step = False
        nat.require_device = lambda t, name: None
        from gnnkeras_amd.data_parallel import DataParallel
        model, seq, s0, gl = _problem(focus, hidden, thr)
        dpm = DataParallel(model, exact=(mode != 'replica'))
        if mode == 'replica':
            # "replicas": the single-process step on the own shard, then one weighted all-reduce
            shard = dpm.shard(seq, 0)
            sizes = [g.nodes.shape[0] for g in gl]
            lo, hi = len(gl) * rank // world, len(gl) * (rank + 1) // world
            n0 = sum(sizes[:lo]); n1 = n0 + sum(sizes[lo:hi])
            st = None if s0 is None else torch.from_numpy(s0[n0:n1])
            own = TR.LoopTrainer(model)
            before = [w.copy() for w in model.net_state.get_weights() + model.net_output.get_weights()]
            r_own = own.train_step(shard[0], shard[1], shard[2], state0=st, apply=False)
            g_own = [g.numpy().copy() for g in own.gs.gradients() + own.go.gradients()]
            mv_own = [w.copy() for w in model.net_state.get_weights()[2:4] + model.net_output.get_weights()[2:4]]
            model.net_state.set_weights(before[:len(model.net_state.get_weights())]); model.net_output.set_weights(before[len(model.net_state.get_weights()):])
            res = dpm.train_step(shard, state0=st, apply=False)
            tr = dpm._trainer
            grads = [g.numpy().copy() for g in tr.gs.gradients() + tr.go.gradients()]
            moving = [w.copy() for w in model.net_state.get_weights()[2:4] + model.net_output.get_weights()[2:4]]
            q.put((rank, res['k'], float(res['loss']), grads, moving, float(shard[1].shape[0]), r_own['k'], float(r_own['loss']), g_own, mv_own))
        elif mode == 'step':
            shard = dpm.shard(seq, 0)
            sizes = [g.nodes.shape[0] for g in gl]
            lo, hi = len(gl) * rank // world, len(gl) * (rank + 1) // world
            n0 = sum(sizes[:lo]); n1 = n0 + sum(sizes[lo:hi])
            assert shard[0][0].shape[0] == n1 - n0
            res = dpm.train_step(shard, state0=None if s0 is None else torch.from_numpy(s0[n0:n1]), apply=False)
            tr = dpm._trainer
            grads = [g.numpy().copy() for g in tr.gs.gradients() + tr.go.gradients()]
            moving = [w.copy() for w in model.net_state.get_weights()[2:4] + model.net_output.get_weights()[2:4]]
            q.put((rank, res['k'], float(res['loss']), float(res['accuracy']), grads, moving, res['y_pred'].numpy()))
        else:
            hist = dpm.fit(seq, epochs=3, verbose=0)
            weights = model.net_state.get_weights() + model.net_output.get_weights()
            q.put((rank, dict(hist), weights, dpm.predict(seq), dpm.evaluate(seq, return_dict=True)))
    finally:
        dist.destroy_process_group()


def _run(world, *args):
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 29500 + (os.getpid() * 7 + sum(map(ord, str(args))) + 131 * world) % 1000
    procs = [ctx.Process(target=_dp_worker, args=(r, world, port) + args + (q,)) for r in range(world)]
    for p in procs: p.start()
    res = sorted([q.get(timeout=240) for _ in procs], key=lambda r: r[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    return res


@pytest.mark.parametrize('world', [2, 3])
@pytest.mark.parametrize('focus,hidden,thr', [('g', None, 0.0), ('n', [7], 0.0), ('g', None, 1.0)])
def test_data_parallel_step_reproduces_the_single_process_step(world, focus, hidden, thr):
    """Every rank trains on its shard of whole graphs; gradients, loss, k, metric and BN moving statistics are those of the
    single-process step on the whole batch (<= 1e-6), on every rank."""
    model, seq, s0, _ = _problem(focus, hidden, thr)
    ref, grads = _single_step(model, seq, s0)
    ref_moving = model.net_state.get_weights()[2:4] + model.net_output.get_weights()[2:4]
    res = _run(world, focus, hidden, thr, 'step')
    y_pred = np.concatenate([r[6] for r in res])
    assert rel_err(y_pred, ref['y_pred'].numpy()) <= 1e-6
    for rank, k, loss, acc, g_r, moving, _ in res:
        assert k == ref['k'] and abs(loss - float(ref['loss'])) <= 1e-6
        for a, b in zip(g_r, grads):
            assert np.max(np.abs(a - b)) <= 1e-6 * max(1.0, np.max(np.abs(b))), rank
        for a, b in zip(moving, ref_moving): assert np.max(np.abs(a - b)) <= 1e-6


def test_data_parallel_fit_predict_evaluate_match_the_single_process_calls():
    """fit() over 3 epochs (one batch, no shuffle) + predict() + evaluate(): weights, history and outputs of a 2-rank run equal
    the single-process ones; both ranks hold identical weights afterwards."""
    # (fit draws state_0 at random when state_vect_dim > 0; the 'g0' problem has state_vect_dim = 0: the state is the label matrix)
    res = _run(2, 'g0', None, 0.0, 'fit')
    (r0, h0, w0, p0, e0), (r1, h1, w1, p1, e1) = res
    for a, b in zip(w0, w1): assert np.array_equal(a, b)
    assert np.array_equal(p0, p1) and e0 == e1 and h0 == h1
    model, seq, s0, _ = _problem('g0', None, 0.0)
    hist = model.fit(seq, epochs=3, verbose=0)
    for a, b in zip(model.net_state.get_weights() + model.net_output.get_weights(), w0):
        assert np.max(np.abs(a - b)) <= 1e-6 * max(1.0, np.max(np.abs(a)))
    assert np.allclose(hist['loss'], h0['loss'], atol=1e-6) and np.allclose(hist['accuracy'], h0['accuracy'], atol=1e-6)


@pytest.mark.parametrize('world', [2, 3])
def test_replica_mode_averages_the_single_process_steps_of_the_shards(world):
    """`DataParallel(model, exact=False)`: every rank runs the ordinary single-process step on its shard (own batch statistics, own k)
    and ONE all-reduce combines them - gradients and loss weighted by the shards' target rows, moving statistics averaged, k = the
    maximum: checked against the same combination formed by hand from the per-rank single-process results; identical on every rank."""
    res = _run(world, 'g', None, 0.0, 'replica')
    rows = np.array([r[5] for r in res]); wts = rows / rows.sum()
    want_g = [sum(w * r[8][i] for w, r in zip(wts, res)) for i in range(len(res[0][8]))]
    want_loss = float(sum(w * r[7] for w, r in zip(wts, res)))
    want_mv = [sum(r[9][i] for r in res) / world for i in range(len(res[0][9]))]
    for rank, k, loss, grads, moving, _, k_own, _, _, _ in res:
        assert k == max(r[6] for r in res) and abs(loss - want_loss) <= 1e-6
        for a, b in zip(grads, want_g): assert np.max(np.abs(a - b)) <= 1e-6 * max(1.0, np.max(np.abs(b)))
        for a, b in zip(moving, want_mv): assert np.max(np.abs(a - b)) <= 1e-6
    for a, b in zip(res[0][3], res[-1][3]): assert np.array_equal(a, b)


# ----------------------------------------------------------------------------------------------------------------------
# heterogeneous models (reference CompositeGNN.py:275-304) in the exact mode - round 5
# ----------------------------------------------------------------------------------------------------------------------
def _composite_problem():
    """Five typed graphs (3 node types; type 2 lives in the LAST graph only: with 2 or 3 ranks some shards have no node of it), one
    state network per type with BatchNormalization, graph focus."""
    from gnnkeras_amd import CompositeGraphObject
    from gnnkeras_amd.Models.CompositeGNN import CompositeGNNgraphBased
    from gnnkeras_amd.Sequencers.GraphSequencers import CompositeMultiGraphSequencer
    rng = np.random.default_rng(11)
    dims, A, d = (3, 2, 2), 2, 5
    gl = []
    for gi in range(5):
        n = int(rng.integers(5, 9))
        src = np.concatenate([np.arange(n), rng.integers(0, n, 2 * n)]); dst = np.concatenate([(np.arange(n) + 1) % n, rng.integers(0, n, 2 * n)])
        ids = np.unique(np.stack([src, dst], 1)[src != dst], axis=0)
        arcs = np.concatenate([ids, rng.normal(size=(len(ids), A)).round(2)], axis=1)
        types = rng.integers(0, 2, n); types[:2] = [0, 1]
        if gi == 4: types[2:4] = 2
        tm = np.zeros((n, 3), bool); tm[np.arange(n), types] = True
        gl.append(CompositeGraphObject(nodes=rng.normal(size=(n, 3)), arcs=arcs, targets=np.eye(2)[rng.integers(0, 2, 1)], type_mask=tm,
                                       dim_node_label=dims, focus='g', aggregation_mode='average', sample_weight=rng.uniform(0.5, 1.5)))
    inp, lay = get_inout_dims('state', dims, A, 2, 'g', d)
    ns = [MLP(i, lay, 'tanh', 'lecun_normal', 'lecun_normal', rng=t, device='cpu') for t, i in enumerate(inp)]
    for n_ in ns: n_.set_weights([w * 0.6 if w.ndim == 2 else (w + 0.1 * rng.normal(size=w.shape)).astype(np.float32) for w in n_.get_weights()])
    inp, lay = get_inout_dims('output', dims, A, 2, 'g', d)
    no = MLP(inp[0], lay, 'softmax', 'glorot_normal', 'glorot_normal', rng=9, device='cpu')
    model = CompositeGNNgraphBased(ns, no, d, 4, 0.0)
    model.compile(optimizer=_SGD(0.05), loss='categorical_crossentropy', metrics=['accuracy'])
    seq = CompositeMultiGraphSequencer(gl, 'g', 'average', 5, shuffle=False, device='cpu')
    s0 = rng.normal(0, 0.1, (seq[0][0][0].shape[0], d)).astype(np.float32)
    return model, seq, s0, gl


def _composite_worker(rank, world, port, q):
    os.environ['MASTER_ADDR'], os.environ['MASTER_PORT'] = '127.0.0.1', str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        TR.LoopTrainer.prim_cls = NumpyPrim
        TR.LoopTrainer.use_native_step = False
        nat.require_device = lambda t, name: None
        from gnnkeras_amd.data_parallel import DataParallel
        model, seq, s0, gl = _composite_problem()
        dpm = DataParallel(model, exact=True)
        shard = dpm.shard(seq, 0)
        sizes = [g.nodes.shape[0] for g in gl]
        lo, hi = len(gl) * rank // world, len(gl) * (rank + 1) // world
        n0 = sum(sizes[:lo]); n1 = n0 + sum(sizes[lo:hi])
        res = dpm.train_step(shard, state0=torch.from_numpy(s0[n0:n1]), apply=False)
        tr = dpm._trainer
        grads = [g.numpy().copy() for gs_ in tr.gs for g in gs_.gradients()] + [g.numpy().copy() for g in tr.go.gradients()]
        moving = [w.copy() for n_ in model.net_state for w in n_.get_weights()[2:4]] + [w.copy() for w in model.net_output.get_weights()[2:4]]
        q.put((rank, res['k'], float(res['loss']), grads, moving))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('world', [2, 3])
def test_exact_data_parallel_step_of_a_heterogeneous_model(world):
    """`DataParallel(composite model)` (exact mode; refused through round 4): every rank trains on its shard of whole typed graphs, the
    statistics of every TYPE's network are combined over the ranks with the shard's row count of that type as weight - a rank without a
    single node of a type takes part with weight zero - and gradients, loss, k and the moving statistics are those of the
    single-process step on the whole batch, on every rank."""
    model, seq, s0, _ = _composite_problem()
    tr = TR.LoopTrainer(model)
    x, y, sw = seq[0]
    ref = tr.train_step(x, y, sw, state0=torch.from_numpy(s0), apply=False)
    ref_g = [g.numpy().copy() for gs_ in tr.gs for g in gs_.gradients()] + [g.numpy().copy() for g in tr.go.gradients()]
    ref_mv = [w.copy() for n_ in model.net_state for w in n_.get_weights()[2:4]] + [w.copy() for w in model.net_output.get_weights()[2:4]]
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 29500 + (os.getpid() * 11 + 977 * world) % 1000
    procs = [ctx.Process(target=_composite_worker, args=(r, world, port, q)) for r in range(world)]
    for p_ in procs: p_.start()
    res = sorted([q.get(timeout=240) for _ in procs], key=lambda r: r[0])
    for p_ in procs:
        p_.join(60)
        assert p_.exitcode == 0
    for rank, k, loss, grads, moving in res:
        assert k == ref['k'] and abs(loss - float(ref['loss'])) <= 1e-6
        assert len(grads) == len(ref_g)
        for a, b in zip(grads, ref_g): assert np.max(np.abs(a - b)) <= 1e-6 * max(1.0, np.max(np.abs(b))), rank
        for a, b in zip(moving, ref_mv): assert np.max(np.abs(a - b)) <= 1e-6, rank
