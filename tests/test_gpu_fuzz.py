"""Seeded fuzzing of the device loop against the oracle: random graph shapes, label / state widths, aggregation modes,
foci, masks, activations, BatchNormalization on / off, hidden layers, thresholds — homogeneous and composite."""
import numpy as np
import pytest
import torch

from gnnkeras_amd import _native as nat
from gnnkeras_amd import GraphObject, CompositeGraphObject
from gnnkeras_amd.Models.MLP import MLP, get_inout_dims
from gnnkeras_amd.Models.GNN import GNNnodeBased, GNNarcBased, GNNgraphBased
from gnnkeras_amd.Models.CompositeGNN import CompositeGNNnodeBased, CompositeGNNarcBased, CompositeGNNgraphBased
from gnnkeras_amd.Sequencers.GraphSequencers import MultiGraphSequencer, CompositeMultiGraphSequencer
from oracle.harness import oracle_loop, oracle_composite_loop, rel_err

import os

pytestmark = pytest.mark.gpu
FUZZ_SCALE = int(os.environ.get('GNN_FUZZ_SCALE', '1'))      # GNN_FUZZ_SCALE=10 runs ten times as many seeds (offline soak)
# every way the iteration can run: size-based default, un-fused kernels, and each fused-kernel generation pinned
PATHS = (0, nat.FLAG_UNFUSED, nat.FLAG_FUSED_GEN2, nat.FLAG_FUSED_GEN4, nat.FLAG_FUSED_GEN5)
ACTS = ['selu', 'tanh', 'relu', 'sigmoid', 'linear', 'elu', 'softplus']


def random_arcs(rng, n, e, A):
    e = min(e, n * (n - 1))
    pairs = set()
    while len(pairs) < e:
        a, b = rng.integers(0, n, 2)
        if a != b: pairs.add((int(a), int(b)))
    ids = np.array(sorted(pairs), dtype=float).reshape(-1, 2)
    return np.concatenate([ids, rng.normal(size=(len(ids), A))], axis=1)


def random_nets_bn(nets, rng):
    for n in nets:
        if n.batch_normalization:
            w = n.get_weights()
            d = len(w[0])
            w[0:4] = [rng.uniform(.6, 1.4, d).astype(np.float32), rng.normal(0, .2, d).astype(np.float32),
                      rng.normal(0, .2, d).astype(np.float32), rng.uniform(.5, 1.5, d).astype(np.float32)]
            n.set_weights(w)


@pytest.mark.parametrize('seed', range(40 * FUZZ_SCALE))
def test_fuzz_homogeneous(seed):
    rng = np.random.default_rng(1000 + seed)
    focus = ['n', 'a', 'g'][seed % 3]
    L, A, T = int(rng.integers(1, 20)), int(rng.integers(0, 5)), int(rng.integers(1, 5))
    d = int(rng.choice([0, 1, 2, 5, 8, 16, 24, 32, 40, 64, 70, 100, 128]))       # > 64: the wide fused kernel
    mode = ['sum', 'average', 'normalized'][int(rng.integers(0, 3))]
    bn = bool(rng.integers(0, 2))
    hidden = [int(rng.integers(3, 50))] if rng.random() < 0.3 else None
    graphs = []
    for _ in range(int(rng.integers(1, 5))):
        n = int(rng.integers(2, 120))
        # 'normalized' divides by the number of arcs: an arc-less graph raises ZeroDivisionError there, as in the reference
        arcs = random_arcs(rng, n, int(rng.integers(1 if mode == 'normalized' else 0, 4 * n)), A)
        cnt = {'n': n, 'a': len(arcs), 'g': n}[focus]
        om = rng.random(cnt) < 0.7 if focus != 'g' else np.ones(n, bool)
        sm = rng.random(cnt) < 0.8 if focus != 'g' else np.ones(n, bool)
        nt = int(om.sum()) if focus != 'g' else 1
        graphs.append(GraphObject(rng.normal(size=(n, L)), arcs, rng.normal(size=(nt, T)), focus=focus, set_mask=sm,
                                  output_mask=om, aggregation_mode=mode))
    seq = MultiGraphSequencer(graphs, focus, mode, len(graphs), shuffle=False)
    x = seq[0][0]
    inp, lay = get_inout_dims('state', L, A, T, focus, d, hidden_units=hidden)
    act = ACTS[int(rng.integers(0, len(ACTS)))]
    ns = MLP(inp[0], lay, act, 'lecun_normal', 'lecun_normal', rng=seed, batch_normalization=bn)
    ns.set_weights([w * 0.4 if w.ndim == 2 else w for w in ns.get_weights()])
    inp, lay = get_inout_dims('output', L, A, T, focus, d, hidden_units=[7] if rng.random() < 0.3 else None)
    no = MLP(inp[0], lay, ['tanh'] * (len(lay) - 1) + [['softmax', 'linear', 'sigmoid'][int(rng.integers(0, 3))]],
             'glorot_normal', 'glorot_normal', rng=seed + 1, batch_normalization=bn)
    random_nets_bn([ns, no], rng)
    thr = float(rng.choice([0.0, 0.0, 0.05, 0.3]))
    model = {'n': GNNnodeBased, 'a': GNNarcBased, 'g': GNNgraphBased}[focus](ns, no, d, int(rng.integers(0, 9)), thr)
    N = x[0].shape[0]
    s0 = rng.normal(0, 0.3, (N, d)).astype(np.float32) if d else None
    k64, st64, o64 = oracle_loop(model, x, s0, np.float64)
    k32, st32, o32 = oracle_loop(model, x, s0, np.float32)
    for flags in PATHS:
        model.native_flags = flags
        k, st, o = model.Loop(*model.process_inputs(x), state0=None if s0 is None else torch.from_numpy(s0).cuda())
        if float(k) == float(k64):
            assert rel_err(st.cpu().numpy(), st64) <= 2e-5 and rel_err(o.cpu().numpy(), o64) <= 2e-5, seed
        elif thr == 0.0:
            # threshold 0 stops only at an EXACT float32 fixed point (narrow states, saturating activations): which
            # iteration that happens at depends on the last bit of every sum, so the device may stop before float64
            # does; its state must then agree with the float64 run cut at the same iteration
            assert float(k) < float(k64), (seed, float(k), k64)
            model.max_iteration, keep = int(float(k)), model.max_iteration
            _, st_cut, o_cut = oracle_loop(model, x, s0, np.float64)
            model.max_iteration = keep
            assert rel_err(st.cpu().numpy(), st_cut) <= 2e-5 and rel_err(o.cpu().numpy(), o_cut) <= 2e-5, seed
        else:
            assert float(k32) != float(k64), (seed, float(k), k32, k64)    # a borderline predicate: the oracles disagree too
        assert np.isfinite(st.cpu().numpy()).all() and np.isfinite(o.cpu().numpy()).all()


@pytest.mark.parametrize('seed', range(16 * FUZZ_SCALE))
def test_fuzz_composite(seed):
    rng = np.random.default_rng(5000 + seed)
    focus = ['n', 'a', 'g'][seed % 3]
    Tt = int(rng.integers(2, 6))        # one type: the reference's get_inout_dims sizes the output net with labels (MLP.py:124)
    dims = tuple(int(v) for v in rng.integers(1, 7, Tt))
    Lw, A, T = max(dims), int(rng.integers(0, 4)), int(rng.integers(1, 4))
    D = int(rng.choice([1, 4, 8, 16, 33, 72]))
    mode = ['sum', 'average', 'normalized', 'composite_average'][int(rng.integers(0, 4))]
    graphs = []
    for _ in range(int(rng.integers(1, 4))):
        n = int(rng.integers(Tt + 1, 90))
        arcs = random_arcs(rng, n, int(rng.integers(1, 4 * n)), A)
        types = rng.integers(0, Tt, n); types[:Tt] = np.arange(Tt)
        tm = np.zeros((n, Tt), bool); tm[np.arange(n), types] = True
        cnt = {'n': n, 'a': len(arcs), 'g': n}[focus]
        om = rng.random(cnt) < 0.7 if focus != 'g' else np.ones(n, bool)
        nt = int(om.sum()) if focus != 'g' else 1
        graphs.append(CompositeGraphObject(rng.normal(size=(n, Lw)), arcs, rng.normal(size=(nt, T)), type_mask=tm,
                                           dim_node_label=dims, focus=focus, output_mask=om, aggregation_mode=mode))
    seq = CompositeMultiGraphSequencer(graphs, focus, mode, len(graphs), shuffle=False)
    x = seq[0][0]
    bn = bool(rng.integers(0, 2))
    inp, lay = get_inout_dims('state', dims, A, T, focus, D, hidden_units=[9] if rng.random() < 0.3 else None)
    ns = [MLP(i, lay, ACTS[int(rng.integers(0, len(ACTS)))], 'lecun_normal', 'lecun_normal', rng=seed + t, batch_normalization=bn)
          for t, i in enumerate(inp)]
    for n_ in ns: n_.set_weights([w * 0.4 if w.ndim == 2 else w for w in n_.get_weights()])
    inp, lay = get_inout_dims('output', dims, A, T, focus, D)
    no = MLP(inp[0], lay, 'softmax', 'glorot_normal', 'glorot_normal', rng=seed + 50, batch_normalization=bn)
    random_nets_bn(ns + [no], rng)
    model = {'n': CompositeGNNnodeBased, 'a': CompositeGNNarcBased, 'g': CompositeGNNgraphBased}[focus](
        ns, no, D, int(rng.integers(1, 8)), 0.0)
    N = x[0].shape[0]
    s0 = rng.normal(0, 0.3, (N, D)).astype(np.float32)
    k64, st64, o64 = oracle_composite_loop(model, x, s0, np.float64)
    for flags in PATHS:
        model.native_flags = flags
        k, st, o = model.Loop(*model.process_inputs(x), state0=torch.from_numpy(s0).cuda())
        assert float(k) == float(k64)
        assert rel_err(st.cpu().numpy(), st64) <= 2e-5 and rel_err(o.cpu().numpy(), o64) <= 2e-5, seed


@pytest.mark.parametrize('seed', range(24 * FUZZ_SCALE))
def test_fuzz_training_gradients(seed):
    """Random train_step configurations against the torch-autograd restatement (gradients, loss, moving statistics)."""
    from test_gpu_training import check_step, oracle_step
    rng = np.random.default_rng(9000 + seed)
    focus = ['g', 'n', 'a'][seed % 3]
    L, A, T = int(rng.integers(1, 12)), int(rng.integers(0, 4)), int(rng.integers(2, 5))
    d = int(rng.choice([0, 3, 8, 16, 32, 40]))
    mode = ['sum', 'average', 'normalized'][int(rng.integers(0, 3))]
    bn = bool(rng.integers(0, 2))
    loss, out_act = [('categorical_crossentropy', 'softmax'), ('mse', 'linear'), ('binary_crossentropy', 'sigmoid'),
                     ('mae', 'tanh')][int(rng.integers(0, 4))]
    graphs = []
    for _ in range(int(rng.integers(2, 6))):
        n = int(rng.integers(3, 60))
        arcs = random_arcs(rng, n, int(rng.integers(2, 3 * n)), A)
        cnt = {'n': n, 'a': len(arcs), 'g': n}[focus]
        om = rng.random(cnt) < 0.7 if focus != 'g' else np.ones(n, bool)
        om[0] = True
        sm = rng.random(cnt) < 0.8 if focus != 'g' else np.ones(n, bool)
        sm[0] = True
        nt = int(om.sum()) if focus != 'g' else 1
        t = rng.random((nt, T)); t = t / t.sum(1, keepdims=True)
        graphs.append(GraphObject(rng.normal(size=(n, L)), arcs, t, focus=focus, set_mask=sm, output_mask=om,
                                  sample_weight=rng.uniform(0.5, 1.5, nt), aggregation_mode=mode))
    seq = MultiGraphSequencer(graphs, focus, mode, len(graphs), shuffle=False)
    x, y, sw = seq[0]
    inp, lay = get_inout_dims('state', L, A, T, focus, d, hidden_units=[int(rng.integers(3, 30))] if rng.random() < 0.4 else None)
    ns = MLP(inp[0], lay, ['tanh', 'selu', 'sigmoid', 'softplus'][int(rng.integers(0, 4))], 'lecun_normal', 'lecun_normal',
             rng=seed, batch_normalization=bn)
    ns.set_weights([w * 0.3 if w.ndim == 2 else w for w in ns.get_weights()])
    inp, lay = get_inout_dims('output', L, A, T, focus, d, hidden_units=[6] if rng.random() < 0.4 else None)
    no = MLP(inp[0], lay, ['tanh'] * (len(lay) - 1) + [out_act], 'glorot_normal', 'glorot_normal', rng=seed + 1,
             batch_normalization=bn)
    for n_ in (ns, no):
        if bn:
            w = n_.get_weights()
            w[0] = rng.uniform(.7, 1.3, w[0].shape).astype(np.float32); w[1] = rng.normal(0, .2, w[1].shape).astype(np.float32)
            n_.set_weights(w)
    model = {'n': GNNnodeBased, 'a': GNNarcBased, 'g': GNNgraphBased}[focus](ns, no, d, int(rng.integers(1, 7)), 0.0)
    s0 = rng.normal(0, 0.2, (x[0].shape[0], d)).astype(np.float32) if d else None
    avg = bool(rng.integers(0, 2))
    k32, k64 = (oracle_step(model, x, y, sw, s0, loss, avg, dtype=dt)['k'] for dt in (torch.float32, torch.float64))
    if k32 != k64:      # a float32 state reached an exact fixed point (threshold 0) before float64 did: that depends on the
                        # last bit of every sum, so only the range of k is checked
        from gnnkeras_amd.Models.training import LoopTrainer, SGD
        model.compile(optimizer=SGD(0.0), loss=loss, average_st_grads=avg)
        res = LoopTrainer(model).train_step(x, y, sw, state0=None if s0 is None else torch.from_numpy(s0).cuda(), apply=False)
        assert min(k32, k64) <= res['k'] <= max(k32, k64)
        return
    check_step(model, x, y, sw, s0, loss=loss, avg=avg)
