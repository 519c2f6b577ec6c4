"""Round-5 tests: the training step (reference GNN/Models/GNN.py:277-306, CompositeGNN.py:275-304) held to the forward's standard.

  * a failed persistent BACKWARD launch can never reach the weights (ABI 7: the validity word, gated moving averages and optimizer), and
    its batch is trained on the general kernels one step late;
  * the thin output head of the large-graph path checks that `out_index` is the identity before it treats output rows as nodes;
  * small-graph training with labels far from zero (BatchNormalization first: reference MLP.py:67-70), both orchestrations;
  * `gnn_train_step` at the size and depth bench.py times it (BASELINE C4: 1 M nodes / 10 M arcs, d = 64, 10 iterations) against torch
    autograd in float64, per tensor.

Gradient bars are per tensor (see `BARS` in test_gpu_training.py), max-norm relative to the tensor's own largest entry."""
import os
import warnings

import numpy as np
import pytest
import torch

from gnnkeras_amd import _native as nat
from gnnkeras_amd import GraphObject
from gnnkeras_amd.Models.MLP import MLP, get_inout_dims
from gnnkeras_amd.Models.GNN import GNNnodeBased, GNNgraphBased
from gnnkeras_amd.Models.training import Adam, SGD, LoopTrainer
from gnnkeras_amd.Sequencers.GraphSequencers import MultiGraphSequencer
from gnnkeras_amd.synth import er_graph
from oracle.harness import rel_err
from test_gpu_training import prefetch_oracle

pytestmark = pytest.mark.gpu


def _weights(model):
    return [w.copy() for w in model.net_state.get_weights() + model.net_output.get_weights()]


def _set_weights(model, ws):
    n = len(model.net_state.get_weights())
    model.net_state.set_weights(ws[:n]); model.net_output.set_weights(ws[n:])


def _mutag_model(mutag_graphs, opt, n_graphs=64, d=32):
    from test_gpu_training import nets
    seq = MultiGraphSequencer(mutag_graphs[:n_graphs], 'g', 'average', n_graphs // 2, shuffle=False)
    ns, no = nets('g', d, True)
    model = GNNgraphBased(ns, no, d, 6, 0.0)
    model.compile(optimizer=opt, loss='categorical_crossentropy')
    return model, seq


# ----------------------------------------------------------------------------------------------------------------------
# an expired grid barrier in the persistent BACKWARD kernel
# ----------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('opt_cls', [Adam, SGD])
def test_a_failed_backward_launch_changes_nothing_and_its_batch_is_trained_late(mutag_graphs, opt_cls, monkeypatch):
    """GNN_DEBUG_FAIL_BWD=1 makes every barrier wait of the persistent backward launch expire at once (the forward launch is
    untouched): the kernel poisons its gradients, the library's validity word stays 0, and everything that would have consumed the
    step - the BatchNormalization moving averages, the optimizer launch - is gated by that word on the device.  No synchronisation was
    added: the host learns of the failure at the NEXT step's one synchronisation (or `resolve_pending()`), warns, and trains the batch
    on the building blocks."""
    model, seq = _mutag_model(mutag_graphs, opt_cls(0.01))
    twin, _ = _mutag_model(mutag_graphs, opt_cls(0.01))
    w0 = _weights(model)
    s0 = [torch.from_numpy(np.random.default_rng(i).normal(0, 0.1, (seq[i][0][0].shape[0], 32)).astype(np.float32)).cuda() for i in range(2)]

    monkeypatch.setenv('GNN_DEBUG_FAIL_BWD', '1')
    r = model.train_step(seq[0], state0=s0[0])
    monkeypatch.delenv('GNN_DEBUG_FAIL_BWD')
    tr = model._trainer
    assert int(r_ok := tr._pending['view'].item()) == 0, r_ok                  # the device word: this step's gradients are not valid
    g = tr.gs.gradients()
    assert not all(bool(torch.isfinite(t).all()) for t in g)                      # ... they are poisoned
    for a, b in zip(_weights(model), w0): assert np.array_equal(a, b)             # weights AND moving statistics: untouched
    # the next step (a good one) fetches the previous word for free, warns, and trains batch 0 late
    with pytest.warns(RuntimeWarning, match='persistent backward kernel'):
        model.train_step(seq[1], state0=s0[1])
    assert tr.recovered_steps == 1 and model._optimizer_obj().iterations == 2
    # the twin: batch 1 on the in-library step, then batch 0 on the building blocks - the same two updates in the same order, Adam's
    # step count included (the launch the gate closed had counted on the host: the count is taken back when the next call fetches the
    # word, BEFORE that call's own update is issued - update n always runs with t = n)
    twin.train_step(seq[1], state0=s0[1])
    twin._trainer.use_native_step = False
    twin.train_step(seq[0], state0=s0[0])
    for a, b in zip(_weights(model), _weights(twin)):
        assert np.allclose(a, b, rtol=1e-5, atol=1e-6), float(np.max(np.abs(a - b)))

    # the same through resolve_pending() (what fit() calls at the end of an epoch) - and a failed LAST step of an epoch inside fit()
    model2, seq2 = _mutag_model(mutag_graphs, opt_cls(0.01))
    monkeypatch.setenv('GNN_DEBUG_FAIL_BWD', '1')
    with pytest.warns(RuntimeWarning, match='persistent backward kernel'):
        h = model2.fit(seq2, epochs=1, verbose=0)
    monkeypatch.delenv('GNN_DEBUG_FAIL_BWD')
    assert model2._trainer.recovered_steps == len(seq2) and model2._trainer._pending is None
    assert all(np.isfinite(w).all() for w in _weights(model2)) and np.isfinite(h['loss'][0])
    assert not all(np.array_equal(a, b) for a, b in zip(_weights(model2), w0))   # every batch WAS trained (late, on the general kernels)


def test_gated_optimizer_launches_leave_everything_alone():
    """gnn_adam_multi / gnn_adam_step / gnn_sgd_step with `gate` (ABI 7): *gate == 0 -> parameters and slots keep their bits."""
    import ctypes as C
    lib = nat.lib()
    p = torch.randn(1000, device='cuda'); g = torch.randn(1000, device='cuda')
    m, v = torch.zeros_like(p), torch.zeros_like(p)
    st = nat.current_stream(p.device)
    for gate_val, moved in ((0, False), (1, True)):
        gate = torch.tensor([gate_val], dtype=torch.int32, device='cuda')
        p0 = p.clone()
        nat.check(lib.gnn_adam_step(nat.ptr(p), nat.ptr(g), nat.ptr(m), nat.ptr(v), 1000, 0.01, 0.9, 0.999, 1e-7, 1, nat.ptr(gate), st))
        assert bool((p != p0).any()) == moved and bool((m != 0).any()) == moved
    p = torch.randn(1000, device='cuda'); vel = torch.zeros_like(p)
    for gate_val, moved in ((0, False), (1, True)):
        gate = torch.tensor([gate_val], dtype=torch.int32, device='cuda')
        p0 = p.clone()
        nat.check(lib.gnn_sgd_step(nat.ptr(p), nat.ptr(g), nat.ptr(vel), 1000, 0.01, 0.9, nat.ptr(gate), st))
        assert bool((p != p0).any()) == moved and bool((vel != 0).any()) == moved
    opt = Adam(0.01)
    w = torch.randn(64, 8, device='cuda'); gw = torch.randn(64, 8, device='cuda'); w0 = w.clone()
    closed = torch.zeros(1, dtype=torch.int32, device='cuda')
    opt.apply_gradients([(gw, w)], gate=closed.data_ptr())
    assert torch.equal(w, w0)
    opt.apply_gradients([(gw, w)])
    assert not torch.equal(w, w0)


# ----------------------------------------------------------------------------------------------------------------------
# head_fast needs out_index to BE the identity
# ----------------------------------------------------------------------------------------------------------------------
def test_large_graph_thin_head_with_a_permuted_out_index_takes_the_general_head():
    """n_out == n_nodes does not make out_index the identity: a C-ABI caller may hand over any permutation.  The library checks on the
    device (one launch, read with k) and takes the general head - gathers and a scatter-add - when output row m is not node m: same loss
    and gradients as the identity order with the targets permuted along."""
    from test_gpu_training import nets
    rng = np.random.default_rng(21)
    N, d = 34_000, 32
    g = er_graph(N, 4 * N, seed=9, aggregation_mode='average')
    t = np.zeros((N, 2)); t[np.arange(N), rng.integers(0, 2, N)] = 1
    g = GraphObject(g.nodes, g.arcs, t, focus='n', aggregation_mode='average', sample_weight=rng.uniform(0.5, 1.5, N))
    x, y, sw = MultiGraphSequencer([g], 'n', 'average', 1, shuffle=False)[0]
    ns, no = nets('n', d, True)
    model = GNNnodeBased(ns, no, d, 3, 0.0)
    model.compile(optimizer=SGD(0.0), loss='categorical_crossentropy')
    s0 = torch.from_numpy(rng.normal(0, 0.1, (N, d)).astype(np.float32)).cuda()
    w0 = _weights(model)
    tr = LoopTrainer(model)
    ref = tr.train_step(x, y, sw, state0=s0, apply=False)
    g_ref = [t_.clone() for t_ in tr.gs.gradients() + tr.go.gradients()]
    _set_weights(model, w0)                                           # (the moving statistics moved)
    perm = torch.from_numpy(rng.permutation(N).astype(np.int32)).cuda()
    identity = model._out_index
    model._out_index = lambda *a, **k: perm
    try:
        tr2 = LoopTrainer(model)
        got = tr2.train_step(x, y.cuda()[perm.long()], sw.cuda()[perm.long()], state0=s0, apply=False)
    finally:
        model._out_index = identity
    assert got['k'] == ref['k'] == 3
    assert abs(float(got['loss']) - float(ref['loss'])) <= 1e-6 * max(1.0, abs(float(ref['loss'])))
    assert rel_err(got['y_pred'].cpu().numpy(), ref['y_pred'][perm.long()].cpu().numpy()) <= 1e-6
    for a, b in zip(tr2.gs.gradients() + tr2.go.gradients(), g_ref):
        assert float((a - b).abs().max()) <= 2e-5 * max(float(b.abs().max()), 1e-12)


# ----------------------------------------------------------------------------------------------------------------------
# small-graph training with node labels far from zero (BatchNormalization first: reference MLP.py:67-70)
# ----------------------------------------------------------------------------------------------------------------------
def _far_labels(graphs, rng, focus='g'):
    """The graphs' structure with node labels N(30, 1) (six columns) / N(-12, 0.5) (the rest) instead of one-hot rows."""
    out = []
    for g in graphs:
        nodes = np.empty(g.nodes.shape, dtype=np.float64)
        nodes[:, :6] = rng.normal(30.0, 1.0, (nodes.shape[0], 6)); nodes[:, 6:] = rng.normal(-12.0, 0.5, (nodes.shape[0], nodes.shape[1] - 6))
        out.append(GraphObject(nodes=nodes, arcs=g.arcs, targets=g.targets, focus=focus))
    return out


@pytest.mark.parametrize('act,d,tiled', [('relu', 32, True), ('selu', 32, False), ('relu', 0, False), ('selu', 0, True), ('tanh', 64, True)])
def test_small_graph_training_with_labels_far_from_zero(mutag_graphs, act, d, tiled, monkeypatch):
    """MUTAG-size batches whose node labels sit 30 sigma from zero, through BOTH orchestrations - the persistent small-graph kernels
    (`k_train_small_fwd / _bwd`: tiles cut at graph boundaries or not) and the Python building blocks - against torch autograd in
    float64.  Every kernel on these paths subtracts the BatchNormalization column mean from a value as it arrives (a (x - mean) + beta,
    (x - mean) rstd, P = (X - mean)^T dZ), and the tiles' statistics are taken around each tile's own first row and merged in double:
    round 4 carried the means in constants (a x + (beta - mean a), P - mean q^T) and lost 1e-4 .. 2e-3 in these gradients.
    d = 0: the state IS the label block (the reference's starter configuration) - state_0 itself is 30 sigma from zero."""
    from test_gpu_training import nets, check_step
    monkeypatch.setattr(LoopTrainer, 'use_tiles', tiled)
    rng = np.random.default_rng(31 + d)
    pool = [g for g in mutag_graphs if g.nodes.shape[0] <= 64] if tiled else mutag_graphs
    gl = _far_labels(pool[:48], rng)
    x, y, sw = MultiGraphSequencer(gl, 'g', 'average', 48, shuffle=False)[0]
    ns, no = nets('g', d, True, act=act, scale=0.5)
    model = GNNgraphBased(ns, no, d, 5, 0.0)
    s0 = None
    if d > 0:
        s0 = rng.normal(0, 0.1, (x[0].shape[0], d)).astype(np.float32)
        if act == 'relu': s0 = np.abs(s0)
    check_step(model, x, y, sw, s0)


# ----------------------------------------------------------------------------------------------------------------------
# gnn_train_step at the size and depth bench.py times it (training.c4_d64_k10)
# ----------------------------------------------------------------------------------------------------------------------
_TRAIN_JOB = {}
TRAIN_C4 = dict(N=1_000_000, E=10_000_000, d=64, K=10)


def _train_c4_job():
    """BASELINE C4 with bench.py's training configuration (starter networks: BatchNormalization + Dense(159 -> 64, selu), BatchNormalization
    + Dense(78 -> 2, softmax); weights default_rng(0 / 1); 10 iterations, threshold 0) and its float64 autograd oracle, one iteration
    checkpointed at a time (oracle/torch_train.py: ~20 GB of host memory instead of ~80)."""
    import time
    from test_gpu_training import oracle_step
    from test_gpu_parity import starter_nets
    c = TRAIN_C4
    t0 = time.time()
    g = er_graph(c['N'], c['E'], aggregation_mode='average')
    x, y, sw = MultiGraphSequencer([g], 'n', 'average', 1, shuffle=False)[0]
    ns, no = starter_nets('n', c['d'])
    assert ns.batch_normalization and no.batch_normalization
    model = GNNnodeBased(ns, no, c['d'], c['K'], 0.0)
    s0 = np.random.default_rng(1).normal(0, 0.1, (c['N'], c['d'])).astype(np.float32)
    t1 = time.time()
    want = oracle_step(model, x, y, sw, s0, 'categorical_crossentropy', checkpoint_iterations=True)
    return dict(model=model, x=x, y=y, sw=sw, s0=s0, want=want, t_graph=t1 - t0, t_oracle=time.time() - t1)


def start_train_oracle(test_names):
    """tests/conftest.py calls this at the end of the collection of a full GPU session: the million-node float64 oracle (minutes of host
    time) runs on a worker thread under the tests in front of this one."""
    if 'job' in _TRAIN_JOB or 'test_train_step_at_the_size_and_depth_the_bench_times_it' not in test_names: return
    from test_gpu_round4 import _Job
    torch.cuda.init()
    _TRAIN_JOB['job'] = _Job(_train_c4_job)


def test_train_step_at_the_size_and_depth_the_bench_times_it(request):
    """`gnn_train_step` on the C4 graph (1 M nodes / 10 M arcs, d = 64, BatchNormalization, 10 iterations: bench.py's
    `training.c4_d64_k10`, the row-streaming kernels of kernels_train_big.hpp - bf16 x 3 split products, the 1 M-row contraction of
    k_train_wgrad32 in float32 MFMA accumulators) against torch autograd in float64: k, loss, training-mode predictions, the final state,
    every gradient per tensor, the moving statistics - on the in-library step AND the building-block orchestration.  Prints the
    per-tensor errors (profiles/r05_train_c4_parity.txt is this test's output on the GPU box)."""
    from test_gpu_training import grad_rows, log_rows, BARS, Y_PRED_BAR
    start_train_oracle([it.name for it in request.session.items] + [request.node.name])
    r = _TRAIN_JOB['job'].result()
    model, x, y, sw, s0, want = r['model'], r['x'], r['y'], r['sw'], r['s0'], r['want']
    c = TRAIN_C4
    model.compile(optimizer=SGD(0.0), loss='categorical_crossentropy')
    w_start = _weights(model)
    kq = want['kinks_state']
    print(f"\nC4-size train step: graph {r['t_graph']:.0f} s, float64 oracle {r['t_oracle']:.0f} s; k = {want['k']}, loss {want['loss']:.6f}; "
          f"pre-activations within 1e-6 of the selu kink: {int(np.sum(kq[0]))} of {c['N'] * c['d'] * c['K']:.1e} (units with one: {int(np.sum(kq[0] > 0))} of {c['d']})")
    assert want['k'] == c['K']
    for native in (True, False):
        _set_weights(model, w_start)
        tr = LoopTrainer(model)
        tr.use_native_step = native
        res = tr.train_step(x, y, sw, state0=torch.from_numpy(s0).cuda(), apply=False)
        torch.cuda.synchronize()
        assert res['k'] == want['k']
        e_loss = abs(float(res['loss']) - want['loss']) / max(1.0, abs(want['loss']))
        e_pred = rel_err(res['y_pred'].cpu().numpy(), want['y_pred'])
        e_state = rel_err(res['state'].cpu().numpy(), want['state'])
        rows = []
        for name, ng_, ref, kinks, M in (('state', tr.gs, want['grads_state'], want['kinks_state'], c['N']),
                                         ('output', tr.go, want['grads_output'], want['kinks_output'], c['N'])):
            scale = max(float(np.max(np.abs(t_))) for t_ in ref)
            rows += grad_rows(name, ng_.gradients(), ref, ng_.bn, kinks, scale, M)
        mv = []
        for net, key in ((model.net_state, 'moving_state'), (model.net_output, 'moving_output')):
            w = net.get_weights()
            mv += [rel_err(w[2], want[key][0]), rel_err(w[3], want[key][1])]
        print(f"  {'gnn_train_step   ' if native else 'building blocks  '} loss {e_loss:.1e}  y_pred {e_pred:.1e}  state {e_state:.1e}  moving "
              + ' '.join(f'{v:.1e}' for v in mv) + '  gradients (own / scale): '
              + '  '.join(f"{r_['net'][0]}.{r_['tensor']} {r_['err_own']:.1e}/{r_['err_scale']:.1e}" for r_ in rows))
        log_rows('c4_train_step', rows, native=native, y_pred=e_pred, state=e_state, loss=e_loss, n_nodes=c['N'], k=c['K'])
        assert e_loss <= 1e-5 and e_pred <= Y_PRED_BAR and e_state <= 1e-5 and max(mv) <= 1e-5, (e_loss, e_pred, e_state, mv)
        bad = [r_ for r_ in rows if not r_['ok']]
        assert not bad, bad


# ----------------------------------------------------------------------------------------------------------------------
# the sharded loop driven from native code (csrc/shard_loop.hpp)
# ----------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('overlap,chunks', [(False, 1), (True, 1), (True, 3)])
def test_native_loop_issues_the_launches_of_the_interpreter_loop(overlap, chunks):
    """`gnn_shard_loop` on 4 emulated ranks of one device: every rank's launches of an iteration come from ONE native call (own-range
    partial sums, the halo kernel whole or in 3 chunk launches, the gate / flag words of the iteration) instead of from
    `ShardedLoop`'s Python methods; the exchange between the ranks is the harness' slice copy in both runs.  Same k, same state bits, same
    output bits, an early exit at the oracle's k.  (The exchange itself - RCCL's C API on the library's own communicator and stream -
    runs in tests/test_gpu_multi.py at world size 1, and at 2 where two GPUs are visible.)"""
    from gnnkeras_amd.distributed import ShardedLoop, partition
    from gnnkeras_amd.synth import er_graph_slice
    from test_gpu_parity import starter_nets
    from oracle.harness import oracle_loop
    N, E, d, R, K = 120_007, 1_200_000, 64, 4, 6
    g = er_graph(N, E, aggregation_mode='average', seed=23)
    x = MultiGraphSequencer([g], 'n', 'average', 1, shuffle=False)[0][0]
    ns, no = starter_nets('n', d, scale=0.25)
    model = GNNnodeBased(ns, no, d, K, 0.05)
    s0 = np.random.default_rng(1).normal(0, 0.1, (N, d)).astype(np.float32)
    k64, st64, o64 = oracle_loop(model, x, s0, np.float64, exact_order=False)
    assert 1 < float(k64) < K
    slices = [er_graph_slice(N, E, lo, hi, aggregation_mode='average', seed=23) for lo, hi in partition(N, R)[1]]
    results = {}
    for native in (False, True):
        shards = [ShardedLoop(model, gs, r, R, 'cuda', overlap=overlap) for r, gs in enumerate(slices)]
        for sl in shards:
            assert sl.overlap == overlap
            if chunks > 1: assert sl.pipeline_supported() and sl.set_pipeline(chunks) == chunks
            if native: sl.enable_native_loop(emulated=True)
            sl._load_state0(torch.from_numpy(s0).cuda()); sl._setup(); sl._initial_flags()
        n = shards[0].plan.rows_per_slice * shards[0].SP
        if overlap and not native:
            for sl in shards: sl._partial(0)
        for it in range(K):
            for sl in shards:
                if native: sl._native_iterations(it, 1)
                elif not overlap: sl._iteration(it)
                else:
                    if chunks == 1: sl._iteration_split(it)
                    else:
                        for ci, (lo, hi) in enumerate(sl._chunk_rows): sl._iteration_split_rows(it, lo, hi, first=ci == 0)
                    if it + 1 < K: sl._partial(it + 1)
            for r, src in enumerate(shards):
                piece = src.buf[(it + 1) & 1].view(-1)[r * n:(r + 1) * n]
                for dst in shards:
                    if dst is not src: dst.buf[(it + 1) & 1].view(-1)[r * n:(r + 1) * n].copy_(piece)
        outs = [sl._output() for sl in shards]
        torch.cuda.synchronize()
        results[native] = ([float(o[0]) for o in outs], torch.cat([o[1] for o in outs]), torch.cat([o[2] for o in outs]))
        del shards
    (kp, stp, op), (kn, stn, on) = results[False], results[True]
    assert kp == kn == [float(k64)] * R
    assert torch.equal(stp, stn) and torch.equal(op, on)
    assert rel_err(stn.cpu().numpy(), st64) <= 1e-5 and rel_err(on.cpu().numpy(), o64) <= 1e-5


# ----------------------------------------------------------------------------------------------------------------------
# heterogeneous models on the persistent small-graph training kernels
# ----------------------------------------------------------------------------------------------------------------------
def _typed_graphs(rng, sizes, dims, A, T, focus, mode, empty_type=None):
    from gnnkeras_amd import CompositeGraphObject
    out = []
    for n, e in sizes:
        pairs = set()
        while len(pairs) < e:
            a, b = rng.integers(0, n, 2)
            if a != b: pairs.add((int(a), int(b)))
        ids = np.array(sorted(pairs), dtype=float)
        arcs = np.concatenate([ids, rng.normal(size=(e, A))], 1)
        types = rng.integers(0, len(dims), n); types[:len(dims)] = np.arange(len(dims))
        if empty_type is not None: types[types == empty_type] = (empty_type + 1) % len(dims)
        tm = np.zeros((n, len(dims)), bool); tm[np.arange(n), types] = True
        if focus == 'g':
            tg = np.zeros((1, T)); tg[0, rng.integers(0, T)] = 1
            kw = {}
        else:
            om = rng.random(n) < 0.8
            tg = np.zeros((int(om.sum()), T)); tg[np.arange(len(tg)), rng.integers(0, T, len(tg))] = 1
            kw = dict(output_mask=om)
        out.append(CompositeGraphObject(nodes=rng.normal(size=(n, max(dims))), arcs=arcs, targets=tg, type_mask=tm, dim_node_label=dims,
                                        focus=focus, aggregation_mode=mode, **kw))
    return out


@pytest.mark.parametrize('focus,D,bn,mode,act,n_graphs,empty', [
    ('g', 32, True, 'composite_average', 'selu', 24, None), ('n', 16, True, 'average', 'tanh', 10, None), ('g', 64, False, 'sum', 'tanh', 30, None),
    ('n', 10, True, 'composite_average', 'relu', 12, None), ('g', 32, True, 'average', 'tanh', 16, 1), ('n', 40, False, 'normalized', 'selu', 6, None)])
def test_composite_small_graph_training_persistent_kernels_match_autograd(focus, D, bn, mode, act, n_graphs, empty):
    """Heterogeneous batches (3 node types, per-type state networks with their own BatchNormalization; reference CompositeGNN.py:275-304)
    through `gnn_train_step` on the PERSISTENT small-graph kernels: the nodes are walked in type order in tiles of one type each
    (kernels_train_small.hpp: TypeTab), the statistics of network t span the tiles of type t, its gradient shares are summed per type, the
    constant input columns (up to d_t + sum d + A = 29: more than one 32-column block with these widths only at d_t > 3 - see the second
    label layout) go through the blocks loop.  k, loss, predictions, the state in the CALLER's node order, every gradient of every
    network and the moving statistics against torch autograd in float64, and against the building-block orchestration.  State widths 16 / 32
    / 64 and padded ones (10 -> 16, 40 -> 64), per-arc weights ('composite_average', 'normalized') and per-row scales, a type without a
    single node, early exit left to the homogeneous tests."""
    from gnnkeras_amd.Models.CompositeGNN import CompositeGNNnodeBased, CompositeGNNgraphBased
    from gnnkeras_amd.Sequencers.GraphSequencers import CompositeMultiGraphSequencer
    from oracle import torch_train
    from oracle.harness import _np, _triple
    from test_gpu_training import grad_rows, log_rows, Y_PRED_BAR
    rng = np.random.default_rng(500 + D + n_graphs)
    dims = (14, 8, 4) if D >= 32 else (4, 2, 3)                    # (14 + 26 + 2 = 42 constant columns for type 0: two blocks of 32)
    A, T = 2, 2
    gl = _typed_graphs(rng, [(int(rng.integers(12, 70)), int(rng.integers(30, 160))) for _ in range(n_graphs)], dims, A, T, focus, mode, empty)
    x, y, sw = CompositeMultiGraphSequencer(gl, focus, mode, n_graphs, shuffle=False)[0]
    inp, lay = get_inout_dims('state', dims, A, T, focus, D)
    ns = [MLP(i, lay, act, 'lecun_normal', 'lecun_normal', rng=t, batch_normalization=bn) for t, i in enumerate(inp)]
    for n_ in ns: n_.set_weights([a * 0.4 if a.ndim == 2 else a for a in n_.get_weights()])
    inp, lay = get_inout_dims('output', dims, A, T, focus, D)
    no = MLP(inp[0], lay, 'softmax', 'glorot_normal', 'glorot_normal', rng=9, batch_normalization=bn)
    if bn:
        for n_ in ns + [no]:
            w = n_.get_weights()
            w[0] = rng.uniform(0.7, 1.3, w[0].shape).astype(np.float32); w[1] = rng.normal(0, 0.2, w[1].shape).astype(np.float32)
            n_.set_weights(w)
    K = 6
    model = {'n': CompositeGNNnodeBased, 'g': CompositeGNNgraphBased}[focus](ns, no, D, K, 0.0)
    model.compile(optimizer=SGD(0.0), loss='categorical_crossentropy')
    N = x[0].shape[0]
    s0 = rng.normal(0, 0.1, (N, D)).astype(np.float32)
    if act == 'relu': s0 = np.abs(s0)
    nodes, arcs, dnl, tmask, sm, om_, cas, adj, an, ng = x
    mask = np.logical_and(_np(sm).reshape(-1), _np(om_).reshape(-1))
    want = torch_train.composite_train_step(
        _np(nodes), _np(arcs), _np(dnl).reshape(-1), _np(tmask).reshape(len(dims), -1), [_triple(c) for c in cas], _triple(adj), _triple(an),
        _triple(ng), mask, net_state=[n_.spec() for n_ in ns], net_output=no.spec(), state_vect_dim=D, max_iteration=K, state_threshold=0.0,
        focus=focus, state0=s0, y=_np(y), sample_weight=_np(sw), loss='categorical_crossentropy')
    assert want['k'] == K
    w_start = [[w.copy() for w in n_.get_weights()] for n_ in ns + [no]]
    for native in (True, False):
        for n_, w in zip(ns + [no], w_start): n_.set_weights(w)
        tr = LoopTrainer(model); tr.use_native_step = native
        res = tr.train_step(x, y, sw, state0=torch.from_numpy(s0).cuda(), apply=False)
        assert res['k'] == K
        assert abs(float(res['loss']) - want['loss']) <= 1e-5 * max(1.0, abs(want['loss']))
        e_pred, e_state = rel_err(res['y_pred'].cpu().numpy(), want['y_pred']), rel_err(res['state'].cpu().numpy(), want['state'])
        allref = [r for g in want['grads_state'] for r in g] + want['grads_output']
        scale = max(float(np.max(np.abs(r))) for r in allref)
        rows = []
        counts = _np(tmask).reshape(len(dims), -1).sum(1)
        kinks_loop = [np.concatenate([kq[0] for kq in want['kinks_state']])]      # (a kink in ANY type's network reaches every network through the loop)
        for t_, (g_, ref) in enumerate(zip(tr.gs, want['grads_state'])):
            if counts[t_] == 0:                                          # a type without nodes: zero gradients, untouched moving statistics
                assert all(float(t.abs().max()) == 0.0 for t in g_.gradients())
                continue
            rows += grad_rows(f'state{t_}', g_.gradients(), ref, bn, kinks_loop, scale, int(counts[t_]))
        rows += grad_rows('output', tr.go.gradients(), want['grads_output'], bn, want['kinks_output'], scale, int(want['y_pred'].shape[0]))
        log_rows(os.environ.get('PYTEST_CURRENT_TEST', ''), rows, native=native, y_pred=e_pred, n_nodes=N, k=K)
        assert e_pred <= Y_PRED_BAR and e_state <= 1e-5, (native, e_pred, e_state)
        bad = [r_ for r_ in rows if not r_['ok']]
        assert not bad, (native, bad)
        if bn:
            for t_, (n_, mv) in enumerate(zip(ns, want['moving_state'])):
                if counts[t_] == 0: continue
                w = n_.get_weights()
                assert rel_err(w[2], mv[0]) <= 1e-5 and rel_err(w[3], mv[1]) <= 1e-5, t_
            w = no.get_weights()
            assert rel_err(w[2], want['moving_output'][0]) <= 1e-5 and rel_err(w[3], want['moving_output'][1]) <= 1e-5



def test_large_graph_training_with_every_dense_kernel_forming_dz_itself(monkeypatch):
    """GNN_TRAIN_DZ=0 (read at every call): the round-4 flow of the large-graph backward sweep - the weight-gradient and input-gradient
    kernels each form dZ = G (.) act'(Y) from G and Y, k_train_bwd_dx applies the whole BatchNorm input gradient, the plain transposed
    aggregate - which round 5 replaced by default with k_aggregate_dz (the aggregate's epilogue leaves dZ, the state half's BatchNorm
    term is added there) and which models without constant inputs still take.  Both flows against the same float64 autograd oracle."""
    from test_gpu_round3 import test_large_graph_training_step_matches_autograd as step
    from test_gpu_round4 import test_thin_output_head_over_every_node_matches_autograd as head, test_large_graph_training_kernels_for_every_activation as act
    monkeypatch.setenv('GNN_TRAIN_DZ', '0')
    step(64, True, 'average', 0.0)
    step(32, False, 'average', -1.0)
    head(64, True, 'n', 2, 'categorical_crossentropy', 0.0)
    act('relu', 64)


@pytest.mark.parametrize('d', [64, 32])
def test_large_graph_training_without_batchnorm_takes_both_gradients_in_one_pass(d):
    """Without BatchNormalization nothing global stands between an iteration's weight gradient and its input gradient: k_train_wgrad_dx_b6
    forms both from one pass over the rows (dZ | state | agg | constants through an LDS ring, 1 412 instead of 1 920 bytes a row).  The step
    against float64 autograd and against the building-block orchestration, as every large-graph model (scripts/micro/rowgemm_check.hip holds
    the kernel bit for bit against the two kernels it replaces)."""
    from test_gpu_round3 import test_large_graph_training_step_matches_autograd as run
    run(d, False, 'average', 0.0)
    run(d, False, 'sum', 0.0)


def test_large_graph_training_on_the_kernels_before_the_lds_ring(monkeypatch):
    """GNN_TRAIN_WGRAD_B6=0 GNN_TRAIN_FUSED_BWD=0 (read at every call): the weight gradient on the f32-input matrix instructions
    (k_train_wgrad32) and the two-kernel backward pass stay selectable; the same tests against the same oracle."""
    from test_gpu_round3 import test_large_graph_training_step_matches_autograd as step
    monkeypatch.setenv('GNN_TRAIN_WGRAD_B6', '0'); monkeypatch.setenv('GNN_TRAIN_FUSED_BWD', '0')
    step(64, True, 'average', 0.0)
    step(32, True, 'sum', 0.0)
    step(64, False, 'average', 0.0)


@pytest.mark.parametrize('N,d,bn,mode', [(40_037, 64, True, 'average'), (33_001, 32, True, 'average'), (40_037, 64, False, 'average'),
                                         (36_001, 64, True, 'normalized'), (36_001, 32, False, 'normalized')])
@prefetch_oracle
def test_large_graph_training_with_a_ragged_last_tile(N, d, bn, mode):
    """Node counts that are no multiple of 64 / 16: the last workgroup of k_train_wgrad_b6 (and of the one-pass kernel without BatchNormalization)
    fills its LDS ring past the end of the rows - LDS-DMA loads outside the buffer window must land ZEROS, not leave the slot's previous
    rows - and the last tile of k_train_fwd_b6 / k_train_bwd_dx_b6 is ragged (windows of the arrays' exact sizes instead of a select per
    load).  'normalized': per-ARC weights - the weighted instances of the buffered gathers (k_aggregate_stats / k_aggregate_dz with HAS_W).
    Every gradient against float64 autograd, both orchestrations, as the other large-graph tests."""
    from test_gpu_training import nets, check_step
    from gnnkeras_amd.synth import er_graph
    from gnnkeras_amd.Sequencers.GraphSequencers import MultiGraphSequencer
    rng = np.random.default_rng(N + d)
    g = er_graph(N, 6 * N + 11, seed=9, aggregation_mode=mode)
    om = rng.random(N) < 0.6
    t = np.zeros((int(om.sum()), 2)); t[np.arange(len(t)), rng.integers(0, 2, len(t))] = 1
    g = GraphObject(g.nodes, g.arcs, t, focus='n', set_mask=rng.random(N) < 0.9, output_mask=om, aggregation_mode=mode,
                    sample_weight=rng.uniform(0.5, 1.5, len(t)))
    x, y, sw = MultiGraphSequencer([g], 'n', mode, 1, shuffle=False)[0]
    ns, no = nets('n', d, bn, scale=0.5)
    s0 = rng.normal(0, 0.1, (N, d)).astype(np.float32)
    check_step(GNNnodeBased(ns, no, d, 3, 0.0), x, y, sw, s0, oracle_key=('ragged', N, d, bn, mode))
