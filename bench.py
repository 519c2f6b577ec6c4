#!/usr/bin/env python3
"""bench.py — the headline measurement of BASELINE.json on MI355X.

    python bench.py --gpus N --steps K --warmup W

N > 1: one rank per GPU over RCCL.  Either the caller starts the ranks (`python -m torch.distributed.run --nproc-per-node N
bench.py --gpus N ...`: RANK / LOCAL_RANK / WORLD_SIZE in the environment) or, when WORLD_SIZE is unset, this process
starts them itself as a CHILD `torch.distributed.run` before anything here touches the GPU and relays rank 0's JSON line.

Workload (config.workload): BASELINE config C4 — synthetic directed Erdős–Rényi graph, 1 000 000 nodes / 10 000 000
arcs, state_dim 64, node-focused, 'average' aggregation, max_iteration 50, threshold 0 (fixed work, k = 50), starter
networks (BN + Dense(159->64, selu) state net, BN + Dense(78->2, softmax) output net), weights default_rng(0/1),
state0 = default_rng(1).normal(0, 0.1).  One *step* = one whole forward pass `Loop(...)` of that graph: setup
aggregates + 50 fused iterations + output network, inputs already resident in HBM.

metric  = node-state updates/s = arcs x iterations / wall seconds (whole job, all ranks), float32 arithmetic.
roofline = the fused iteration kernel: algorithmic bytes per launch / its average duration,
           measured live with HIP events recorded on the launch stream around the 50 iteration launches.
cpu_baseline = the restatement of the reference's un-fused op sequence (oracle/cpu_baseline.py) timed on rank 0 at
           N = 1 on a bounded sample (iterations of the same graph: warm-up, then the median), three variants - torch on
           all host threads, torch on one thread, NumPy/SciPy on one thread; `value` is the fastest.
Also reported (extra keys): the MUTAG batch-32 forward (BASELINE config C2) in ms/graph next to its CPU baseline, and
`beyond_infinity_cache`: the same iteration kernel on a 4 M-node / 40 M-arc graph whose 1 GB state array cannot sit in
the 256 MiB Infinity Cache (the C4 state array, 256 MB, can); `wide_state_d200`: the per-iteration launch of the 129..256-wide
fused kernel (d = 200 on 300 k nodes / 3 M arcs) with its share of the HBM roofline.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# predict() / evaluate() overlap independent batches on up to 8 HIP streams; by default HIP multiplexes all streams of
# a process onto 4 hardware queues.  Must be set before the HIP runtime starts (importing torch does that).
os.environ.setdefault('GPU_MAX_HW_QUEUES', '16')
# Multi-process GPU work on this image needs dmabuf IPC (RCCL's peer mappings fail with "hipIpcGetMemHandle: invalid argument"
# under the legacy mode).  The image exports it already; set here as well so that every way of starting ranks - the driver's
# torch.distributed.run, this file's own child launch below, the tests' subprocesses - runs with the same value.
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')

import numpy as np
import torch

HBM_PEAK = 8.0e12          # B/s, MI355X_MICROARCH.md chip-level parameters (spec)


def starter_nets(d, device, focus='n'):
    from gnnkeras_amd.Models.MLP import MLP, get_inout_dims
    inp, lay = get_inout_dims('state', 14, 3, 2, focus, d)
    ns = MLP(inp[0], lay, 'selu', 'lecun_normal', 'lecun_normal', rng=0, device=device)
    inp, lay = get_inout_dims('output', 14, 3, 2, focus, d)
    no = MLP(inp[0], lay, 'softmax', 'glorot_normal', 'glorot_normal', rng=1, device=device)
    return ns, no


def algorithmic_bytes_per_iteration(n_nodes, n_arcs, d, h1, per_arc_weights):
    """E(4 + 4d [+4]) + N(4 + 4d + 4d + 4 H1): arc source id + gathered neighbour row (+ weight);
    per node row pointer, own state read, new state write, constant-term read."""
    return n_arcs * (4 + 4 * d + (4 if per_arc_weights else 0)) + n_nodes * (4 + 8 * d + 4 * h1)


def nat_kernel():
    from gnnkeras_amd import _native as nat
    return nat.lib().gnn_last_kernel_name().decode()


def timed_median(fn, reps=5):
    """(median wall seconds of `reps` synchronised calls, the last result): a ~1 ms walk timed once is at the mercy of one host hiccup
    (single shots of the same build ranged 1.2 .. 1.7 ms across boxes of the pool)."""
    ts, res = [], None
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        res = fn()
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    return float(np.median(ts)), res


def mutag_section(device, cpu: bool):
    """BASELINE C2: all 136 MUTAG batches of 32 graphs, d = 32, max_iteration = 50, threshold 0.01."""
    from gnnkeras_amd.load_MUTAG import load_graphs
    from gnnkeras_amd.Models.GNN import GNNgraphBased
    from gnnkeras_amd.Sequencers.GraphSequencers import MultiGraphSequencer
    graphs = load_graphs()
    seq = MultiGraphSequencer(graphs, 'g', 'average', 32, shuffle=False, device=device)
    ns, no = starter_nets(32, device, 'g')
    gnn = GNNgraphBased(ns, no, 32, 50, 0.01)
    items = [seq[i][0] for i in range(len(seq))]
    inputs = [gnn.process_inputs(x) for x in items]
    rng = np.random.default_rng(1)
    s0s = [torch.from_numpy(rng.normal(0, 0.1, (x[0].shape[0], 32)).astype(np.float32)).to(device) for x in items]
    for inp, s0 in zip(inputs, s0s): gnn.Loop(*inp, state0=s0)              # warm-up: one untimed pass
    torch.cuda.synchronize()
    ks = []
    t0 = time.perf_counter()
    for inp, s0 in zip(inputs, s0s):
        k, st, out = gnn.Loop(*inp, state0=s0)
        ks.append(k)
    torch.cuda.synchronize()
    t_one = time.perf_counter() - t0                                    # one stream: a batch's latency, 136 times
    # the way predict() / evaluate() walk a sequencer: independent batches side by side on a few HIP streams (a batch
    # keeps ~16 of the 256 CUs busy)
    width = gnn._round_width(seq, device)
    run = lambda i: gnn.Loop(*inputs[i], state0=s0s[i])
    for _ in gnn._batches_concurrently(len(inputs), run, device, width): pass
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ks = [r[0] for _, r in gnn._batches_concurrently(len(inputs), run, device, width)]
    torch.cuda.synchronize()
    t_gpu = time.perf_counter() - t0
    ks = [float(k) for k in ks]
    n_graphs = len(graphs)
    # ... and the way they walk it where the library supports convergence groups: runs of batches merged into one graph
    # whose batches are independent loops of ONE launch (own predicate, own k: include/gnnloop.h group_node_begin)
    plan = gnn._group_plan(seq, device)
    s0_cat = {id(bs): torch.cat([s0s[b] for b in bs]) for bs in (plan or []) if len(bs) > 1 or getattr(bs, 'parts', None)}     # (predict() draws state_0 per launch)
    def grouped(model):                    # as _LoopModel._forward_batches does: the launches of the plan on side streams
        def launch(li):
            bs = plan[li]
            parts = getattr(bs, 'parts', None)
            if len(bs) == 1 and not parts: return model.Loop(*inputs[bs[0]], state0=s0s[bs[0]])
            x, begin = seq.merged_batches(bs)
            kw = {}
            if parts:        # a batch too big for one CU's LDS, cut along graph boundaries into groups that share the loop's condition
                begin, kw['group_sets'] = bs.groups_and_sets({b: begin[i + 1] - begin[i] for i, b in enumerate(bs)})
            return model.Loop(*model.process_inputs(x), state0=s0_cat[id(bs)], groups=begin, **kw)
        ks_b = [None] * len(items)
        for li, (k, st, o) in model._run_plan(plan, launch, device):
            for j, b in enumerate(plan[li]): ks_b[b] = k.reshape(-1)[j]
        return torch.stack(ks_b)
    t_grp = None
    if plan is not None:
        grouped(gnn); torch.cuda.synchronize()
        t_grp, ks_g = timed_median(lambda: grouped(gnn))
        assert [float(v) for v in ks_g.cpu()] == ks, 'grouped launches must reproduce every batch\'s iteration count'
    t_best = min(t_gpu, t_grp) if t_grp is not None else t_gpu
    arcs_iters = sum(x[1].shape[0] * k for x, k in zip(items, ks))
    res = {'workload': 'MUTAG (TU Mutagenicity) 4337 graphs as 136 batches of 32, state_dim=32, max_iteration=50, '
                       'threshold=0.01, graph-focused forward',
           'fwd_ms_per_graph': 1e3 * t_best / n_graphs, 'fwd_ms_per_batch': 1e3 * t_best / len(items),
           'how': ('grouped launches: %d (%s), each batch an independent loop: one CU per batch with its state in LDS where it fits, '
                   'cut along graph boundaries into groups that share the convergence flag otherwise' % (len(plan), ' + '.join(str(len(bs)) for bs in plan) + ' batches')) if t_best == t_grp
                  else '%d side streams' % width,
           'grouped_fwd_ms_per_graph': None if t_grp is None else 1e3 * t_grp / n_graphs,
           'side_streams_fwd_ms_per_graph': 1e3 * t_gpu / n_graphs, 'concurrent_batches': width,
           'one_stream_fwd_ms_per_graph': 1e3 * t_one / n_graphs, 'one_stream_fwd_ms_per_batch': 1e3 * t_one / len(items),
           'us_per_iteration': 1e6 * t_one / max(sum(ks), 1), 'mean_k': float(np.mean(ks)),
           'updates_per_s': arcs_iters / t_best}
    # the early-exit path: the same batches with a contractive state network (kernel x 0.25) stop well before
    # max_iteration at threshold 0.01 (the random-initialised network above never does: mean_k = 50)
    w = ns.get_weights()
    ns_c = ns.clone(True); ns_c.set_weights([a * 0.25 if a.ndim == 2 else a for a in w])
    gnn_c = GNNgraphBased(ns_c, no, 32, 50, 0.01)
    run_c = lambda i: gnn_c.Loop(*inputs[i], state0=s0s[i])
    for _ in gnn_c._batches_concurrently(len(inputs), run_c, device, width): pass
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ks_c = [r[0] for _, r in gnn_c._batches_concurrently(len(inputs), run_c, device, width)]
    torch.cuda.synchronize()
    t_c = time.perf_counter() - t0
    ks_c = [float(k) for k in ks_c]
    t_cg = None
    if plan is not None:
        grouped(gnn_c); torch.cuda.synchronize()
        t_cg, ks_cg = timed_median(lambda: grouped(gnn_c))
        # (another kernel, other summation orders: a batch sitting exactly on the threshold may stop one iteration apart)
        assert max(abs(a_ - b_) for a_, b_ in zip([float(v) for v in ks_cg.cpu()], ks_c)) <= 1
    res['converging'] = {'note': 'state-network kernel x 0.25: contractive, the device-side predicate stops the loop early',
                         'mean_k': float(np.mean(ks_c)), 'fwd_ms_per_graph': 1e3 * (min(t_c, t_cg) if t_cg is not None else t_c) / n_graphs,
                         'grouped_fwd_ms_per_graph': None if t_cg is None else 1e3 * t_cg / n_graphs,
                         'side_streams_fwd_ms_per_graph': 1e3 * t_c / n_graphs}
    # BASELINE C1, the reference's own default (starter.py): state_vect_dim = 0 (the state is the 14 label columns), max_iteration = 5
    ns0, no0 = starter_nets(0, device, 'g')
    gnn0 = GNNgraphBased(ns0, no0, 0, 5, 0.01)
    plan0 = gnn0._group_plan(seq, device)
    def walk0():
        gnn0._k_seen = []
        outs = [o for _, o in gnn0._forward_batches(seq, device)]
        return torch.cat([k.reshape(-1) for k in gnn0._k_seen])
    walk0(); torch.cuda.synchronize()
    t_0, ks0 = timed_median(walk0)
    res['starter_config'] = {'workload': 'state_vect_dim=0, max_iteration=5, threshold=0.01 (starter.py), the predict() walk over the 136 batches',
                             'fwd_ms_per_graph': 1e3 * t_0 / n_graphs, 'mean_k': float(ks0.mean()),
                             'how': ('grouped launches: %d' % len(plan0)) if plan0 is not None else 'side streams'}
    if cpu:
        from oracle import torch_cpu
        from oracle.harness import _np, _triple
        # every one of the 136 batches, warm-up + median (a CPU forward of a ~1 k-node batch is ~10 ms on one thread:
        # ~1.5 s per pass).  1 thread and all threads are both timed - at this size thread hand-offs cost more than they buy - and
        # the faster is the baseline.
        n_thr = torch.get_num_threads()
        per_threads = {}
        def cpu_pass(thr, batches, reps):
            torch.set_num_threads(thr)
            per_batch = []
            for i in batches:
                x = items[i]
                ts = []
                for rep in range(reps + 1):              # 1 warm-up + reps timed, median of the timed
                    t1 = time.perf_counter()
                    torch_cpu.loop(_np(x[0]), _np(x[1]), _triple(x[5]), _triple(x[6]), _triple(x[7]),
                                   np.ones(x[0].shape[0], bool), net_state=ns.spec(), net_output=no.spec(), state_vect_dim=32,
                                   max_iteration=50, state_threshold=0.01, focus='g', state0=_np(s0s[i]))
                    ts.append(time.perf_counter() - t1)
                per_batch.append(float(np.median(ts[1:])))
            return 1e3 * sum(per_batch) / sum(int(seq[i][1].shape[0]) for i in batches)         # ms per graph (graph focus: one target row per graph)
        all_b = list(range(len(items)))
        per_threads[1] = cpu_pass(1, all_b, 3)
        probe = cpu_pass(n_thr, all_b[::17], 1)           # all threads: a probe first (25x slower than one thread on the 256-cpu box)
        per_threads[n_thr] = cpu_pass(n_thr, all_b, 3) if probe < 2 * per_threads[1] else probe
        torch.set_num_threads(n_thr)
        fastest = min(per_threads, key=per_threads.get)
        res['cpu_fwd_ms_per_graph'] = per_threads[fastest]
        res['cpu_fwd_ms_per_graph_by_threads'] = {str(k): v for k, v in per_threads.items()}
        res['speedup_vs_cpu'] = per_threads[fastest] / res['fwd_ms_per_graph']
        res['cpu_sample'] = (f'all {len(items)} batches, torch CPU, 1 warm-up + median of 3 per batch, faster of 1 / {n_thr} threads = {fastest}'
                             + ('' if probe < 2 * per_threads[1] else f' ({n_thr} threads probed on 8 batches only: {probe / per_threads[1]:.0f}x slower)'))
    return res


def mutag_dp_section(device, rank, world):
    """N > 1: the MUTAG data set over the ranks (gnnkeras_amd/data_parallel.py; SURVEY 8e "Other cases": batches are
    block-diagonal, so they shard by graph with no halo).  predict(): the launches of the group plan dealt to the ranks, outputs
    all-gathered; train_step(): every batch of 32 graphs cut into one shard per rank, the step of the whole batch reproduced
    (BatchNorm statistics, convergence flag, loss and gradient sums across ranks).  Collective: every rank calls it."""
    import torch.distributed as dist
    from gnnkeras_amd.load_MUTAG import load_graphs
    from gnnkeras_amd.Models.GNN import GNNgraphBased
    from gnnkeras_amd.Models.training import Adam
    from gnnkeras_amd.Sequencers.GraphSequencers import MultiGraphSequencer
    from gnnkeras_amd.data_parallel import DataParallel
    graphs = load_graphs()
    seq = MultiGraphSequencer(graphs, 'g', 'average', 32, shuffle=False, device=device)
    ns, no = starter_nets(32, device, 'g')
    gnn = GNNgraphBased(ns, no, 32, 50, 0.01)
    gnn.compile(optimizer=Adam(0.001), loss='categorical_crossentropy', metrics=['accuracy'])
    dpm = DataParallel(gnn)

    def timed(fn, reps):
        fn(); torch.cuda.synchronize(); dist.barrier()
        t0 = time.perf_counter()
        for _ in range(reps): fn()
        torch.cuda.synchronize(); dist.barrier()
        t = torch.tensor([(time.perf_counter() - t0) / reps], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t)
    t_pred = timed(lambda: dpm.predict(seq), 3)
    n_steps = 8
    t_step = timed(lambda: [dpm.train_step(dpm.shard(seq, i)) for i in range(n_steps)], 1) / n_steps
    ns2, no2 = starter_nets(32, device, 'g')
    gnn2 = GNNgraphBased(ns2, no2, 32, 50, 0.01)
    gnn2.compile(optimizer=Adam(0.001), loss='categorical_crossentropy', metrics=['accuracy'])
    dpr = DataParallel(gnn2, exact=False)
    t_rep = timed(lambda: [dpr.train_step(dpr.shard(seq, i)) for i in range(n_steps)], 2) / n_steps
    return {'replica_train_step_ms_per_batch': 1e3 * t_rep,
            'replica_train_how': 'exact=False: the in-library step (persistent kernels) on the own shard, one all-reduce of the gradients '
                                 '(weighted by target rows) and the BatchNorm moving statistics per step; not the merged-batch step',
            'workload': 'MUTAG 4337 graphs as 136 batches of 32, state_dim=32, max_iteration=50, threshold=0.01',
            'predict_ms_per_graph': 1e3 * t_pred / len(graphs), 'predict_ms': 1e3 * t_pred,
            'how': f'group-plan launches dealt round-robin to {world} ranks, outputs all-gathered (RCCL)',
            'train_step_ms_per_batch': 1e3 * t_step,
            'train_how': f'each batch of 32 graphs as {world} shards of whole graphs; BatchNorm statistics, convergence flag, P / q and '
                         f'gradient sums exchanged per iteration (building-block path)'}


def training_section(device, graph_x, d):
    """SURVEY 8f rank 1 next to the forward numbers: `train_step` (training-mode forward on batch statistics, loss, BPTT through the
    executed iterations, Adam) - MUTAG batches of 32 at d = 32 x 50 iterations and in the reference's starter configuration
    (state_vect_dim = 0, 5 iterations), a fit() epoch of each over the whole data set (3 469 training + 868 validation graphs), and
    one step on the C4 graph (d = 64, 10 iterations)."""
    from gnnkeras_amd.load_MUTAG import load_graphs
    from gnnkeras_amd.Models.GNN import GNNgraphBased, GNNnodeBased
    from gnnkeras_amd.Models.training import Adam
    from gnnkeras_amd.Sequencers.GraphSequencers import MultiGraphSequencer
    graphs = load_graphs()
    out = {}
    # (BASELINE C1 quotes batch = 32; the reference's starter.py itself defaults to batch_size = 1000: 4 training batches of ~30 k nodes)
    for name, dd, it, bs in (('mutag_d32_k50', 32, 50, 32), ('mutag_starter_config', 0, 5, 32), ('mutag_starter_py_batch_1000', 0, 5, 1000)):
        ns, no = starter_nets(dd, device, 'g')
        gnn = GNNgraphBased(ns, no, dd, it, 0.01)
        gnn.compile(optimizer=Adam(0.01), loss='categorical_crossentropy', metrics=['accuracy'])
        seq = MultiGraphSequencer(graphs[:bs * 20], 'g', 'average', bs, shuffle=False, device=device)
        for i in range(len(seq)): gnn.train_step(seq[i], seed=0)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(len(seq)): r = gnn.train_step(seq[i], seed=0)
        torch.cuda.synchronize(); t_step = (time.perf_counter() - t0) / len(seq)
        tr = MultiGraphSequencer(graphs[:-868], 'g', 'average', bs, shuffle=True, device=device)
        va = MultiGraphSequencer(graphs[-868:], 'g', 'average', bs, shuffle=False, device=device)
        gnn.fit(tr, epochs=1, validation_data=va, verbose=0)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        gnn.fit(tr, epochs=2, validation_data=va, verbose=0)
        torch.cuda.synchronize(); t_epoch = (time.perf_counter() - t0) / 2
        out[name] = {'train_step_ms_per_batch': 1e3 * t_step, 'k': int(r['k']), 'fit_epoch_ms': 1e3 * t_epoch,
                     'epoch': f'{len(tr)} training steps of {bs} graphs + validation on 868 graphs + reshuffle / device re-merge'}
    try:
        out['composite_small_graphs'] = composite_training_section(device)
    except Exception as e:                            # never lose the other numbers to this one
        out['composite_small_graphs'] = {'error': str(e)[:300]}
    try:
        out['mutag_lgnn_starter'] = lgnn_starter_section(device, graphs)
    except Exception as e:
        out['mutag_lgnn_starter'] = {'error': str(e)[:300]}
    ns, no = starter_nets(d, device, 'n')
    gnn = GNNnodeBased(ns, no, d, 10, 0.0)
    gnn.compile(optimizer=Adam(0.001), loss='categorical_crossentropy', metrics=['accuracy'])
    N = graph_x[0][0].shape[0]
    y = torch.zeros((N, 2), device=device); y[:, 0] = 1.0
    data = (graph_x[0], y, None)
    for _ in range(2): gnn.train_step(data, seed=0)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(3): r = gnn.train_step(data, seed=0)
    torch.cuda.synchronize()
    out['c4_d64_k10'] = {'train_step_ms': 1e3 * (time.perf_counter() - t0) / 3, 'k': int(r['k']),
                         'workload': 'the C4 graph, node-focused, every node a target, 10 iterations, BatchNormalization on batch statistics',
                         'arithmetic': 'f32; the first Dense forward and dZ.W^T of an iteration as six bf16 MFMA products of three-term bf16 splits with f32 '
                                       'accumulation (f32-chain accuracy, DESIGN.md 6b; GNN_TRAIN_BF16X6=0: f32-input MFMAs), weight gradient on f32-input MFMAs'}
    del gnn, data, y
    try:
        out['c5_d64_k10'] = c5_training_entry(device, d)
    except Exception as e:                            # never lose the other numbers to this one
        out['c5_d64_k10'] = {'error': str(e)[:300]}
    return out


def c5_training_entry(device, d=64, K_it=10, N=500_000, E=5_000_000, steps=3):
    """BASELINE C5 as a TRAINING step (reference CompositeGNN.py:275-304): 500 k nodes / 5 M arcs, 3 node types with label widths
    (14, 8, 4), one BatchNormalization + Dense(selu) state network per type, d = 64, 10 iterations, node-focused, every node a target -
    `gnn_train_step` with `composite` on the row-streaming kernels in position space (csrc/train_composite_big.hpp)."""
    from gnnkeras_amd import _native as nat
    from gnnkeras_amd.synth import er_composite_graph
    from gnnkeras_amd.Models.training import Adam
    from gnnkeras_amd.Sequencers.GraphSequencers import CompositeMultiGraphSequencer
    gnn, _, dims = composite_model(d, K_it, device)
    gnn.compile(optimizer=Adam(0.001), loss='categorical_crossentropy', metrics=['accuracy'])
    graph = er_composite_graph(N, E, dim_node_label=dims, aggregation_mode='average', seed=1234)
    x, y, sw = CompositeMultiGraphSequencer([graph], 'n', 'average', 1, shuffle=False, device=device)[0]
    data = (x, y, None)
    for _ in range(2): gnn.train_step(data, seed=0)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps): r = gnn.train_step(data, seed=0)
    torch.cuda.synchronize()
    t_step = (time.perf_counter() - t0) / steps
    path = nat.lib().gnn_last_kernel_name().decode()
    res = {'train_step_ms': 1e3 * t_step, 'k': int(r['k']), 'orchestration': path,
           'workload': f'composite ER {N} nodes / {E} arcs, 3 node types (label widths {tuple(dims)}), d = {d}, {K_it} iterations, node-focused, every node a '
                       f'target, BatchNormalization on the batch statistics of each type\'s rows, Adam; aggregation average'}
    ks = kernel_stats_lookup('r06_c5_train')
    if ks is not None: res['kernels_us'] = ks
    return res


def kernel_stats_lookup(tag):
    """Per-kernel average durations (us) of a committed `rocprofv3 --kernel-trace --stats` summary, profiles/<tag>_kernel_stats.json
    (scripts/parse_kernel_stats.py writes it with the library's source hash): reported only when it belongs to the sources that run."""
    from gnnkeras_amd import _native as nat
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'profiles', tag + '_kernel_stats.json')
    try:
        with open(path) as f: rec = json.load(f)
    except (OSError, ValueError):
        return None
    if rec.get('source_hash') != nat.source_hash(): return {'stale': 'profiles/%s_kernel_stats.json was taken on other sources' % tag}
    return {'from': 'profiles/%s_kernel_stats.json (rocprofv3 --kernel-trace --stats of scripts/train_c5.py)' % tag, 'avg_us': rec['avg_us'], 'calls_per_step': rec.get('calls_per_step')}


def lgnn_starter_section(device, graphs):
    """The reference's DEFAULT layered model (starter.py:37-47: 3 GNN layers, get_state and get_output, training_mode 'serial'; each layer the
    starter GNN - state = the label columns, 5 iterations, threshold 0.01 - batch_size 1000, Adam 0.01; reference LGNN.py:290-362): one
    `lgnn.fit()` epoch over the starter split (2 837 training graphs, 750 validation graphs) - per layer one epoch of training, then every
    graph alone through the trained layer (batch size 1, as the reference propagates states / outputs into the next layer's labels) - and
    the training step of each layer on its batches of 1 000 graphs."""
    from gnnkeras_amd.Models.GNN import GNNgraphBased
    from gnnkeras_amd.Models.LGNN import LGNN
    from gnnkeras_amd.Models.MLP import MLP, get_inout_dims
    from gnnkeras_amd.Models.training import Adam
    from gnnkeras_amd.Sequencers.GraphSequencers import MultiGraphSequencer
    layers, bs = 3, 1000
    gnns = []
    for i in range(layers):
        inp, lay = get_inout_dims('state', 14, 3, 2, 'g', 0, layer=i, get_state=True, get_output=True)
        ns = MLP(inp[0], lay, 'selu', 'lecun_normal', 'lecun_normal', rng=10 + i, device=device)
        inp, lay = get_inout_dims('output', 14, 3, 2, 'g', 0, layer=i, get_state=True, get_output=True)
        no = MLP(inp[0], lay, 'softmax', 'glorot_normal', 'glorot_normal', rng=20 + i, device=device)
        gnns.append(GNNgraphBased(ns, no, 0, 5, 0.01))
    lg = LGNN(gnns, True, True)
    lg.compile(optimizer=Adam(0.01), loss='categorical_crossentropy', average_st_grads=True, metrics=['accuracy'], training_mode='serial')
    gs = [g.copy() for g in graphs]
    for g in gs: g.setAggregation('average')
    tr = MultiGraphSequencer(gs[:-1500], 'g', 'average', bs, shuffle=True, device=device)
    va = MultiGraphSequencer(gs[-750:], 'g', 'average', bs, shuffle=False, device=device)
    step_s = [[] for _ in gnns]
    for i, g_ in enumerate(gnns):           # time every layer's training steps inside fit() (a synchronisation per step: the steps are 1 000 graphs each)
        inner = g_.train_step
        def timed(data, _inner=inner, _acc=step_s[i], **kw):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            r = _inner(data, **kw)
            torch.cuda.synchronize(); _acc.append(time.perf_counter() - t0)
            return r
        g_.train_step = timed
    torch.cuda.synchronize(); t0 = time.perf_counter()
    hists = lg.fit(tr, epochs=1, validation_data=va, verbose=0)
    torch.cuda.synchronize(); t_first = time.perf_counter() - t0
    for acc in step_s: acc.clear()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    hists = lg.fit(tr, epochs=1, validation_data=va, verbose=0)
    torch.cuda.synchronize(); t_fit = time.perf_counter() - t0
    return {'fit_one_epoch_ms': 1e3 * t_fit, 'first_fit_ms_with_operator_builds': 1e3 * t_first,
            'train_step_ms_per_layer': [1e3 * float(np.median(a)) if a else None for a in step_s], 'steps_per_layer': [len(a) for a in step_s],
            'final_layer_loss': float(hists[-1]['loss'][-1]),
            'workload': f'MUTAG starter split ({len(tr.data)} training / {len(va.data)} validation graphs), {layers} layers, serial, get_state + get_output, '
                        f'state_vect_dim 0, 5 iterations, threshold 0.01, batches of {bs}; an epoch = per layer: {len(tr)} training steps + validation, then '
                        f'every graph alone through the layer (batch size 1, training-mode forward) to relabel the next layer\'s graphs'}


def composite_training_section(device):
    """Heterogeneous small graphs (reference CompositeGNN.py:275-304 `train_step`): 640 typed graphs of 20 .. 60 nodes (3 node types,
    MUTAG-like size), batches of 32, node-focused, d = 32 x 20 iterations - the step runs one state network per node type on that
    type's rows inside the library (`gnn_train_step` with `composite`, round 4), and for comparison on the device building blocks
    driven from Python (`Models/training.py`, what round 3 ran)."""
    from gnnkeras_amd.synth import er_composite_graph
    from gnnkeras_amd.Models.MLP import MLP, get_inout_dims
    from gnnkeras_amd.Models.CompositeGNN import CompositeGNNnodeBased
    from gnnkeras_amd.Models.training import Adam
    from gnnkeras_amd.Sequencers.GraphSequencers import CompositeMultiGraphSequencer
    rng = np.random.default_rng(7)
    dims, d, it = (14, 8, 4), 32, 20
    graphs = [er_composite_graph(int(n), int(2.2 * n), dim_node_label=dims, seed=100 + i) for i, n in enumerate(rng.integers(20, 61, 640))]
    inp, lay = get_inout_dims('state', dims, 3, 2, 'n', d)
    nets = [MLP(i, lay, 'selu', 'lecun_normal', 'lecun_normal', rng=t, device=device) for t, i in enumerate(inp)]
    inp, lay = get_inout_dims('output', dims, 3, 2, 'n', d)
    no = MLP(inp[0], lay, 'softmax', 'glorot_normal', 'glorot_normal', rng=9, device=device)
    gnn = CompositeGNNnodeBased(nets, no, d, it, 0.01)
    gnn.compile(optimizer=Adam(0.01), loss='categorical_crossentropy', metrics=['accuracy'])
    seq = CompositeMultiGraphSequencer(graphs, 'n', 'average', 32, shuffle=False, device=device)
    for i in range(len(seq)): gnn.train_step(seq[i], seed=0)           # (every batch once: its device operators are built on first use)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(len(seq)): r = gnn.train_step(seq[i], seed=0)
    torch.cuda.synchronize(); t_step = (time.perf_counter() - t0) / len(seq)
    native = gnn._trainer._native_step_applies(seq[0][1])
    torch.cuda.synchronize(); t0 = time.perf_counter()
    gnn.fit(seq, epochs=1, verbose=0)
    torch.cuda.synchronize(); t_epoch = time.perf_counter() - t0
    gnn.predict(seq)
    t0 = time.perf_counter(); gnn.predict(seq); torch.cuda.synchronize(); t_pred = time.perf_counter() - t0
    gnn._trainer.use_native_step = False                                # the same step on the building blocks driven from Python (round 3's path)
    for i in range(3): gnn.train_step(seq[i], seed=0)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(len(seq)): gnn.train_step(seq[i], seed=0)
    torch.cuda.synchronize(); t_blocks = (time.perf_counter() - t0) / len(seq)
    # the same graphs ARC-focused (reference CompositeGNN.py:315-327: the output network over [state_src | state_dst | arc label] of every arc):
    # in the library since round 6
    from gnnkeras_amd.Models.CompositeGNN import CompositeGNNarcBased
    graphs_a = [er_composite_graph(int(g.nodes.shape[0]), int(g.arcs.shape[0]), dim_node_label=dims, seed=100 + i, focus='a') for i, g in enumerate(graphs)]
    inp, lay = get_inout_dims('state', dims, 3, 2, 'a', d)
    nets_a = [MLP(i, lay, 'selu', 'lecun_normal', 'lecun_normal', rng=t, device=device) for t, i in enumerate(inp)]
    inp, lay = get_inout_dims('output', dims, 3, 2, 'a', d)
    no_a = MLP(inp[0], lay, 'softmax', 'glorot_normal', 'glorot_normal', rng=9, device=device)
    gnn_a = CompositeGNNarcBased(nets_a, no_a, d, it, 0.01)
    gnn_a.compile(optimizer=Adam(0.01), loss='categorical_crossentropy', metrics=['accuracy'])
    seq_a = CompositeMultiGraphSequencer(graphs_a, 'a', 'average', 32, shuffle=False, device=device)
    for i in range(len(seq_a)): gnn_a.train_step(seq_a[i], seed=0)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(len(seq_a)): gnn_a.train_step(seq_a[i], seed=0)
    torch.cuda.synchronize(); t_arc = (time.perf_counter() - t0) / len(seq_a)
    arc_native = gnn_a._trainer._native_step_applies(seq_a[0][1])
    return {'train_step_ms_per_batch': 1e3 * t_step, 'k': int(r['k']), 'fit_epoch_ms': 1e3 * t_epoch, 'predict_ms': 1e3 * t_pred,
            'arc_focused_train_step_ms_per_batch': 1e3 * t_arc, 'arc_focused_in_library_step': bool(arc_native),
            'in_library_step': bool(native), 'train_step_ms_per_batch_building_blocks_from_python': 1e3 * t_blocks,
            'workload': f'{len(graphs)} heterogeneous graphs (3 node types, 20..60 nodes), {len(seq)} batches of 32, node-focused, d = {d}, '
                        f'max_iteration = {it}; gnn_train_step with per-type networks (csrc/train_composite.hpp), predict() batch by batch'}


def measure_loop(gnn, inputs, s0, steps, warmup):
    """(elapsed seconds of `steps` forwards, k, seconds per iteration kernel launch) on one GPU; the per-launch time comes from
    HIP events the library records on the launch stream around the iteration launches (gnn.loop_events)."""
    ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
    for e in ev: e.record()
    gnn.loop_events = ev
    step = lambda: gnn.Loop(*inputs, state0=s0)
    for _ in range(warmup): step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps): k, state, out = step()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    k_val = gnn.check_last_k()
    t_loop_ms = []
    for _ in range(5):
        step(); torch.cuda.synchronize()
        t_loop_ms.append(ev[0].elapsed_time(ev[1]))
    gnn.loop_events = None
    return elapsed, k_val, 1e-3 * float(np.median(t_loop_ms)) / max(k_val, 1)


def roofline_record(b_iter, t_iter, kernel_name, n_nodes=None, h1=None):
    achieved = b_iter / t_iter
    rec = {'bound': 'hbm', 'achieved': achieved / 1e9, 'peak': HBM_PEAK / 1e9, 'unit': 'GB/s', 'frac': achieved / HBM_PEAK,
           'traffic': None, 'kernel': kernel_name, 'algorithmic_bytes_per_launch': b_iter, 'avg_launch_us': 1e6 * t_iter}
    if kernel_name.startswith('k_state_fused4') and kernel_name.endswith(',true>') and n_nodes and h1:
        # the XC form of the kernel reads 128 B of constant inputs per node where SURVEY's per-unit figure counts the 4*H1 bytes
        # of the constant C: `achieved` / `frac` stay on SURVEY's algorithmic bytes (the work done), this is what it really moves
        moved = b_iter - n_nodes * (4 * h1 - 128)
        rec['bytes_the_kernel_requests_per_launch'] = moved
        rec['frac_on_requested_bytes'] = moved / t_iter / HBM_PEAK
    return rec


def same_kernel(profiled, reported):
    """rocprofv3 prints every template argument of an instantiation, the defaulted trailing ones too
    (`k_state_fused4<64,false,4,4,false,false,true,1,false>`), the library's gnn_last_kernel_name() only the ones it selects
    (`k_state_fused4<64,false,4,4,false,false,true>`): the same kernel when one argument list is a prefix of the other."""
    if not profiled or not reported: return False
    a, b = (x.replace(' ', '').rstrip('>') for x in (profiled, reported))
    if '<' not in a or '<' not in b: return a == b
    return a == b or a.startswith(b + ',') or b.startswith(a + ',')


def traffic_lookup(rec, kernel_name, N, E, d, traffic_file=None):
    """HBM bytes per launch from the PMC counters (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over the same workload,
    scripts/parse_pmc.py -> profiles/hbm_traffic.json): recorded per (kernel, workload, library sources), used only when all three
    match this run - a record taken on other sources of csrc/ (its `library_source_hash` differs from
    gnnkeras_amd._native.source_hash()) is NOT this binary's traffic: `traffic` stays null and `traffic_null_reason` says why."""
    from gnnkeras_amd._native import source_hash
    traffic_file = traffic_file or os.path.join(ROOT, 'profiles', 'hbm_traffic.json')
    if not os.path.exists(traffic_file):
        rec['traffic_null_reason'] = 'profiles/hbm_traffic.json is missing'
        return rec
    here = source_hash()
    rec['library_source_hash'] = here
    try:
        stale = None
        for tr in json.load(open(traffic_file)).get('records', []):
            if same_kernel(tr.get('kernel'), kernel_name) and tr.get('nodes') == N and tr.get('arcs') == E and tr.get('state_dim') == d:
                if tr.get('library_source_hash') != here:
                    stale = tr.get('library_source_hash')
                    continue
                rec['traffic'] = tr['hbm_bytes_per_launch']
                rec['traffic_counts'] = ('L2 -> fabric bytes per launch: 2 x FETCH_SIZE + WRITE_SIZE (the gfx950 correction of MI355X_MICROARCH.md); '
                                         'requests served by the 256 MiB Infinity Cache are INCLUDED - this is not an HBM-only figure')
                rec['traffic_bounds'] = tr.get('bounds')
                rec['traffic_source'] = (f"profiles/hbm_traffic.json: {tr.get('launches')} launches of this kernel on this workload, "
                                         f"sources {here}, {tr.get('taken', 'PMC passes of the round')}")
        if rec.get('traffic') is None:
            rec['traffic_null_reason'] = (f'the PMC record of this kernel and workload was taken on other library sources ({stale} != {here})'
                                          if stale is not None else 'no PMC record for this kernel and workload')
    except Exception as e:
        rec['traffic_null_reason'] = f'profiles/hbm_traffic.json unreadable: {e}'
    return rec


SIDE_STEPS = 5          # timed forwards of the side sections (beyond_infinity_cache, wide_state_d200), after one warm-up


def beyond_cache_section(device, d, K_it, aggregation):
    """The same model on a 4 M-node / 40 M-arc ER graph: state array 1 GB (> the 256 MiB Infinity Cache), operands built on
    the device (gnnkeras_amd.synth.er_device_batch)."""
    from gnnkeras_amd import _native as nat
    from gnnkeras_amd.synth import er_device_batch
    from gnnkeras_amd.Models.GNN import GNNnodeBased
    N, E = 4_000_000, 40_000_000
    x = er_device_batch(N, E, device, aggregation_mode=aggregation, seed=1234)
    ns, no = starter_nets(d, device)
    gnn = GNNnodeBased(ns, no, d, K_it, 0.0)
    gen = torch.Generator(device=device); gen.manual_seed(1)
    s0 = torch.randn((N, d), generator=gen, device=device) * 0.1
    elapsed, k_val, t_iter = measure_loop(gnn, gnn.process_inputs(x), s0, steps=SIDE_STEPS, warmup=1)
    b_iter = algorithmic_bytes_per_iteration(N, E, d, ns.units[0], False)
    kernel_name = nat.lib().gnn_last_kernel_name().decode()
    rec = roofline_record(b_iter, t_iter, kernel_name, N, ns.units[0])
    rec.update({'workload': f'Erdos-Renyi {N} nodes / {E} arcs, state_dim={d}, k={k_val:g}, {aggregation} aggregation '
                            f'(state array {N * d * 4 / 2**20:.0f} MiB: does not fit the 256 MiB Infinity Cache)',
                'updates_per_s': E * k_val * SIDE_STEPS / elapsed, 'fwd_ms': 1e3 * elapsed / SIDE_STEPS, 'steps': SIDE_STEPS, 'warmup': 1,
                'launches_timed': int(k_val) * SIDE_STEPS})
    return traffic_lookup(rec, kernel_name, N, E, d)


def wide_state_section(device, aggregation):
    """State widths 129 .. 256 (k_state_xwide_b3, kernel_state_xwide.hpp): d = 200 on a 300 k-node / 3 M-arc ER graph built on the device, time per
    iteration launch from the library's HIP events around the loop, roofline on the algorithmic bytes of an iteration."""
    from gnnkeras_amd import _native as nat
    from gnnkeras_amd.synth import er_device_batch
    from gnnkeras_amd.Models.GNN import GNNnodeBased
    N, E, d = 300_000, 3_000_000, 200
    x = er_device_batch(N, E, device, aggregation_mode=aggregation, seed=77)
    ns, no = starter_nets(d, device)
    gen = torch.Generator(device=device); gen.manual_seed(2)
    s0 = torch.randn((N, d), generator=gen, device=device) * 0.1
    gnn = GNNnodeBased(ns, no, d, 20, 0.0)
    elapsed, k_val, t_iter = measure_loop(gnn, gnn.process_inputs(x), s0, steps=SIDE_STEPS, warmup=1)      # t_iter: HIP events around the iteration launches
    b_iter = algorithmic_bytes_per_iteration(N, E, d, ns.units[0], False)
    kernel_name = nat.lib().gnn_last_kernel_name().decode()
    rec = {'workload': f'Erdos-Renyi {N} nodes / {E} arcs, state_dim={d}, {aggregation} aggregation', 'k': k_val, 'us_per_iteration': 1e6 * t_iter,
           'algorithmic_bytes_per_iteration': b_iter, 'frac_of_8TBps': b_iter / t_iter / 8e12, 'kernel': kernel_name, 'steps': SIDE_STEPS, 'warmup': 1,
           'launches_timed': int(k_val) * SIDE_STEPS, 'traffic': None}
    return traffic_lookup(rec, kernel_name, N, E, d)


def gpu_state():
    """Clocks / power / temperature rocm-smi reports right after the timed region (a child process; never fails the bench): the same
    kernel has measured 460 us on one box of the pool and 533 us on another (profiles/r03_notes.txt) - this says which kind ran."""
    import subprocess, sys
    # Under a profiler the tool's preloaded library initialises the GPU in every child before its program starts, and rocm-smi is a
    # `#!/usr/bin/env python3` script: that second exec is one this pool refuses.  Skip there; elsewhere start the interpreter on the
    # script directly (one exec of a child that has not touched the GPU) with the preload variables removed.
    if any('rocprof' in str(os.environ.get(k, '')).lower() for k in ('LD_PRELOAD', 'ROCP_TOOL_LIBRARIES', 'HSA_TOOLS_LIB', 'ROCPROFILER_LIBRARY')):
        return {'unavailable': 'skipped under rocprofv3'}
    try:
        import shutil
        smi = shutil.which('rocm-smi') or '/opt/rocm/bin/rocm-smi'
        env = {k: v for k, v in os.environ.items() if k not in ('LD_PRELOAD', 'HSA_TOOLS_LIB', 'ROCP_TOOL_LIBRARIES')}
        out = subprocess.run([sys.executable, os.path.realpath(smi), '--showclocks', '--showpower', '--showtemp', '--json'], capture_output=True, text=True,
                             timeout=20, env=env).stdout
        card = next(iter(json.loads(out).values()))
        keep = {k: v for k, v in card.items() if any(w in k.lower() for w in ('mclk', 'fclk', 'sclk', 'power', 'junction', 'memory)'))}
        return keep or None
    except Exception as e:
        return {'unavailable': str(e)[:80]}


def host_rss_mb():
    with open('/proc/self/status') as f:
        for line in f:
            if line.startswith('VmRSS:'): return int(line.split()[1]) / 1024.0
    return None


def composite_model(d, K_it, device, dims=(14, 8, 4)):
    """BASELINE C5's model: 3 node types with label widths `dims`, one BN + Dense state network per type."""
    from gnnkeras_amd.Models.CompositeGNN import CompositeGNNnodeBased
    from gnnkeras_amd.Models.MLP import MLP, get_inout_dims
    inp, lay = get_inout_dims('state', dims, 3, 2, 'n', d)
    nets_s = [MLP(i, lay, 'selu', 'lecun_normal', 'lecun_normal', rng=t, device=device) for t, i in enumerate(inp)]
    inp, lay = get_inout_dims('output', dims, 3, 2, 'n', d)
    no = MLP(inp[0], lay, 'softmax', 'glorot_normal', 'glorot_normal', rng=9, device=device)
    return CompositeGNNnodeBased(nets_s, no, d, K_it, 0.0), nets_s[0], dims


def emulate_shard(args):
    """`--emulate-shard r/R`: rank r's share of an R-GPU run of the selected workload on ONE GPU.  The shard is built the way a
    rank of the real job builds it (from its own `GraphSlice`), runs the real shard kernels against a full-size
    [R x (N / R + 1), SP] state buffer and never talks to anyone: the exchange is the half of an iteration this cannot
    measure (DESIGN.md 6 prices it from the xGMI link rate)."""
    r, R = (int(v) for v in args.emulate_shard.split('/'))
    if not (0 <= r < R): raise SystemExit('--emulate-shard r/R needs 0 <= r < R')
    if not torch.cuda.is_available(): raise SystemExit('bench.py needs an MI355X: the message-passing loop has no CPU path')
    device = torch.device('cuda', 0)
    torch.cuda.set_device(0)
    from gnnkeras_amd import _native as nat
    from gnnkeras_amd.distributed import ShardedLoop, partition
    from gnnkeras_amd.synth import er_graph_slice, er_composite_graph_slice
    from gnnkeras_amd.Models.GNN import GNNnodeBased
    if args.workload not in ('c4', 'c3', 'c5'): raise SystemExit('--emulate-shard: workloads c4 / c3 / c5')
    sizes = {'c4': (1e6, 1e7), 'c3': (1e5, 1e6), 'c5': (5e5, 5e6)}[args.workload]
    N, E = int(args.nodes or sizes[0]), int(args.arcs or sizes[1])
    d, K_it = args.state_dim, args.max_iteration
    rss0 = host_rss_mb()
    t0 = time.perf_counter()
    chunk, ranges = partition(N, R)
    if args.workload == 'c5':            # BASELINE C5: 3 node types, per-type state networks; the rank generates its own slice
        gnn, ns, dims = composite_model(d, K_it, device)
        gs = er_composite_graph_slice(N, E, *ranges[r], dim_node_label=dims, aggregation_mode=args.aggregation, seed=1234)
    else:
        ns, no = starter_nets(d, device)
        gnn = GNNnodeBased(ns, no, d, K_it, 0.0)
        gs = er_graph_slice(N, E, *ranges[r], aggregation_mode=args.aggregation, seed=1234)
    t_slice = time.perf_counter() - t0
    sl = ShardedLoop(gnn, gs, r, R, device, overlap=not args.no_overlap)
    if args.pipeline_chunks > 1 and sl.set_pipeline(args.pipeline_chunks) != args.pipeline_chunks:
        raise SystemExit('--pipeline-chunks: this shard does not take chunk launches (composite model, no overlap split, hub rows)')
    if args.native_loop: sl.enable_native_loop(emulated=True)
    torch.cuda.synchronize()
    t_plan = time.perf_counter() - t0
    gen = torch.Generator(device=device); gen.manual_seed(1)
    s0 = torch.randn((N, d), generator=gen, device=device) * 0.1
    prof = sl.profile_iteration(s0, reps=max(args.steps, 10), collective=False)
    kernel_name = nat.lib().gnn_last_kernel_name().decode()
    b_iter = algorithmic_bytes_per_iteration(sl.n_local, sl.e_local, d, ns.units[0], sl.per_arc_weights)
    rec = roofline_record(b_iter, prof['kernel_s'], kernel_name, sl.n_local, ns.units[0])
    print(json.dumps({
        'emulated_shard': f'{r}/{R}', 'workload': f'{args.workload.upper()} Erdos-Renyi {N} nodes / {E} arcs, state_dim={d}, {args.aggregation}',
        'n_local': sl.n_local, 'e_local': sl.e_local, 'e_own_range': getattr(sl, 'e_own', None), 'overlap_split': bool(sl.overlap),
        'pipeline_chunks': sl.pipeline_chunks, 'loop_driver': 'native (gnn_shard_loop)' if sl.native_loop else 'interpreter',
        'host_issue_us_per_iteration': 1e6 * prof['host_issue_s'],
        'per_iteration_ms': {'kernel': 1e3 * prof['kernel_s'],
                             'note': 'own-range partial + halo kernel (or the one fused kernel with --no-overlap) of rank r, gates open, '
                                     'no collective: the exchange is not measurable on one GPU'},
        'exchange_bytes_per_rank_per_iteration': sl.exchange_bytes(),
        'plan_build_s': t_plan, 'slice_generation_s': t_slice, 'host_rss_mb': host_rss_mb(), 'host_rss_mb_before_plan': rss0,
        'roofline': rec}), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--workload', choices=['c4', 'c3', 'c5', 'c4x4'], default='c4',
                    help='c4: ER 1M/10M (headline); c3: ER 100k/1M; c5: composite ER 500k/5M, 3 node types; '
                         'c4x4: ER 4M/40M built on the device (state array beyond the Infinity Cache)')
    ap.add_argument('--nodes', type=float, default=None)
    ap.add_argument('--arcs', type=float, default=None)
    ap.add_argument('--state-dim', type=int, default=64)
    ap.add_argument('--max-iteration', type=int, default=50)
    ap.add_argument('--aggregation', default='average')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-mutag', action='store_true')
    ap.add_argument('--mutag-dp', action='store_true', help='N>1: also run the MUTAG data-parallel side section (collective; a rank that fails in it ends the job)')
    ap.add_argument('--no-beyond-cache', action='store_true')
    ap.add_argument('--no-training', action='store_true', help='skip the train_step / fit() section')
    ap.add_argument('--unfused', action='store_true')
    ap.add_argument('--force-sharded', action='store_true', help='run the N>1 code path even with one rank (smoke test)')
    ap.add_argument('--exchange', choices=['auto', 'allgather', 'direct', 'halo', 'peer'], default='auto',
                    help='N>1 state exchange: whole slices by RCCL all-gather / by concurrent point-to-point pairs, or compacted halos '
                         '(all-to-all); auto = halo for graphs with locality, else the faster whole-slice transport, measured at start-up; '
                         'peer = the iteration kernel stores its rows into the peers\' IPC-mapped buffers (opt-in: never run on links)')
    ap.add_argument('--no-overlap', action='store_true', help='N>1: exchange strictly after the iteration kernel (no own-range / halo split)')
    ap.add_argument('--native-loop', action='store_true',
                    help='N>1 / --emulate-shard: drive the iterations from the library (gnn_shard_loop: one C call, the exchange over the RCCL C API) '
                         'instead of from the interpreter')
    ap.add_argument('--exchange-pipeline', default='off',
                    help="N>1: 'off' (default: one halo-kernel launch + one exchange per iteration, the validated path), 'auto' (1 / 2 / 4 chunk "
                         "launches timed at the first forward, a chunked count kept only when > 5 %% faster) or a chunk count")
    ap.add_argument('--pipeline-chunks', type=int, default=1,
                    help='--emulate-shard: launch the halo kernel in this many chunk launches (the pipelined exchange, distributed.py set_pipeline)')
    ap.add_argument('--emulate-shard', default=None, metavar='r/R',
                    help="one GPU, no process group: build rank r's plan of an R-rank job from its own graph slice and time its "
                         "iteration kernels (own-range partial + halo kernel) against a full-size state buffer; prints plan-build "
                         "seconds, host RSS and the per-iteration kernel time - the compute half of the N-GPU prediction in DESIGN.md 6")
    args = ap.parse_args()
    if args.emulate_shard:
        return emulate_shard(args)

    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        # Start the N ranks as a child torch.distributed.run.  Nothing in this process has touched the GPU (importing torch
        # does not), and the child is a fresh process tree: no exec from a GPU-initialised process.
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(('127.0.0.1', 0)); port = sk.getsockname()[1]
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={args.gpus}',
               '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world != args.gpus:
        raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE={world}: the launcher started a different number of ranks')
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X: the message-passing loop has no CPU path')
    torch.cuda.set_device(local_rank)
    device = torch.device('cuda', local_rank)
    sharded = world > 1 or args.force_sharded
    if sharded:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29511')
        dist.init_process_group('nccl', rank=rank, world_size=world, device_id=device)

    from gnnkeras_amd import _native as nat
    from gnnkeras_amd.synth import er_graph, er_composite_graph, er_device_batch
    from gnnkeras_amd.Models.GNN import GNNnodeBased
    from gnnkeras_amd.Models.CompositeGNN import CompositeGNNnodeBased
    from gnnkeras_amd.Models.MLP import MLP, get_inout_dims
    from gnnkeras_amd.Sequencers.GraphSequencers import MultiGraphSequencer, CompositeMultiGraphSequencer

    sizes = {'c4': (1e6, 1e7), 'c3': (1e5, 1e6), 'c5': (5e5, 5e6), 'c4x4': (4e6, 4e7)}[args.workload]
    N, E = int(args.nodes or sizes[0]), int(args.arcs or sizes[1])
    d, K_it = args.state_dim, args.max_iteration
    composite = args.workload == 'c5'
    on_device = args.workload == 'c4x4'
    if on_device and sharded: raise SystemExit('--workload c4x4 is a single-GPU point (operands are built on one device)')
    graph_is_slice = False
    t_graph0 = time.perf_counter()
    if composite:
        gnn, ns, dims = composite_model(d, K_it, device)
        # a rank of the sharded run generates only ITS slice of the heterogeneous graph (the compacted halo exchange wants the whole)
        graph_is_slice = sharded and args.exchange != 'halo'
        if graph_is_slice:
            from gnnkeras_amd.distributed import partition
            from gnnkeras_amd.synth import er_composite_graph_slice
            graph = er_composite_graph_slice(N, E, *partition(N, world)[1][rank], dim_node_label=dims, aggregation_mode=args.aggregation, seed=1234)
        else:
            graph = er_composite_graph(N, E, dim_node_label=dims, aggregation_mode=args.aggregation, seed=1234)
        Sequencer = CompositeMultiGraphSequencer
    else:
        # a rank of the sharded run generates only ITS slice of the graph (exchange 'auto' / 'allgather' / 'direct'); the compacted
        # halo exchange derives every peer's pack lists and so needs the replicated GraphObject
        graph_is_slice = sharded and args.exchange != 'halo'
        if graph_is_slice:
            from gnnkeras_amd.distributed import partition
            from gnnkeras_amd.synth import er_graph_slice
            graph = er_graph_slice(N, E, *partition(N, world)[1][rank], aggregation_mode=args.aggregation, seed=1234)
        else:
            graph = None if on_device else er_graph(N, E, aggregation_mode=args.aggregation, seed=1234)
        ns, no = starter_nets(d, device)
        gnn = GNNnodeBased(ns, no, d, K_it, 0.0)
        Sequencer = MultiGraphSequencer
    t_graph = time.perf_counter() - t_graph0
    if args.unfused: gnn.native_flags = nat.FLAG_UNFUSED
    if on_device:
        gen = torch.Generator(device=device); gen.manual_seed(1)
        s0 = torch.randn((N, d), generator=gen, device=device) * 0.1
        s0_host = None
    else:
        s0_host = np.random.default_rng(1).normal(0, 0.1, (N, d)).astype(np.float32)
        s0 = torch.from_numpy(s0_host).to(device)

    extra = {}
    if not sharded:
        x = er_device_batch(N, E, device, aggregation_mode=args.aggregation, seed=1234) if on_device else \
            Sequencer([graph], 'n', args.aggregation, 1, shuffle=False, device=device)[0][0]
        inputs = gnn.process_inputs(x)
        per_arc_w = inputs[7 if composite else 5].device_csr(device)['w'] is not None
        elapsed, k_val, t_iter = measure_loop(gnn, inputs, s0, args.steps, args.warmup)
        kernel_name = nat.lib().gnn_last_kernel_name().decode()
    else:
        import torch.distributed as dist
        from gnnkeras_amd.distributed import make_sharded_loop
        t_plan0 = time.perf_counter()
        sl = make_sharded_loop(gnn, graph, rank=rank, world_size=world, device=device, exchange=args.exchange,
                               overlap=not args.no_overlap, pipeline=args.exchange_pipeline)
        if args.force_sharded and world == 1 and hasattr(sl, 'force_exchange'): sl.force_exchange = True      # (the collective is issued although there is nobody to talk to: its HOST cost is real)
        if args.native_loop and hasattr(sl, 'enable_native_loop') and type(sl)._layout == 'allgather': sl.enable_native_loop(with_comm=True)
        extra['loop_driver'] = 'native (gnn_shard_loop)' if getattr(sl, 'native_loop', False) else 'interpreter'
        torch.cuda.synchronize()
        extra['plan_build_s'] = (time.perf_counter() - t_plan0) + t_graph
        extra['host_rss_mb'] = host_rss_mb()
        extra['graph_source'] = 'per-rank GraphSlice (own destination range only)' if graph_is_slice else 'replicated GraphObject'
        step = lambda: sl.forward(s0)
        per_arc_w = sl.per_arc_weights

        def sync_all():
            torch.cuda.synchronize()
            dist.barrier()
            torch.cuda.synchronize()

        for _ in range(args.warmup): step()
        sync_all()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            k, state, out = step()
        sync_all()
        elapsed = time.perf_counter() - t0
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t)
        k_val = float(k)
        prof = sl.profile_iteration(s0)            # per-iteration device time: kernel(s) alone, exchange alone, both overlapped
        t_iter = prof['kernel_s']
        kernel_name = nat.lib().gnn_last_kernel_name().decode()
        tt = torch.tensor([prof['kernel_s'], prof['exchange_s'], prof['iteration_s']], dtype=torch.float64, device=device)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        extra['per_iteration_ms'] = {'kernel': 1e3 * float(tt[0]), 'exchange': 1e3 * float(tt[1]),
                                     'iteration_overlapped': 1e3 * float(tt[2]),
                                     'note': 'max over ranks; kernel = own-range + halo launches without the collective, exchange = '
                                             'the collective alone, iteration = what one iteration costs with both in flight'}
        if 'host_issue_s' in prof: extra['host_issue_us_per_iteration'] = 1e6 * prof['host_issue_s']
        extra['exchange_bytes_per_rank_per_iteration'] = sl.exchange_bytes()
        extra['exchange_transport'] = getattr(sl, 'transport', 'all_to_all')
        if getattr(sl, 'transport_times', None): extra['exchange_transport_ms_measured'] = {k_: 1e3 * v for k_, v in sl.transport_times.items()}
        extra['exchange_pipeline_chunks'] = getattr(sl, 'pipeline_chunks', 1)          # > 1: chunk launches, every chunk's rows sent as soon as written
        if getattr(sl, 'pipeline_times', None): extra['exchange_pipeline_ms_per_iteration_measured'] = {str(k_): 1e3 * v for k_, v in sl.pipeline_times.items()}
    ms_per_step = 1e3 * elapsed / args.steps
    value = E * k_val * args.steps / elapsed
    if rank == 0: extra['gpu_state_after_timed_region'] = gpu_state()

    h1 = ns.units[0]
    n_local = N if not sharded else sl.n_local
    e_local = E if not sharded else sl.e_local
    b_iter = algorithmic_bytes_per_iteration(n_local, e_local, d, h1, per_arc_w)
    roofline = roofline_record(b_iter, t_iter, kernel_name, n_local, h1)
    if not sharded: traffic_lookup(roofline, kernel_name, N, E, d)

    result = {
        'metric': 'node-state updates/s (arcs x iterations / s), forward Loop',
        'value': value, 'unit': 'arc-updates/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': ms_per_step, 'higher_is_better': True, 'scaling': 'strong', 'vs_baseline': None,
        'dtype': 'f32', 'data': 'synthetic',
        'config': {'workload': f'{args.workload.upper()} Erdos-Renyi {N} nodes / {E} arcs, state_dim={d}, max_iteration={K_it}, '
                               f'threshold=0 (k={k_val:g}), node-focused, {args.aggregation} aggregation, '
                               + ('3 node types with per-type ' if composite else '') +
                               f'BN+Dense({ns.input_dim}->{h1},selu) state net',
                   'sharding': 'single GPU' if not sharded else f'node-range shards over {world} GPUs, '
                                                               f'RCCL {"all-to-all of compacted halos" if type(sl).__name__ == "HaloShardedLoop" else ("all-gather of state slices" if sl.transport == "ring" else "pair-wise send/recv of state slices (one-hop all-gather)")} per iteration'
                                                               + (', own-range arcs overlapped with the exchange' if sharded and sl.overlap else '')},
        'roofline': roofline,
        'loop_only_updates_per_s': E / t_iter if not sharded else None,
        'fwd_ms_per_graph': ms_per_step,
    }
    result.update(extra)

    if rank == 0 and not sharded and not args.no_cpu_baseline and not composite and not on_device:
        from oracle import cpu_baseline as cb
        from oracle.harness import _np, _triple
        ops = (_np(x[0]), _np(x[1]), _triple(x[5]), _triple(x[6]), ns.spec(), d, 0.0, s0_host)
        n_thr = torch.get_num_threads()
        variants = {'torch_all': cb.time_torch(*ops, threads=n_thr, warmup=3, max_timed=10, budget_s=12.0),
                    'torch_1': cb.time_torch(*ops, threads=1, warmup=1, max_timed=5, budget_s=8.0),
                    'numpy_1': cb.time_numpy(*ops, warmup=1, max_timed=5, budget_s=8.0)}
        for v in variants.values(): v['arc_updates_per_s'] = E / v['median_iter_s']
        best = max(variants, key=lambda n: variants[n]['arc_updates_per_s'])
        result['cpu_baseline'] = {'value': variants[best]['arc_updates_per_s'], 'unit': 'arc-updates/s',
                                  'cores': variants[best]['threads'], 'kind': 'port',
                                  'sample': f'median of {variants[best]["timed_iterations"]} iterations (after '
                                            f'{variants[best]["warmup_iterations"]} warm-up) of the same {args.workload.upper()} graph, loop '
                                            f'only, variant {best} = the fastest of torch on {n_thr} threads / torch on 1 thread / '
                                            f'NumPy+SciPy on 1 thread; restatement of the TF op sequence (not TensorFlow), '
                                            f'{os.cpu_count()} host cpus',
                                  'variants': variants}
        result['speedup_vs_cpu_loop'] = (E / t_iter) / result['cpu_baseline']['value']
    if rank == 0 and not sharded and not args.no_mutag and args.workload == 'c4':
        result['mutag'] = mutag_section(device, cpu=not args.no_cpu_baseline)
    if rank == 0 and not sharded and not args.no_training and args.workload == 'c4' and not args.unfused and args.state_dim in (16, 32, 64):
        try:
            result['training'] = training_section(device, (x,), d)
        except Exception as e:                        # never lose the headline line to the extras
            result['training'] = {'error': str(e)[:300]}
    if rank == 0 and not sharded and not args.no_beyond_cache and args.workload == 'c4' and not args.unfused:
        del gnn, inputs, x, s0
        torch.cuda.empty_cache()
        result['beyond_infinity_cache'] = beyond_cache_section(device, d, K_it, args.aggregation)
        # Both fractions in the headline object: `frac` is measured on the C4 graph, whose 256 MB state array sits in the 256 MiB Infinity Cache
        # (a last-level-cache-assisted figure: it can exceed what HBM alone sustains); the same kernel on a 4 M-node graph (state array 1 GB)
        # is the HBM-bound figure.
        result['roofline']['frac_state_beyond_infinity_cache'] = result['beyond_infinity_cache'].get('frac')
        result['roofline']['frac_note'] = ('frac: BASELINE C4 (1 M nodes: the state array fits the Infinity Cache); frac_state_beyond_infinity_cache: '
                                          'the same kernel on 4 M nodes / 40 M arcs (state array 1 GB), both on algorithmic bytes over 8 TB/s')
        torch.cuda.empty_cache()
        try:
            result['wide_state_d200'] = wide_state_section(device, args.aggregation)
        except Exception as e:                        # never lose the headline line to the extras
            result['wide_state_d200'] = {'error': str(e)[:300]}

    # The MUTAG data-parallel record is a COLLECTIVE side section: a rank that fails inside it leaves the others waiting in some collective,
    # and no new collective can be issued from an exception path to agree on the failure (it would pair with whatever collective
    # the others are in).  So at world size > 1 it runs only when asked for (--mutag-dp), after everything the headline line needs has been
    # measured, and a failing rank exits non-zero - the launcher then tears the job down instead of letting it hang.  At world size 1
    # (--force-sharded on a 1-GPU box) a failure is local and simply recorded.
    if sharded and ((world == 1 and args.force_sharded) or (world > 1 and args.mutag_dp)) and not args.no_mutag and args.workload == 'c4':
        del sl
        torch.cuda.empty_cache()
        try:
            result['mutag_data_parallel'] = mutag_dp_section(device, rank, world)
        except Exception as e:
            if world > 1:
                print(f'bench.py: rank {rank} failed in the MUTAG data-parallel section: {e}', file=sys.stderr, flush=True)
                os._exit(3)
            result['mutag_data_parallel'] = {'error': str(e)[:300]}
    # RCCL writes its version banner through C stdio, which would otherwise reach the pipe AFTER this process' last Python write (at
    # exit): drain it first so that the JSON line is the LAST line of stdout (it is also the only line that starts with '{').
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    if sharded and world > 1:
        torch.distributed.barrier()            # every rank has drained its C stdio before rank 0 writes the line
    if rank == 0:
        print(json.dumps(result), flush=True)
    if sharded:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
