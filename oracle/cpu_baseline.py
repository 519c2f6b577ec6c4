"""CPU BASELINE TIMER — TEST / BENCH INFRASTRUCTURE ONLY (see gnn_oracle.py header; Loop parity is UNPINNED by the
reference).  Used by bench.py's `cpu_baseline` leg and nothing else.

Times ONE ITERATION of the reference's un-fused op sequence (`condition` + `convergence`, GNN/Models/GNN.py:196-236) on the
host, the way SURVEY §8d asks: warm-up iterations, then the median over the timed ones, in three variants

    torch_all : torch CPU ops on every host thread      (intra-op parallel sparse-dense matmul / GEMM / elementwise)
    torch_1   : the same ops on one thread
    numpy_1   : NumPy + SciPy CSR, single thread         (BLAS pinned to one thread where threadpoolctl is importable)

It is a port ("TF-op-sequence restatement, not TensorFlow"): it omits TF's eager per-op dispatch, so it is a faster
stand-in than the reference and the GPU/CPU ratio quoted against it is conservative.  The operands (sparse operator in
CSR form, aggregated labels) are built once outside the timed region, like the reference's `tf.SparseTensor`s.
Each variant stops after `max_timed` iterations or when its time budget is spent (at least 2 timed iterations).
"""
from __future__ import annotations

import time

import numpy as np


def _bn_fold(spec, w, dtype):
    """(BN affine a, c; dense list) of a (spec, weights) network — inference BN: y = x * a + c."""
    w = [np.asarray(x, dtype=dtype) for x in w]
    pos, a, c = 0, None, None
    if spec['batch_normalization']:
        g, b, m, v = w[:4]
        a = g / np.sqrt(v + dtype(1e-3))
        c = b - m * a
        pos = 4
    layers = []
    for act in spec['activations']:
        layers.append((w[pos], w[pos + 1], act))
        pos += 2
    return a, c, layers


def _run(step, warmup, max_timed, budget_s):
    for _ in range(warmup): step()
    ts, t_begin = [], time.perf_counter()
    while len(ts) < max_timed and (len(ts) < 2 or time.perf_counter() - t_begin < budget_s):
        t0 = time.perf_counter()
        step()
        ts.append(time.perf_counter() - t0)
    return {'median_iter_s': float(np.median(ts)), 'min_iter_s': float(np.min(ts)), 'timed_iterations': len(ts),
            'warmup_iterations': warmup}


def time_torch(nodes, arcs, adjacency, arcnode, net_state, state_vect_dim, state_threshold, state0, threads,
               warmup=3, max_timed=10, budget_s=12.0):
    import torch
    from .torch_cpu import _sparse_t, ACT
    old = torch.get_num_threads()
    torch.set_num_threads(int(threads))
    try:
        dtype = torch.float32
        X = torch.from_numpy(np.asarray(nodes)).to(dtype)
        lab = torch.from_numpy(np.asarray(arcs)[:, 2:]).to(dtype)
        At, ANt = _sparse_t(adjacency, dtype), _sparse_t(arcnode, dtype)
        agg_arcs = torch.sparse.mm(ANt, lab)
        agg_nodes = torch.sparse.mm(At, X) if state_vect_dim > 0 else None
        spec, w = net_state
        w = [torch.from_numpy(np.asarray(a)).to(dtype) for a in w]
        st = {'state': torch.from_numpy(np.asarray(state0)).to(dtype) if state_vect_dim > 0 else X.clone()}
        st['old'] = torch.ones_like(st['state'])

        def step():
            s, so = st['state'], st['old']
            dist = torch.sqrt(torch.sum(torch.square(s - so), dim=1))
            norm = torch.sqrt(torch.sum(torch.square(so), dim=1))
            bool(torch.any(dist > state_threshold * norm))                       # the reference's host bool per iteration
            comps = [s, X, torch.sparse.mm(At, s), agg_nodes, agg_arcs] if state_vect_dim > 0 else [s, torch.sparse.mm(At, s), agg_arcs]
            x = torch.cat(comps, dim=1)
            pos = 0
            if spec['batch_normalization']:
                g, b, m, v = w[:4]; pos = 4
                x = torch.nn.functional.batch_norm(x, m, v, g, b, training=False, eps=1e-3)
            for act in spec['activations']:
                x = ACT[act](x @ w[pos] + w[pos + 1]); pos += 2
            st['state'], st['old'] = x, s

        res = _run(step, warmup, max_timed, budget_s)
        res['threads'] = int(threads)
        return res
    finally:
        torch.set_num_threads(old)


def time_numpy(nodes, arcs, adjacency, arcnode, net_state, state_vect_dim, state_threshold, state0,
               warmup=1, max_timed=5, budget_s=12.0):
    import scipy.sparse as sp
    dtype = np.float32

    def csr_t(triple):
        idx, val, shp = triple
        idx = np.asarray(idx).reshape(-1, 2)
        return sp.csr_matrix((np.asarray(val, dtype=dtype).reshape(-1), (idx[:, 1], idx[:, 0])), shape=(int(shp[1]), int(shp[0])))

    X = np.asarray(nodes, dtype=dtype)
    lab = np.asarray(arcs, dtype=dtype)[:, 2:]
    At, ANt = csr_t(adjacency), csr_t(arcnode)
    agg_arcs = ANt @ lab
    agg_nodes = At @ X if state_vect_dim > 0 else None
    a, c, layers = _bn_fold(net_state[0], net_state[1], dtype)
    acts = {'linear': lambda x: x, 'relu': lambda x: np.maximum(x, 0), 'tanh': np.tanh,
            'selu': lambda x: dtype(1.0507009873554805) * np.where(x > 0, x, dtype(1.6732632423543772) * (np.exp(np.minimum(x, 0)) - 1)),
            'sigmoid': lambda x: 1 / (1 + np.exp(-x))}
    st = {'state': np.asarray(state0, dtype=dtype) if state_vect_dim > 0 else X.copy()}
    st['old'] = np.ones_like(st['state'])

    def step():
        s, so = st['state'], st['old']
        dist = np.sqrt(np.sum(np.square(s - so), axis=1))
        norm = np.sqrt(np.sum(np.square(so), axis=1))
        bool(np.any(dist > dtype(state_threshold) * norm))
        comps = [s, X, At @ s, agg_nodes, agg_arcs] if state_vect_dim > 0 else [s, At @ s, agg_arcs]
        x = np.concatenate(comps, axis=1)
        if a is not None: x = x * a + c
        for W, b, act in layers: x = acts[act](x @ W + b)
        st['state'], st['old'] = x.astype(dtype, copy=False), s

    try:
        from threadpoolctl import threadpool_limits
        with threadpool_limits(limits=1):
            res = _run(step, warmup, max_timed, budget_s)
    except ImportError:
        res = _run(step, warmup, max_timed, budget_s)
    res['threads'] = 1
    return res
