"""ORACLE / CPU BASELINE — TEST INFRASTRUCTURE ONLY (see gnn_oracle.py header; Loop parity is UNPINNED by the reference).

A second, independent restatement of the reference's un-fused op sequence with torch CPU ops, so that the CPU baseline
timed next to the GPU path (`bench.py`, `cpu_baseline`) uses every host core the way TensorFlow's CPU kernels would
(intra-op parallel sparse-dense matmul, GEMM and elementwise ops) — the NumPy oracle is single-threaded outside BLAS.
Per iteration, exactly as the reference dispatches them (GNN/Models/GNN.py:196-236):
  condition : sub, square, reduce_sum, sqrt (x2), scalar_mul, greater, reduce_any, less, logical_and  + host bool
  convergence: sparse_dense_matmul(adjoint) ; concat ; BatchNormalization ; matmul + bias ; activation
It is labelled "TF-op-sequence restatement, not TensorFlow": it omits TF's eager per-op dispatch cost, so it is a
faster stand-in and speed-ups quoted against it are conservative.
"""
from __future__ import annotations

import numpy as np
import torch

ACT = {'linear': lambda x: x, 'relu': torch.relu, 'selu': torch.selu, 'tanh': torch.tanh, 'sigmoid': torch.sigmoid,
       'elu': torch.nn.functional.elu, 'softplus': torch.nn.functional.softplus,
       'softmax': lambda x: torch.softmax(x, dim=-1)}


def _sparse_t(triple, dtype):
    """A^T as a torch CSR tensor (built once, outside the timed loop, like the reference's SparseTensor)."""
    idx, val, shp = triple
    idx = np.asarray(idx).reshape(-1, 2)
    At = torch.sparse_coo_tensor(torch.from_numpy(np.ascontiguousarray(idx.T[[1, 0]])),
                                 torch.from_numpy(np.asarray(val, dtype=np.float64).reshape(-1)).to(dtype),
                                 (int(shp[1]), int(shp[0]))).coalesce()
    return At.to_sparse_csr()


def _mlp(net, x, dtype):
    spec, w = net
    w = [torch.from_numpy(np.asarray(a)).to(dtype) for a in w]
    pos = 0
    if spec['batch_normalization']:
        g, b, m, v = w[:4]
        pos = 4
        x = torch.nn.functional.batch_norm(x, m, v, g, b, training=False, eps=1e-3)
    for act in spec['activations']:
        x = ACT[act](x @ w[pos] + w[pos + 1])
        pos += 2
    return x


def loop(nodes, arcs, adjacency, arcnode, nodegraph, out_mask, *, net_state, net_output, state_vect_dim,
         max_iteration, state_threshold, focus='n', state0=None, dtype=torch.float32, timings=None):
    """Homogeneous Loop (node / graph focus) on torch CPU; returns (k, state, out) as numpy.
    `timings` (dict, optional) receives 'loop_s': wall seconds of the while-loop alone (operand conversion, the
    once-per-call aggregates and the output network excluded), the quantity comparable with the device t_loop."""
    import time
    X = torch.from_numpy(np.asarray(nodes)).to(dtype)
    lab = torch.from_numpy(np.asarray(arcs)[:, 2:]).to(dtype)
    At, ANt = _sparse_t(adjacency, dtype), _sparse_t(arcnode, dtype)
    agg_arcs = torch.sparse.mm(ANt, lab) if lab.shape[1] else torch.zeros((X.shape[0], 0), dtype=dtype)
    if state_vect_dim > 0:
        state = torch.from_numpy(np.asarray(state0)).to(dtype)
        agg_nodes = torch.sparse.mm(At, X)
        comps = lambda s: [s, X, torch.sparse.mm(At, s), agg_nodes, agg_arcs]
    else:
        state = X.clone()
        comps = lambda s: [s, torch.sparse.mm(At, s), agg_arcs]
    state_old = torch.ones_like(state)
    k = 0
    t0 = time.perf_counter()
    while True:
        dist = torch.sqrt(torch.sum(torch.square(state - state_old), dim=1))
        norm = torch.sqrt(torch.sum(torch.square(state_old), dim=1))
        if not (bool(torch.any(dist > state_threshold * norm)) and k < max_iteration):
            break
        state, state_old, k = _mlp(net_state, torch.cat(comps(state), dim=1), dtype), state, k + 1
    if timings is not None:
        timings['loop_s'] = time.perf_counter() - t0
    mask = torch.from_numpy(np.asarray(out_mask, dtype=bool))
    inp = torch.cat([state, X], dim=1)[mask] if state_vect_dim > 0 else state[mask]
    out = _mlp(net_output, inp, dtype)
    if focus == 'g':
        out = torch.sparse.mm(_sparse_t(nodegraph, dtype), out)
    return float(k), state.numpy(), out.numpy()
