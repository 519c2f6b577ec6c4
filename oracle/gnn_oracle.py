"""ORACLE — TEST INFRASTRUCTURE ONLY.  Not product code; never imported by `gnnkeras_amd`.

Only `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py` may import this
module, and only as the *checker* (or as the CPU baseline being timed), never as the thing shipped.

What it is: a literal, op-for-op NumPy restatement of GNNkeras' convergent message-passing loop
(reference `GNN/Models/GNN.py:181-274, 317-346` and `GNN/Models/CompositeGNN.py:178-272, 315-343`),
executing the reference's *un-fused* op sequence (adjoint SpMM, concat, BatchNormalization affine,
matmul + bias, activation, the 12-op convergence predicate, one Python-level bool per iteration).

PARITY UNPINNED for the Loop itself: the arithmetic of the reference lives in TensorFlow/Keras
(`requirements.txt:2`, version unpinned), which is not installed in the build container, is absent from
`/root/reference`, and the reference ships no tests or golden vectors for this path.
The TF/Keras op semantics restated here (from the published TF API semantics) are:
  * `tf.sparse.sparse_dense_matmul(A, B, adjoint_a=True)` = AᵀB, accumulated in nnz order of the
    row-major-reordered SparseTensor (`graph_class.py:558`);
  * Keras `Dense`: `act(x @ W + b)`; Keras `BatchNormalization` (eps 1e-3): inference
    `x * (γ/√(σ²+ε)) + (β − μ·γ/√(σ²+ε))`, training = same with biased batch statistics over axis 0;
  * `selu` scale 1.0507009873554805, alpha 1.6732632423543772; `softmax` with max-subtraction;
  * eager `tf.while_loop` = `while cond(*vars): vars = body(*vars)`;
  * `tf.boolean_mask` keeps order; `tf.scatter_nd` writes into zeros; `tf.gather(x, idx[nnz,2])` -> [nnz,2,F].
What *is* pinned: the graph operands fed to it (ArcNode / Adjacency / NodeGraph / CompositeAdjacencies),
by golden fixtures generated from the reference's own numpy/scipy code (`tests/golden/make_golden.py`);
hand-computable known-answer cases in `tests/test_oracle.py`; and an independent torch-CPU
re-implementation cross-check of the Keras op semantics (`tests/test_oracle.py`).

All functions take an explicit `dtype` (np.float32 = the reference's floatx, np.float64 = arbiter).
"""
from __future__ import annotations

import numpy as np

SELU_SCALE = 1.0507009873554805
SELU_ALPHA = 1.6732632423543772
BN_EPS = 1e-3


# ----------------------------------------------------------------------------------------------------------------------
# TF op restatements
# ----------------------------------------------------------------------------------------------------------------------
def sparse_reorder(indices, values):
    """`tf.sparse.reorder`: canonical row-major ordering (graph_class.py:558)."""
    indices = np.asarray(indices, dtype=np.int64).reshape(-1, 2)
    values = np.asarray(values).reshape(-1)
    order = np.lexsort((indices[:, 1], indices[:, 0]))
    return indices[order], values[order]


def sparse_dense_matmul_adjoint(indices, values, dense_shape, B, dtype, exact_order=True):
    """`tf.sparse.sparse_dense_matmul(A, B, adjoint_a=True)` -> AᵀB  (GNN.py:228, :254, :258, :345).

    `exact_order=True`: out[col] += val * B[row], nnz by nnz in storage order (unbuffered np.add.at),
    which is the accumulation order of TF's CPU kernel. `exact_order=False` uses scipy CSR (fast path used
    only for timing the CPU baseline on large graphs)."""
    indices = np.asarray(indices, dtype=np.int64).reshape(-1, 2)
    values = np.asarray(values, dtype=dtype).reshape(-1)
    B = np.asarray(B, dtype=dtype)
    n_rows, n_cols = int(dense_shape[0]), int(dense_shape[1])
    assert B.shape[0] == n_rows, (B.shape, dense_shape)
    if not exact_order:
        return np.asarray(_scipy_adjoint(indices, values, n_rows, n_cols, dtype) @ B, dtype=dtype)
    out = np.zeros((n_cols, B.shape[1]), dtype=dtype)
    if len(values):
        np.add.at(out, indices[:, 1], values[:, None] * B[indices[:, 0]])
    return out


import threading
_SCIPY_TLS = threading.local()    # per thread: [(key, indices, values, At)] - the loop hands the SAME adjacency buffers to every iteration


def _scipy_adjoint(indices, values, n_rows, n_cols, dtype):
    """CSR of Aᵀ for the `exact_order=False` path, built once per (indices, values) buffer pair instead of once per iteration
    (the COO -> CSR conversion of 10 M entries costs more than the product itself).  Keyed on the buffers' addresses; the
    arrays are kept referenced so that an address cannot be reused while its entry lives."""
    from scipy.sparse import csr_matrix
    key = (indices.__array_interface__['data'][0], values.__array_interface__['data'][0], len(values), np.dtype(dtype).str, n_rows, n_cols)
    cache = _SCIPY_TLS.__dict__.setdefault('cache', [])
    for k_, _i, _v, At in cache:
        if k_ == key: return At
    At = csr_matrix((values, (indices[:, 1], indices[:, 0])), shape=(n_cols, n_rows), dtype=dtype)
    cache.append((key, indices, values, At))
    if len(cache) > 4: cache.pop(0)
    return At


def activation(name, x, dtype):
    """Keras activation by name (MLP.py:16 passes the strings through to `Dense`)."""
    name = 'linear' if name is None else str(name).lower()
    if name == 'linear':
        return x
    if name == 'relu':
        return np.maximum(x, dtype(0))
    if name == 'selu':
        neg = dtype(SELU_ALPHA) * np.expm1(np.minimum(x, dtype(0)))
        return (dtype(SELU_SCALE) * np.where(x > 0, x, neg)).astype(dtype)
    if name == 'elu':
        return np.where(x > 0, x, np.expm1(np.minimum(x, dtype(0)))).astype(dtype)
    if name == 'tanh':
        return np.tanh(x).astype(dtype)
    if name == 'sigmoid':
        return (dtype(1) / (dtype(1) + np.exp(-x))).astype(dtype)
    if name == 'softplus':
        return np.logaddexp(x, dtype(0)).astype(dtype)
    if name == 'softmax':
        z = x - np.max(x, axis=-1, keepdims=True)
        e = np.exp(z)
        return (e / np.sum(e, axis=-1, keepdims=True)).astype(dtype)
    raise ValueError(f'unknown activation {name!r}')


def mlp_apply(spec, weights, x, training, dtype):
    """Keras `Sequential` built by the reference `MLP()` (MLP.py:12-78): [BatchNormalization] + Dense x n.

    spec    : {'batch_normalization': bool, 'activations': [str]*n}   (Dropout layers are the identity at inference
              and are not modelled in training mode either: the oracle is deterministic)
    weights : flat list in Keras `get_weights()` order: BN -> [gamma, beta, moving_mean, moving_variance],
              each Dense -> [kernel(in,out), bias(out)]."""
    w = [np.asarray(a, dtype=dtype) for a in weights]
    x = np.asarray(x, dtype=dtype)
    pos = 0
    if spec.get('batch_normalization', False):
        gamma, beta, mean, var = w[0:4]
        pos = 4
        if training and x.shape[0] > 0:
            mean = np.mean(x, axis=0, dtype=dtype)
            var = np.mean(np.square(x - mean), axis=0, dtype=dtype)  # biased batch variance
        inv = gamma / np.sqrt(var + dtype(BN_EPS))
        x = (x * inv + (beta - mean * inv)).astype(dtype)
    for act in spec['activations']:
        kernel, bias = w[pos], w[pos + 1]
        pos += 2
        x = activation(act, (x @ kernel + bias).astype(dtype), dtype)
    assert pos == len(w), 'weights list longer than the layer spec'
    return x


# ----------------------------------------------------------------------------------------------------------------------
# homogeneous GNN  (GNN/Models/GNN.py)
# ----------------------------------------------------------------------------------------------------------------------
def condition(k, state, state_old, max_iteration, state_threshold, dtype):
    """GNN.py:196-214 — strict `>`; reduce_any over all nodes of the merged batch; `k < max_iteration`."""
    outDistance = np.sqrt(np.sum(np.square(state - state_old), axis=1, dtype=dtype))
    state_norm = np.sqrt(np.sum(np.square(state_old), axis=1, dtype=dtype))
    scaled_state_norm = dtype(state_threshold) * state_norm
    checkDistanceVec = outDistance > scaled_state_norm
    c1 = bool(np.any(checkDistanceVec))
    c2 = bool(k < max_iteration)
    return c1 and c2


def convergence(state, nodes, adjacency, aggregated_nodes, aggregated_arcs, net_state, state_vect_dim, training,
                dtype, exact_order=True):
    """GNN.py:217-236 — one state-transition step; returns state_new."""
    node_components = [state]
    if state_vect_dim > 0:
        node_components += [nodes]
    aggregated_states = sparse_dense_matmul_adjoint(*adjacency, state, dtype, exact_order)
    inp_state = np.concatenate(node_components + [aggregated_states, aggregated_nodes, aggregated_arcs], axis=1)
    return mlp_apply(net_state[0], net_state[1], inp_state, training, dtype)


def apply_filters_node(state_converged, nodes, mask, state_vect_dim):
    """GNN.py:239-242."""
    if state_vect_dim:
        state_converged = np.concatenate([state_converged, nodes], axis=1)
    return state_converged[np.asarray(mask, dtype=bool)]


def apply_filters_arc(state_converged, nodes, adjacency_indices, arcs_label, mask, state_vect_dim):
    """GNN.py:317-330 — `tf.gather(state, adjacency.indices)` then reshape to (E, 2F)."""
    if state_vect_dim:
        state_converged = np.concatenate([state_converged, nodes], axis=1)
    idx = np.asarray(adjacency_indices, dtype=np.int64).reshape(-1, 2)
    states = state_converged[idx]                                   # (E, 2, F)
    states = states.reshape(arcs_label.shape[0], 2 * state_converged.shape[1])
    arc_state = np.concatenate([states, arcs_label], axis=1)
    return arc_state[np.asarray(mask, dtype=bool)]


def loop(nodes, arcs, dim_node_label, set_mask, output_mask, adjacency, arcnode, nodegraph, *,
         net_state, net_output, state_vect_dim, max_iteration, state_threshold, focus='n', training=False,
         state0=None, dtype=np.float32, exact_order=True, return_trace=False):
    """GNNnodeBased.Loop / GNNarcBased (apply_filters) / GNNgraphBased.Loop — GNN.py:245-274, :317-330, :341-346.

    adjacency / arcnode / nodegraph : (indices[nnz,2] int, values[nnz], dense_shape[2]) triples as emitted by
        the sequencer (GraphSequencers.py:108-110), already in `tf.sparse.reorder` order.
    net_state / net_output : (spec, weights) pairs, see `mlp_apply`.
    state0 : explicit initial state replacing `tf.random.normal(stddev=0.1)` (GNN.py:257).
    Returns (k as dtype scalar [Q6: k is a float tensor], state, out)."""
    nodes = np.asarray(nodes, dtype=dtype)
    arcs = np.asarray(arcs, dtype=dtype)
    arcs_label = arcs[:, 2:]
    adjacency = (np.asarray(adjacency[0]).reshape(-1, 2), np.asarray(adjacency[1], dtype=dtype).reshape(-1),
                 np.asarray(adjacency[2]).reshape(-1))

    # GNN.py:254-259
    aggregated_arcs = sparse_dense_matmul_adjoint(arcnode[0], arcnode[1], np.asarray(arcnode[2]).reshape(-1),
                                                  arcs_label, dtype, exact_order)
    aggregated_nodes = np.zeros((nodes.shape[0], 0), dtype=dtype)
    if state_vect_dim > 0:
        assert state0 is not None, 'state_vect_dim>0 needs an explicit state0 (reference draws it at random)'
        state = np.asarray(state0, dtype=dtype)
        assert state.shape == (nodes.shape[0], state_vect_dim)
        aggregated_nodes = np.concatenate(
            [aggregated_nodes, sparse_dense_matmul_adjoint(*adjacency, nodes, dtype, exact_order)], axis=1)
    else:
        state = nodes.copy()
    k = 0
    state_old = np.ones_like(state, dtype=dtype)
    trace = [state]

    # GNN.py:265 — eager tf.while_loop
    while condition(k, state, state_old, max_iteration, state_threshold, dtype):
        state_new = convergence(state, nodes, adjacency, aggregated_nodes, aggregated_arcs, net_state,
                                state_vect_dim, training, dtype, exact_order)
        k, state, state_old = k + 1, state_new, state
        if return_trace:
            trace.append(state)

    # GNN.py:269-273
    mask = np.logical_and(np.asarray(set_mask, dtype=bool).reshape(-1), np.asarray(output_mask, dtype=bool).reshape(-1))
    if focus == 'a':
        inp_out = apply_filters_arc(state, nodes, adjacency[0], arcs_label, mask, state_vect_dim)
    else:
        inp_out = apply_filters_node(state, nodes, mask, state_vect_dim)
    out = mlp_apply(net_output[0], net_output[1], inp_out, training, dtype)

    # GNN.py:341-346
    if focus == 'g':
        out = sparse_dense_matmul_adjoint(nodegraph[0], nodegraph[1], np.asarray(nodegraph[2]).reshape(-1), out,
                                          dtype, exact_order)
    if return_trace:
        return dtype(k), state, out, trace
    return dtype(k), state, out


# ----------------------------------------------------------------------------------------------------------------------
# composite (heterogeneous) GNN  (GNN/Models/CompositeGNN.py)
# ----------------------------------------------------------------------------------------------------------------------
def composite_convergence(state, nodes, dim_node_label, type_mask, adjacency, aggregated_component, net_state,
                          training, dtype, exact_order=True):
    """CompositeGNN.py:215-234 — per-type net_state on boolean-masked rows, scatter_nd back, sum over types."""
    aggregated_states = sparse_dense_matmul_adjoint(*adjacency, state, dtype, exact_order)
    state_new = []
    for d, m, net in zip(dim_node_label, type_mask, net_state):
        inp_state_i = np.concatenate([nodes[:, :int(d)], state, aggregated_states, aggregated_component], axis=1)
        inp_state_i = inp_state_i[np.asarray(m, dtype=bool)]
        state_new.append(mlp_apply(net[0], net[1], inp_state_i, training, dtype))
    scattered = []
    for m, s in zip(type_mask, state_new):
        full = np.zeros((len(m), s.shape[1]), dtype=dtype)
        full[np.flatnonzero(m)] = s
        scattered.append(full)
    return np.sum(np.stack(scattered, axis=0), axis=0, dtype=dtype)


def composite_loop(nodes, arcs, dim_node_label, type_mask, set_mask, output_mask, composite_adjacencies, adjacency,
                   arcnode, nodegraph, *, net_state, net_output, state_vect_dim, max_iteration, state_threshold,
                   focus='n', training=False, state0=None, dtype=np.float32, exact_order=True):
    """CompositeGNNnodeBased.Loop / arc / graph — CompositeGNN.py:242-272, :315-327, :338-343.

    type_mask : (T, N) bool  (transposed layout of CompositeGraphTensor, composite_graph_class.py:263)
    net_state : list of T (spec, weights) pairs."""
    nodes = np.asarray(nodes, dtype=dtype)
    arcs = np.asarray(arcs, dtype=dtype)
    arcs_label = arcs[:, 2:]
    dim_node_label = [int(d) for d in np.asarray(dim_node_label).reshape(-1)]
    type_mask = np.asarray(type_mask, dtype=bool)
    adjacency = (np.asarray(adjacency[0]).reshape(-1, 2), np.asarray(adjacency[1], dtype=dtype).reshape(-1),
                 np.asarray(adjacency[2]).reshape(-1))

    # CompositeGNN.py:251-253
    aggregated_nodes = [sparse_dense_matmul_adjoint(a[0], a[1], np.asarray(a[2]).reshape(-1), nodes[:, :d], dtype,
                                                    exact_order)
                        for a, d in zip(composite_adjacencies, dim_node_label)]
    aggregated_arcs = sparse_dense_matmul_adjoint(arcnode[0], arcnode[1], np.asarray(arcnode[2]).reshape(-1),
                                                  arcs_label, dtype, exact_order)
    aggregated_component = np.concatenate(aggregated_nodes + [aggregated_arcs], axis=1)

    k = 0
    if state_vect_dim > 0:
        assert state0 is not None
        state = np.asarray(state0, dtype=dtype)
    else:
        state = nodes.copy()
    state_old = np.ones_like(state, dtype=dtype)

    while condition(k, state, state_old, max_iteration, state_threshold, dtype):
        state_new = composite_convergence(state, nodes, dim_node_label, type_mask, adjacency, aggregated_component,
                                          net_state, training, dtype, exact_order)
        k, state, state_old = k + 1, state_new, state

    mask = np.logical_and(np.asarray(set_mask, dtype=bool).reshape(-1), np.asarray(output_mask, dtype=bool).reshape(-1))
    if focus == 'a':
        # CompositeGNN.py:315-327 — state only, no label concat
        idx = adjacency[0]
        states = state[idx].reshape(arcs_label.shape[0], 2 * state.shape[1])
        inp_out = np.concatenate([states, arcs_label], axis=1)[mask]
    else:
        inp_out = state[mask]                                       # CompositeGNN.py:237-239
    out = mlp_apply(net_output[0], net_output[1], inp_out, training, dtype)
    if focus == 'g':
        out = sparse_dense_matmul_adjoint(nodegraph[0], nodegraph[1], np.asarray(nodegraph[2]).reshape(-1), out,
                                          dtype, exact_order)
    return dtype(k), state, out
