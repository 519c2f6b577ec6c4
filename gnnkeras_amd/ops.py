"""`torch.ops.gnnkeras.*` — the PyTorch-ROCm custom ops the host code calls the HIP kernels through (BASELINE.json north_star;
SURVEY.md §8b). The ops are registered in C++ (`csrc/torch_ops.cpp` -> `csrc/libgnnkeras_torch.so`, TORCH_LIBRARY over the C
ABI of libgnnloop.so); this module loads that library and marshals the host objects (`SparseMatrix`, `Sequential`) into the
op schemas. There is no CPU implementation: the ops are registered for HIP devices only and the dispatcher rejects CPU tensors.

    k, state, out = ops.loop_forward(...)        torch.ops.gnnkeras.loop_forward   Loop        (GNN.py:245-274)
    y   = ops.aggregate(csr_dict, X)            torch.ops.gnnkeras.aggregate      A^T . X      (GNN.py:228, :254, :258)
    y   = ops.pool(csr_dict, out_nodes)         torch.ops.gnnkeras.pool           NodeGraph^T . out (GNN.py:345)
    f   = ops.converged(state, state_old, thr)  torch.ops.gnnkeras.converged      condition    (GNN.py:196-212)
    s,f = ops.state_step(...)                   torch.ops.gnnkeras.state_step     convergence  (GNN.py:217-236)
    y   = ops.mlp_forward(net, X)               torch.ops.gnnkeras.mlp_forward    Sequential inference call
"""
from __future__ import annotations

import os

import torch

from . import _native as nat

OPS_LIB_PATH = os.path.join(nat.CSRC, 'libgnnkeras_torch.so')
BN_EPSILON = 1e-3
_loaded = False


def load():
    """Load libgnnkeras_torch.so (registers the `gnnkeras` op namespace); raises if it has not been built."""
    global _loaded
    if not _loaded:
        nat.lib()                                               # libgnnloop.so first: the op library links against it
        if not os.path.exists(OPS_LIB_PATH):
            raise nat.NativeError(f'{OPS_LIB_PATH} is missing: run `python -c "import __graft_entry__ as g; g.build()"` or '
                                  f'`make -C {nat.CSRC}`. There is no fallback for the custom ops.')
        torch.ops.load_library(OPS_LIB_PATH)
        _loaded = True
    return torch.ops.gnnkeras


def csr_args(d):
    """(`Tensor?[4]`, `int[3]`) of a `SparseMatrix.device_csr()` dict; (`[]`, `[]`) for an absent operator."""
    if d is None: return [], []
    return [d['rowptr'], d['src'], d['w'], d['row_scale']], [d['n_dst'], d['n_src'], d['nnz']]


def hub_args(adj):
    """Hub rows of an adjacency's device CSR (sparse.split_heavy): the light operator + segment lists, or ([], [])."""
    if adj.get('heavy') is None: return [], []
    l, h = adj['light'], adj['heavy']
    return [l['rowptr'], l['src'], l['w'], l['row_scale'], h['seg_beg'], h['seg_end']], [l['n_dst'], l['n_src'], l['nnz'], h['n_seg']]


def net_args(net, device, shape_only=False):
    """(`Tensor[]` weights in get_weights() order, `int[]` spec) of a `Sequential`."""
    if not shape_only: net.to(device)
    n = len(net.units)
    return (None if shape_only else net.weights), [net.input_dim, int(net.batch_normalization), n] + list(net.units) + [nat.ACTIVATIONS[a] for a in net.activations]


def loop_forward(nodes, arcs, adjacency, arcnode, nodegraph, net_state, net_output, state0, out_index, arc_ends, state_dim,
                 max_iteration, state_threshold, focus, flags, composite=None, loop_events=None, groups=None, group_sets=None):
    """`adjacency` / `arcnode` / `nodegraph`: device-CSR dicts (nodegraph None unless graph focus); `net_state`: one
    `Sequential`, or the list of per-type networks with `composite` = (type_nodes i32[N], type_offsets [T+1], type_dim_label
    [T], [device-CSR dict per type]); `arc_ends` = (arc_src, arc_dst) for arc focus; `loop_events` = (begin, end)
    `torch.cuda.Event`s recorded on the launch stream around the iteration launches; `groups` = node offsets [G + 1] of
    merged batches that run as independent loops of this one call (k is then [G]; ask `loop_groups_supported` first);
    `group_sets` = first-group offsets [B + 1] when a batch was cut into several groups that share the loop's condition."""
    ops = load()
    dev = nodes.device
    adj_t, adj_d = csr_args(adjacency)
    an_t, an_d = csr_args(arcnode)
    ng_t, ng_d = csr_args(nodegraph)
    hub_t, hub_d = hub_args(adjacency)
    nets = list(net_state) if isinstance(net_state, (list, tuple)) else [net_state]
    sw, ss = [], []
    for n_ in nets:
        w, s = net_args(n_, dev)
        sw += w; ss += s
    ow, os_ = net_args(net_output, dev)
    if composite is not None:
        type_nodes, type_offsets, type_dims, cas = composite
        ca_t, ca_d = [], []
        for c in cas:
            t, d = csr_args(c)
            ca_t += t; ca_d += d
        type_offsets, type_dims = [int(v) for v in type_offsets], [int(v) for v in type_dims]
    else:
        type_nodes, type_offsets, type_dims, ca_t, ca_d = None, [], [], [], []
    ev = [e.cuda_event for e in loop_events] if loop_events is not None else []
    es, ed = arc_ends if arc_ends is not None else (None, None)
    return ops.loop_forward(nodes, arcs, adj_t, adj_d, an_t, an_d, ng_t, ng_d, sw, ss, ow, os_, BN_EPSILON, state0, out_index, es, ed,
                            int(state_dim), int(max_iteration), float(state_threshold), int(focus), int(flags), hub_t, hub_d,
                            type_nodes, type_offsets, type_dims, ca_t, ca_d, ev, [int(v) for v in groups] if groups is not None else [],
                            [int(v) for v in group_sets] if group_sets is not None else [])


def loop_groups_supported(n_nodes, dim_node_label, dim_arc_label, net_state, net_output, state_dim, max_iteration, focus, flags,
                          n_out, groups, group_sets=None):
    """May these merged batches run as independent loops of ONE call?  0 no, 1 spread over the CUs (<= 32 groups), 2 one CU
    per group with its state in LDS (any number of groups).  (shapes only: include/gnnloop.h, gnn_loop_groups_supported)"""
    _, ss = net_args(net_state, None, shape_only=True)
    _, os_ = net_args(net_output, None, shape_only=True)
    return int(load().loop_groups_supported(int(n_nodes), int(dim_node_label), int(dim_arc_label), ss, os_, int(state_dim),
                                             int(max_iteration), int(focus), int(flags), int(n_out), [int(v) for v in groups],
                                             [int(v) for v in group_sets] if group_sets is not None else []))


def aggregate(csr, X):
    t, d = csr_args(csr)
    return load().aggregate(t, d, X)


def pool(nodegraph_csr, out_nodes):
    t, d = csr_args(nodegraph_csr)
    return load().pool(t, d, out_nodes)


def converged(state, state_old, threshold):
    return load().converged(state, state_old, float(threshold))


def state_step(nodes, arcs, adjacency, arcnode, net_state, state, state_dim, state_threshold, flags=0, composite=None, aggregated=None):
    """`net_state`: one `Sequential`, or the per-type list with `composite` = (type_nodes, type_offsets, type_dim_label, [device-CSR
    dict per type]) as for `loop_forward` (CompositeGNN.py:215-234).  `aggregated` = (aggregated_nodes, aggregated_arcs): the
    iteration constants the reference's `convergence` is handed (GNN.py:217; the column blocks of `aggregated_component` for the
    composite form) - `arcs` may then be an empty [0, 2 + A] matrix, `arcnode` None and the composite adjacencies absent."""
    adj_t, adj_d = csr_args(adjacency)
    an_t, an_d = csr_args(arcnode)
    hub_t, hub_d = hub_args(adjacency)
    nets = list(net_state) if isinstance(net_state, (list, tuple)) else [net_state]
    w, s = [], []
    for n_ in nets:
        w_, s_ = net_args(n_, nodes.device)
        w += w_; s += s_
    agg_n, agg_a = aggregated if aggregated is not None else (None, None)
    if composite is None:
        return load().state_step(nodes, arcs, adj_t, adj_d, an_t, an_d, w, s, BN_EPSILON, state, int(state_dim), float(state_threshold),
                                 int(flags), hub_t, hub_d, None, [], [], [], [], agg_n, agg_a)
    type_nodes, type_offsets, type_dims, cas = composite
    ca_t, ca_d = [], []
    for c in (cas or []):
        t, d = csr_args(c)
        ca_t += t; ca_d += d
    return load().state_step(nodes, arcs, adj_t, adj_d, an_t, an_d, w, s, BN_EPSILON, state, int(state_dim), float(state_threshold),
                             int(flags), hub_t, hub_d, type_nodes, [int(v) for v in type_offsets], [int(v) for v in type_dims], ca_t, ca_d,
                             agg_n, agg_a)


def mlp_forward(net, X):
    w, s = net_args(net, X.device)
    return load().mlp_forward(w, s, BN_EPSILON, X)
