"""Seeded synthetic Erdős–Rényi workloads of BASELINE.md (configs C3 / C4 / C5).

Directed G(n, M): draw (src, dst) pairs uniformly with `default_rng(seed)`, drop self-loops and duplicates, top up
until exactly M unique arcs remain. Labels are MUTAG-shaped so one network definition serves every config:
node label one-hot L uniform, arc label one-hot A uniform, node-focused, all masks true."""
from __future__ import annotations

import numpy as np

from .graph_class import GraphObject
from .composite_graph_class import CompositeGraphObject


def er_arcs(n_nodes: int, n_arcs: int, seed: int = 1234) -> np.ndarray:
    """int64 [n_arcs, 2] unique (src, dst) pairs without self-loops, sorted by (src, dst)."""
    if n_arcs > n_nodes * (n_nodes - 1):
        raise ValueError('more arcs than a simple directed graph can hold')
    rng = np.random.default_rng(seed)
    keys = np.zeros(0, dtype=np.int64)
    while len(keys) < n_arcs:
        need = n_arcs - len(keys)
        src = rng.integers(0, n_nodes, size=int(need * 1.1) + 16)
        dst = rng.integers(0, n_nodes, size=len(src))
        ok = src != dst
        keys = np.unique(np.concatenate([keys, src[ok] * n_nodes + dst[ok]]))
        if len(keys) > n_arcs:
            keys = np.sort(rng.choice(keys, size=n_arcs, replace=False))
    return np.stack([keys // n_nodes, keys % n_nodes], axis=1)


def er_graph(n_nodes: int, n_arcs: int, dim_node_label: int = 14, dim_arc_label: int = 3, dim_target: int = 2,
             focus: str = 'n', aggregation_mode: str = 'average', seed: int = 1234) -> GraphObject:
    ids = er_arcs(n_nodes, n_arcs, seed)
    rng = np.random.default_rng(seed + 1)
    nodes = np.zeros((n_nodes, dim_node_label), dtype=np.float32)
    nodes[np.arange(n_nodes), rng.integers(0, dim_node_label, n_nodes)] = 1
    arcs = np.zeros((n_arcs, 2 + dim_arc_label), dtype=np.float64)
    arcs[:, :2] = ids
    arcs[np.arange(n_arcs), 2 + rng.integers(0, dim_arc_label, n_arcs)] = 1
    n_t = {'n': n_nodes, 'a': n_arcs, 'g': 1}[focus]
    targets = np.zeros((n_t, dim_target), dtype=np.float32)
    targets[np.arange(n_t), rng.integers(0, dim_target, n_t)] = 1
    return GraphObject(nodes=nodes, arcs=arcs, targets=targets, focus=focus, aggregation_mode=aggregation_mode)


_ER_ARCS_CACHE = {}


def er_graph_slice(n_nodes: int, n_arcs: int, lo: int, hi: int, dim_node_label: int = 14, dim_arc_label: int = 3,
                   aggregation_mode: str = 'average', seed: int = 1234):
    """The `GraphSlice` (gnnkeras_amd.distributed) of `er_graph(n_nodes, n_arcs, ..., seed)` for the destination range [lo, hi):
    the same arcs, labels and aggregation weights as the whole `GraphObject` would hand to that rank, without ever building
    the whole graph's [E, 2 + A] matrix or its scipy operators.  What a rank of the sharded loop generates for itself: the id
    draw and the label draws are replayed in full (they are what fixes the graph; 80 + 80 MB at C4 size), everything derived
    from them only for the own arcs."""
    from .distributed import GraphSlice
    if aggregation_mode not in ('sum', 'normalized', 'average'): raise ValueError("ERROR: Unknown aggregation mode")
    key = (n_nodes, n_arcs, seed)
    if key not in _ER_ARCS_CACHE:
        _ER_ARCS_CACHE.clear()
        _ER_ARCS_CACHE[key] = er_arcs(n_nodes, n_arcs, seed)
    ids = _ER_ARCS_CACHE[key]
    rng = np.random.default_rng(seed + 1)
    nodes = np.zeros((n_nodes, dim_node_label), dtype=np.float32)
    nodes[np.arange(n_nodes), rng.integers(0, dim_node_label, n_nodes)] = 1
    arc_label_of = rng.integers(0, dim_arc_label, n_arcs)          # (same draw order as er_graph: node labels, arc labels)
    mine = np.flatnonzero((ids[:, 1] >= lo) & (ids[:, 1] < hi))
    src, dst = ids[mine, 0], ids[mine, 1]
    arc_labels = np.zeros((len(mine), dim_arc_label), dtype=np.float32)
    arc_labels[np.arange(len(mine)), arc_label_of[mine]] = 1
    values = np.ones(len(mine), dtype=np.float64)                   # reference graph_class.py:105-121
    if aggregation_mode == 'normalized': values *= float(1 / n_arcs)
    elif aggregation_mode == 'average':
        values /= np.bincount(dst - lo, minlength=hi - lo)[dst - lo]   # every arc of a destination lives with it
    ones = np.ones(hi - lo, dtype=bool)
    return GraphSlice(n_nodes, nodes, lo, hi, src, dst, arc_labels, values.astype(np.float32), ones, ones, arc_index=mine)


def er_composite_graph(n_nodes: int, n_arcs: int, dim_node_label=(14, 8, 4), dim_arc_label: int = 3,
                       dim_target: int = 2, focus: str = 'n', aggregation_mode: str = 'average',
                       seed: int = 1234) -> CompositeGraphObject:
    """C5: node types uniform over len(dim_node_label) types; type t uses the first dim_node_label[t] label columns."""
    ids = er_arcs(n_nodes, n_arcs, seed)
    rng = np.random.default_rng(seed + 1)
    T, Lmax = len(dim_node_label), int(max(dim_node_label))
    types = rng.integers(0, T, n_nodes)
    type_mask = np.zeros((n_nodes, T), dtype=bool)
    type_mask[np.arange(n_nodes), types] = True
    nodes = np.zeros((n_nodes, Lmax), dtype=np.float32)
    dims = np.asarray(dim_node_label)[types]
    nodes[np.arange(n_nodes), (rng.random(n_nodes) * dims).astype(int)] = 1
    arcs = np.zeros((n_arcs, 2 + dim_arc_label), dtype=np.float64)
    arcs[:, :2] = ids
    arcs[np.arange(n_arcs), 2 + rng.integers(0, dim_arc_label, n_arcs)] = 1
    n_t = {'n': n_nodes, 'a': n_arcs, 'g': 1}[focus]
    targets = np.zeros((n_t, dim_target), dtype=np.float32)
    targets[np.arange(n_t), rng.integers(0, dim_target, n_t)] = 1
    return CompositeGraphObject(nodes=nodes, arcs=arcs, targets=targets, type_mask=type_mask,
                                dim_node_label=dim_node_label, focus=focus, aggregation_mode=aggregation_mode)


def er_composite_graph_slice(n_nodes: int, n_arcs: int, lo: int, hi: int, dim_node_label=(14, 8, 4), dim_arc_label: int = 3,
                             aggregation_mode: str = 'average', seed: int = 1234):
    """The `GraphSlice` of `er_composite_graph(n_nodes, n_arcs, ..., seed)` for the destination range [lo, hi) - what ONE rank of the
    sharded composite loop (BASELINE C5) generates for itself: the draws that fix the graph are replayed in full (node types and
    labels, arc ids, arc labels), everything derived from them - aggregation weights (reference graph_class.py:105-121,
    composite_graph_class.py:73-103), the own rows of `type_mask`, the own columns of every CompositeAdjacency (reference
    composite_graph_class.py:57-70: the Adjacency entries whose SOURCE has that type) - only for the own arcs."""
    from .distributed import GraphSlice
    if aggregation_mode not in ('sum', 'normalized', 'average', 'composite_average'): raise ValueError("ERROR: Unknown aggregation mode")
    key = (n_nodes, n_arcs, seed)
    if key not in _ER_ARCS_CACHE:
        _ER_ARCS_CACHE.clear()
        _ER_ARCS_CACHE[key] = er_arcs(n_nodes, n_arcs, seed)
    ids = _ER_ARCS_CACHE[key]
    rng = np.random.default_rng(seed + 1)                        # (the draw order of er_composite_graph: types, labels, arc labels)
    T, Lmax = len(dim_node_label), int(max(dim_node_label))
    types = rng.integers(0, T, n_nodes)
    nodes = np.zeros((n_nodes, Lmax), dtype=np.float32)
    dims = np.asarray(dim_node_label)[types]
    nodes[np.arange(n_nodes), (rng.random(n_nodes) * dims).astype(int)] = 1
    arc_label_of = rng.integers(0, dim_arc_label, n_arcs)
    mine = np.flatnonzero((ids[:, 1] >= lo) & (ids[:, 1] < hi))
    src, dst = ids[mine, 0], ids[mine, 1]
    arc_labels = np.zeros((len(mine), dim_arc_label), dtype=np.float32)
    arc_labels[np.arange(len(mine)), arc_label_of[mine]] = 1
    values = np.ones(len(mine), dtype=np.float64)
    if aggregation_mode == 'normalized': values *= float(1 / n_arcs)
    elif aggregation_mode == 'average':
        values /= np.bincount(dst - lo, minlength=hi - lo)[dst - lo]           # every arc of a destination lives with it
    elif aggregation_mode == 'composite_average':                              # 1 / #(in-neighbours of dst with the type of src)
        src_type = types[src]
        for t in range(T):
            sel = src_type == t
            if np.any(sel): values[sel] /= np.bincount(dst[sel] - lo, minlength=hi - lo)[dst[sel] - lo]
    values = values.astype(np.float32)
    type_mask = np.zeros((hi - lo, T), dtype=bool)
    type_mask[np.arange(hi - lo), types[lo:hi]] = True
    src_type = types[src]
    cas = [(src[src_type == t], dst[src_type == t], values[src_type == t]) for t in range(T)]
    ones = np.ones(hi - lo, dtype=bool)
    comp = dict(type_mask=type_mask, dim_node_label=[int(v) for v in dim_node_label], adjacencies=cas)
    return GraphSlice(n_nodes, nodes, lo, hi, src, dst, arc_labels, values, ones, ones, arc_index=mine, composite=comp)


def er_device_batch(n_nodes: int, n_arcs: int, device, dim_node_label: int = 14, dim_arc_label: int = 3,
                    aggregation_mode: str = 'average', seed: int = 1234):
    """The 8-element `x` list of `MultiGraphSequencer.__getitem__` for ONE directed G(n, M) graph, assembled on the
    device with torch (sort / unique on the GPU) instead of through `GraphObject` on the host: the C4-times-4 point of
    bench.py (4 M nodes / 40 M arcs) would otherwise spend minutes in numpy before the first kernel.  Same conventions as
    the host path: arcs unique, no self-loops, sorted by (src, dst) (`graph_class.py:47`); Adjacency / ArcNode entries of a
    destination in ascending source order; 'sum' or 'average' weights as one scale per destination row.
    The sparse operands are `SparseMatrix` objects that carry only their device CSR (what the kernels walk)."""
    import torch
    from .sparse import SparseMatrix
    if aggregation_mode not in ('sum', 'average'): raise ValueError("er_device_batch builds 'sum' or 'average' operators")
    dev = torch.device(device)
    gen = torch.Generator(device=dev); gen.manual_seed(seed)
    keys = torch.zeros(0, dtype=torch.int64, device=dev)
    while keys.numel() < n_arcs:
        need = n_arcs - keys.numel()
        m = int(need * 1.1) + 16
        src = torch.randint(0, n_nodes, (m,), generator=gen, device=dev)
        dst = torch.randint(0, n_nodes, (m,), generator=gen, device=dev)
        ok = src != dst
        keys = torch.unique(torch.cat([keys, src[ok] * n_nodes + dst[ok]]))          # sorted by (src, dst)
        if keys.numel() > n_arcs:
            keep = torch.randperm(keys.numel(), generator=gen, device=dev)[:n_arcs]
            keys = keys[torch.sort(keep).values]
    src, dst = keys // n_nodes, keys % n_nodes
    del keys
    order = torch.sort(dst, stable=True).indices                                    # arcs grouped by destination, ascending source inside
    counts = torch.bincount(dst, minlength=n_nodes)
    rowptr = torch.zeros(n_nodes + 1, dtype=torch.int64, device=dev)
    rowptr[1:] = torch.cumsum(counts, 0)
    rowptr = rowptr.to(torch.int32)
    adj_src = src[order].to(torch.int32)
    arc_src = order.to(torch.int32)                                                 # ArcNode: rows are arc ids
    row_scale = None
    if aggregation_mode == 'average':
        row_scale = torch.where(counts > 0, 1.0 / counts.clamp(min=1).to(torch.float32), torch.ones((), device=dev))
    max_degree = int(counts.max())

    def operand(src_ids, n_src):
        return SparseMatrix.device_only((n_src, n_nodes), dict(rowptr=rowptr, src=src_ids, w=None, row_scale=row_scale, n_src=int(n_src),
                                                                n_dst=int(n_nodes), nnz=int(n_arcs), max_degree=max_degree), dev)

    if max_degree > 512: raise ValueError('er_device_batch does not split hub rows; use the GraphObject path')
    nodes = torch.zeros((n_nodes, dim_node_label), dtype=torch.float32, device=dev)
    nodes[torch.arange(n_nodes, device=dev), torch.randint(0, dim_node_label, (n_nodes,), generator=gen, device=dev)] = 1
    arcs = torch.zeros((n_arcs, 2 + dim_arc_label), dtype=torch.float32, device=dev)
    arcs[:, 0], arcs[:, 1] = src.to(torch.float32), dst.to(torch.float32)            # float ids like the reference (Q4); never read as ids here
    arcs[torch.arange(n_arcs, device=dev), 2 + torch.randint(0, dim_arc_label, (n_arcs,), generator=gen, device=dev)] = 1
    ones = torch.ones(n_nodes, dtype=torch.bool, device=dev)
    nodegraph = SparseMatrix(np.zeros((0, 2), np.int64), np.zeros(0, np.float32), (n_nodes, 1))
    return [nodes, arcs, torch.tensor([[dim_node_label]], dtype=torch.int32), ones, ones.clone(),
            operand(adj_src, n_nodes), operand(arc_src, n_arcs), nodegraph]
