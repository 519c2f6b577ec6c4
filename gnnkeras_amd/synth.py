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
