"""ORACLE HARNESS — TEST INFRASTRUCTURE ONLY (see gnn_oracle.py header; parity of the Loop is UNPINNED by the
reference: no TensorFlow here and no reference tests).

Glue that feeds the NumPy restatement with exactly what the product's sequencer hands to the product's model, so that
tests / smoke / bench compare the two on identical inputs:

    x_list (torch tensors + sparse triples, as emitted by MultiGraphSequencer.__getitem__)  ->  numpy  ->  oracle.loop
"""
from __future__ import annotations

import numpy as np

from . import gnn_oracle as O


def _np(x):
    try:
        import torch
        if isinstance(x, torch.Tensor):
            return x.detach().cpu().numpy()
    except ImportError:
        pass
    return np.asarray(x)


def _triple(t):
    idx, val, shp = t
    return _np(idx).reshape(-1, 2), _np(val).reshape(-1), _np(shp).reshape(-1)


def net_of(seq_model):
    """(spec, weights) of a product `Sequential` (weights copied to host)."""
    return seq_model.spec()


def oracle_loop(model, x_list, state0=None, dtype=np.float32, exact_order=True, training=False):
    """Run the oracle on one homogeneous sequencer item with the product model's weights and hyper-parameters."""
    nodes, arcs, dim_node_label, set_mask, output_mask, adjacency, arcnode, nodegraph = x_list
    return O.loop(_np(nodes), _np(arcs), _np(dim_node_label).reshape(-1), _np(set_mask).reshape(-1),
                  _np(output_mask).reshape(-1), _triple(adjacency), _triple(arcnode), _triple(nodegraph),
                  net_state=net_of(model.net_state), net_output=net_of(model.net_output),
                  state_vect_dim=model.state_vect_dim, max_iteration=model.max_iteration,
                  state_threshold=model.state_threshold, focus=model._focus, training=training,
                  state0=None if state0 is None else _np(state0), dtype=dtype, exact_order=exact_order)


def oracle_composite_loop(model, x_list, state0=None, dtype=np.float32, exact_order=True, training=False):
    """Same for a composite sequencer item (10-element x_list, GraphSequencers.py:239-244)."""
    (nodes, arcs, dim_node_label, type_mask, set_mask, output_mask, composite_adjacencies, adjacency, arcnode,
     nodegraph) = x_list
    tm = _np(type_mask)
    tm = tm.reshape(tm.shape[0], -1)
    return O.composite_loop(_np(nodes), _np(arcs), _np(dim_node_label).reshape(-1), tm, _np(set_mask).reshape(-1),
                            _np(output_mask).reshape(-1), [_triple(c) for c in composite_adjacencies],
                            _triple(adjacency), _triple(arcnode), _triple(nodegraph),
                            net_state=[net_of(n) for n in model.net_state], net_output=net_of(model.net_output),
                            state_vect_dim=model.state_vect_dim, max_iteration=model.max_iteration,
                            state_threshold=model.state_threshold, focus=model._focus, training=training,
                            state0=None if state0 is None else _np(state0), dtype=dtype, exact_order=exact_order)


def rel_err(a, b):
    """max|a-b| / max(|b|, tiny): the parity measure of BASELINE.md (C2)."""
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    if a.size == 0 and b.size == 0:
        return 0.0
    return float(np.max(np.abs(a - b)) / max(float(np.max(np.abs(b))), 1e-30))
