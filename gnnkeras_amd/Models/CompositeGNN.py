"""Composite (heterogeneous) recurrent GNN on MI355X: one state network per node type.

Host-side mirror of the reference's `GNN/Models/CompositeGNN.py`. The reference applies each `net_state[t]` to the
`tf.boolean_mask`ed rows of type t and `tf.scatter_nd`s them back, summing T full-size tensors per iteration
(`CompositeGNN.py:223-232`). Here nodes are grouped by type once (an index list per type); each iteration is one fused
launch per type that gathers, multiplies with that type's weights on the matrix cores and writes its rows in place.
"""
from __future__ import annotations

import json

import numpy as np
import torch

from .. import _native as nat
from .. import ops
from ..sparse import SparseMatrix
from .GNN import GNNnodeBased, _LoopModel, _squeeze_last, _arc_endpoints
from .MLP import Sequential


class CompositeGNNnodeBased(GNNnodeBased):
    """Composite GNN for node-focused problems (reference CompositeGNN.py:8-304)."""
    name = "node"
    _focus = 'n'

    def __init__(self, net_state: list, net_output: Sequential, state_vect_dim: int, max_iteration: int,
                 state_threshold: float) -> None:
        assert state_vect_dim >= 0
        assert max_iteration > 0
        assert state_threshold >= 0
        self.net_state = list(net_state)
        self.net_output = net_output
        self.state_vect_dim = int(state_vect_dim)
        self.max_iteration = int(max_iteration)
        self.state_threshold = state_threshold
        self.native_flags = 0
        self.loop_events = None
        self._engine_init()
        self._type_cache = {}

    def copy(self, copy_weights: bool = True):
        config = self.get_config()
        config["net_state"] = [i.clone(copy_weights) for i in config["net_state"]]
        config["net_output"] = config["net_output"].clone(copy_weights)
        return self.from_config(config)

    def __repr__(self):
        return f"CompositeGNN(type={self.name}, state_dim={self.state_vect_dim}, " \
               f"threshold={self.state_threshold}, max_iter={self.max_iteration}), avg={self.average_st_grads}"

    __str__ = __repr__

    def save(self, path: str, *args, **kwargs):
        """`<path>/net_state_{i}/`, `<path>/net_output/`, `<path>/config.json` (reference CompositeGNN.py:87-108)."""
        if path[-1] != '/': path += '/'
        config = self.get_config()
        for i, elem in enumerate(config.pop("net_state")): self._save_net(elem, f'{path}net_state_{i}/')
        self._save_net(config.pop("net_output"), f'{path}net_output/')
        with open(f'{path}config.json', 'w') as json_file: json.dump(config, json_file)

    @classmethod
    def load(cls, path: str, *args, **kwargs):
        import os
        if path[-1] != '/': path += '/'
        with open(f'{path}config.json', 'r') as read_file: config = json.loads(read_file.read())
        n = len([d for d in os.listdir(path) if d.startswith('net_state_')])
        netS = [cls._load_net(f'{path}net_state_{i}/') for i in range(n)]
        return cls(net_state=netS, net_output=cls._load_net(f'{path}net_output/'), **config)

    def summary(self, *args, **kwargs):
        print(repr(self))
        for net in self.net_state + [self.net_output]:
            print('\n')
            net.summary(*args, **kwargs)

    @staticmethod
    def process_inputs(inputs):
        """Squeeze [2] dim_node_label, [3] type_mask, [4] set_mask, [5] output_mask; [6] list of composite adjacency
        triples and [7:] triples -> `SparseMatrix` (reference CompositeGNN.py:178-191)."""
        inputs = list(inputs)
        inputs[2:6] = [_squeeze_last(k) for k in inputs[2:6]]
        inputs[6] = [SparseMatrix.from_triple(k) for k in inputs[6]]
        inputs[7:] = [SparseMatrix.from_triple(k) for k in inputs[7:]]
        return inputs

    def apply_filters(self, state_converged, nodes, adjacency, arcs_label, mask):
        """State only, no label concat (reference CompositeGNN.py:237-239)."""
        return state_converged[mask]

    def convergence(self, k, state, state_old, nodes, dim_node_label, type_mask, adjacency, aggregated_component, training, *,
                    arcs=None, arcnode=None, composite_adjacencies=None):
        """One state-transition step of a heterogeneous graph with the reference's own nine positional arguments (reference
        CompositeGNN.py:215-234): per node type t

            state_new[type t rows] = net_state[t]([nodes[:, :d_t] | state | adjacency^T . state | aggregated_component][type t rows])

        through `torch.ops.gnnkeras.state_step` -> `gnn_state_step_agg` (every type's network on that type's rows, one fused launch);
        `aggregated_component` = [aggregated_nodes_0 | .. | aggregated_arcs] is used as handed in (CompositeGNN.py:251-253), `None`
        rebuilds it from `arcs=` / `arcnode=` / `composite_adjacencies=`.  `training=True` runs the state networks in training mode
        (batch statistics of each type's rows, moving-average update, Dropout).  Returns the reference's 9-tuple."""
        nat.require_device(nodes, 'nodes'); nat.require_device(state, 'state')
        dev = nodes.device
        training = bool(training)
        dims = [int(d) for d in (dim_node_label.reshape(-1).tolist() if isinstance(dim_node_label, torch.Tensor)
                                 else np.asarray(dim_node_label).reshape(-1))]
        if len(dims) != len(self.net_state): raise ValueError(f'{len(dims)} node types but {len(self.net_state)} state networks')
        type_nodes, offsets = self._type_lists(_squeeze_last(type_mask).to(dev))
        nodes32 = nodes.to(torch.float32).contiguous()
        state32 = state.to(dev, torch.float32).contiguous()
        adj = SparseMatrix.from_triple(adjacency).device_csr(dev)
        if aggregated_component is None:
            if arcs is None or arcnode is None or composite_adjacencies is None:
                raise ValueError('convergence() needs aggregated_component (the reference\'s argument), or arcs=, arcnode= and composite_adjacencies= to rebuild it')
            arcs32 = arcs.to(dev, torch.float32).contiguous()
            parts = [ops.aggregate(SparseMatrix.from_triple(c).device_csr(dev), nodes32[:, :d].contiguous()) for c, d in zip(composite_adjacencies, dims) if d > 0]
            parts.append(ops.aggregate(SparseMatrix.from_triple(arcnode).device_csr(dev), arcs32[:, 2:].contiguous()))
            aggregated_component = torch.cat(parts, dim=1)
        comp = aggregated_component.to(dev, torch.float32)
        if comp.stride(-1) != 1: comp = comp.contiguous()
        sum_d = sum(dims)
        N = nodes32.shape[0]
        if training and any(n.batch_normalization or n.dropout_rate for n in self.net_state):
            from .training import mlp_training_call
            agg = ops.aggregate(adj, state32)
            new = torch.zeros_like(state32)
            for t, net in enumerate(self.net_state):
                rows = type_nodes[int(offsets[t]):int(offsets[t + 1])].contiguous()
                if len(rows) == 0: continue
                segs = ([(nodes32[:, :dims[t]], rows)] if dims[t] > 0 else []) + [(state32, rows), (agg, rows)]
                if comp.shape[1] > 0: segs.append((comp, rows))
                new.index_copy_(0, rows.long(), mlp_training_call(net, segs, len(rows), net_id=t))
        else:
            dummy_arcs = nodes32.new_zeros((0, 2 + comp.shape[1] - sum_d))
            new, _moving = ops.state_step(nodes32, dummy_arcs, adj, None, self.net_state, state32, self.state_vect_dim, self.state_threshold,
                                          self.native_flags, composite=(type_nodes, offsets, dims, None),
                                          aggregated=(comp[:, :sum_d], comp[:, sum_d:]))
        return k + 1, new, state, nodes, dim_node_label, type_mask, adjacency, aggregated_component, training


    def _type_lists(self, type_mask: torch.Tensor):
        """(node ids grouped by type int32 [N] on device, host offsets [T+1]); cached per type_mask tensor."""
        key = (type_mask.data_ptr(), type_mask._version, tuple(type_mask.shape))
        hit = self._type_cache.get(key)
        if hit is None:
            tm = type_mask.to(torch.bool)
            if tm.dim() != 2: raise ValueError('type_mask must be (n_types, n_nodes)')
            per_node = tm.sum(0)
            if not bool(torch.all(per_node == 1)):
                raise ValueError('type_mask must be one-hot: every node needs exactly one type')
            t_idx, n_idx = torch.nonzero(tm, as_tuple=True)           # row-major: grouped by type, ascending node id
            counts = torch.bincount(t_idx, minlength=tm.shape[0]).cpu().numpy()
            offsets = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
            if len(self._type_cache) > 1024: self._type_cache.clear()
            hit = self._type_cache[key] = (n_idx.to(torch.int32).contiguous(), offsets, type_mask)
        return hit[0], hit[1]

    def Loop(self, nodes, arcs, dim_node_label, type_mask, set_mask, output_mask, composite_adjacencies, adjacency,
             arcnode, nodegraph, training: bool = False, *, state0=None, seed=None, node_level: bool = False):
        """(k, state, out) for one (merged) heterogeneous graph — reference CompositeGNN.py:242-272.
        `state0` / `seed` / `node_level` as in `GNNnodeBased.Loop`."""
        focus = 'n' if (node_level and self._focus == 'g') else self._focus
        if training:
            from .training import LoopTrainer
            if getattr(self, '_trainer', None) is None: self._trainer = LoopTrainer(self)
            tp = self._trainer.forward([nodes, arcs, dim_node_label, type_mask, set_mask, output_mask, composite_adjacencies,
                                        adjacency, arcnode, nodegraph], state0=state0, seed=seed, node_level=node_level)
            return torch.tensor(float(tp.k), device=tp.dev), tp.state.clone(), tp.y_pred
        nat.require_device(nodes, 'nodes'); nat.require_device(arcs, 'arcs')
        dev = nodes.device
        nodes = nodes.to(torch.float32).contiguous()
        arcs = arcs.to(torch.float32).contiguous()
        N, Lw = nodes.shape
        dims = [int(d) for d in (dim_node_label.reshape(-1).tolist() if isinstance(dim_node_label, torch.Tensor)
                                 else np.asarray(dim_node_label).reshape(-1))]
        T = len(dims)
        if T != len(self.net_state): raise ValueError(f'{T} node types but {len(self.net_state)} state networks')
        if T > nat.GNN_MAX_TYPES: raise ValueError(f'at most {nat.GNN_MAX_TYPES} node types are supported')
        type_mask = _squeeze_last(type_mask).to(dev)
        set_mask, output_mask = _squeeze_last(set_mask).to(dev), _squeeze_last(output_mask).to(dev)
        out_index = self._out_index(set_mask, output_mask)
        type_nodes, offsets = self._type_lists(type_mask)

        adjacency = SparseMatrix.from_triple(adjacency)
        adj, arcn = adjacency.device_csr(dev), SparseMatrix.from_triple(arcnode).device_csr(dev)
        cas = [SparseMatrix.from_triple(c).device_csr(dev) for c in composite_adjacencies]

        S = self.state_vect_dim if self.state_vect_dim > 0 else Lw
        if self.state_vect_dim > 0:
            if state0 is None:
                gen = None
                if seed is not None:
                    gen = torch.Generator(device=dev); gen.manual_seed(int(seed))
                state0 = torch.empty((N, S), device=dev, dtype=torch.float32).normal_(0.0, 0.1, generator=gen)      # tf.random.normal(stddev=0.1), one launch
            state0 = state0.to(dev, torch.float32).contiguous()
            if tuple(state0.shape) != (N, S): raise ValueError('state0 must be (n_nodes, state_vect_dim)')
        else:
            state0 = None
        ends = _arc_endpoints(adjacency, dev) if focus == 'a' else None
        ng = SparseMatrix.from_triple(nodegraph).device_csr(dev) if focus == 'g' else None
        # one custom op for the whole heterogeneous Loop: torch.ops.gnnkeras.loop_forward with the per-type lists
        k, state, out = ops.loop_forward(nodes, arcs, adj, arcn, ng, self.net_state, self.net_output, state0, out_index, ends,
                                         self.state_vect_dim, self.max_iteration, self.state_threshold, nat.FOCUS[focus],
                                         self.native_flags, composite=(type_nodes, offsets, dims, cas), loop_events=self.loop_events)
        self._last_k = k
        return k, state, out


class CompositeGNNarcBased(CompositeGNNnodeBased):
    """Composite GNN for arc-focused problems (reference CompositeGNN.py:310-327): [state_src | state_dst | arc label]."""
    name = "arc"
    _focus = 'a'

    def apply_filters(self, state_converged, nodes, adjacency, arcs_label, mask):
        adjacency = SparseMatrix.from_triple(adjacency)
        idx = torch.from_numpy(adjacency.indices).to(state_converged.device)
        states = state_converged[idx].reshape(arcs_label.shape[0], 2 * state_converged.shape[1])
        return torch.cat([states, arcs_label], dim=1)[mask]


class CompositeGNNgraphBased(CompositeGNNnodeBased):
    """Composite GNN for graph-focused problems (reference CompositeGNN.py:333-343)."""
    name = "graph"
    _focus = 'g'
