"""Layered GNN over Composite GNNs (reference `GNN/Models/CompositeLGNN.py`): same label plumbing as `LGNN`, with the
10-argument composite `Loop` (type_mask, CompositeAdjacencies). `update_graph` widens every node type's label by the
same amount (`dim_node_label + plus`, one entry per type)."""
from __future__ import annotations

from .CompositeGNN import CompositeGNNnodeBased, CompositeGNNarcBased, CompositeGNNgraphBased
from .LGNN import LGNN


class CompositeLGNN(LGNN):
    """Composite Layered GNN for node-, arc- or graph-focused problems (reference CompositeLGNN.py:12-57)."""
    _gnn_classes = {"node": CompositeGNNnodeBased, "arc": CompositeGNNarcBased, "graph": CompositeGNNgraphBased}
    process_inputs = staticmethod(CompositeGNNnodeBased.process_inputs)

    def __repr__(self):
        return f"Composite{super().__repr__()}"

    __str__ = __repr__

    def Loop(self, nodes, arcs, dim_node_label, type_mask, set_mask, output_mask, composite_adjacencies, adjacency, arcnode,
             nodegraph, training: bool = False, *, state0=None, seed=None):
        """Lists (K, states, outs), one entry per layer (reference CompositeLGNN.py:25-57)."""
        constant_inputs = [type_mask, set_mask, output_mask, composite_adjacencies, adjacency, arcnode, nodegraph]
        nodes_0, arcs_0 = nodes, arcs
        s0 = state0 if state0 is not None else [None] * self.LAYERS
        K, states, outs = [], [], []
        graph_based = self.GNN_CLASS is self._gnn_classes['graph']
        for idx, gnn in enumerate(self.gnns[:-1]):
            k, state, out = gnn.Loop(nodes, arcs, dim_node_label, *constant_inputs, training=training, state0=s0[idx],
                                     seed=seed, node_level=True)
            K.append(k); states.append(state)
            outs.append(self._pool(nodegraph, out) if graph_based else out)
            nodes, arcs, dim_node_label = self.update_graph(nodes_0, arcs_0, dim_node_label, set_mask, output_mask, state, out)
        k, state, out = self.gnns[-1].Loop(nodes, arcs, dim_node_label, *constant_inputs, training=training,
                                           state0=s0[-1], seed=seed)
        return K + [k], states + [state], outs + [out]
