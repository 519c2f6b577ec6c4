"""Transductive sequencers: homogeneous graphs turned into 2-type heterogeneous graphs on the host, then fed to the
composite device loop. Mirror of the reference's `GNN/Sequencers/TransductiveGraphSequencers.py`.

A random share (`transductive_rate`) of the targeted nodes becomes "transductive": their target is appended to their
label (type 1, label width L + T), they leave the output mask, and the remaining targeted nodes (type 0, label width L)
are the ones the model is supervised on. The split is re-drawn every epoch (`on_epoch_end`). Pure numpy preprocessing
(`get_transduction` consumes `np.random.shuffle` exactly like the reference, so a seeded run draws the same split);
the arithmetic runs in `CompositeGNN*` on the device. As in the reference, only node-focused graphs are meaningful here
(`get_transduction` indexes node-level masks, reference :66-80).
"""
from __future__ import annotations

import numpy as np

from ..composite_graph_class import CompositeGraphObject
from ..graph_class import GraphObject
from .GraphSequencers import CompositeMultiGraphSequencer, CompositeSingleGraphSequencer


class TransductiveMultiGraphSequencer(CompositeMultiGraphSequencer):
    """Sequencer for many homogeneous graphs, re-typed as transductive / non-transductive (reference :13-95)."""

    def __init__(self, graphs, focus: str, aggregation_mode: str, transductive_rate: float = 0.5, batch_size: int = 32,
                 shuffle: bool = True, device=None):
        self.graph_objects = graphs if isinstance(graphs, list) else [graphs]
        self.transductive_rate = transductive_rate
        gs = [self.get_transduction(g, transductive_rate, focus, 'float32') for g in self.graph_objects]
        super().__init__(gs, focus, aggregation_mode, batch_size, shuffle, device=device)

    def get_config(self):
        config = super().get_config()
        config["graphs"] = self.graph_objects                      # from_config must see the homogeneous originals
        config["transductive_rate"] = self.transductive_rate
        return config

    def __repr__(self):
        problem = {'a': 'edge', 'n': 'node', 'g': 'graph'}[self.focus]
        return f"transductive_graph_sequencer(multiple {problem}-focused, len={len(self)}, " \
               f"transductive_rate={self.transductive_rate}, aggregation='{self.aggregation_mode}', " \
               f"batch_size={self.batch_size}, shuffle={self.shuffle})"

    __str__ = __repr__

    def on_epoch_end(self):
        """Re-draw the transductive split of every graph, then reshuffle / re-merge (reference :56-59)."""
        self.data = [self.get_transduction(g, self.transductive_rate, self.focus, self.dtype) for g in self.graph_objects]
        if self.shuffle:
            super().on_epoch_end()
        else:
            self.build_batches()

    @staticmethod
    def get_transduction(g: GraphObject, transductive_rate: float, focus: str, dtype='float32'):
        """Heterogeneous version of `g` with non-transductive (type 0) / transductive (type 1) nodes (reference :62-95)."""
        transductive_node_mask = np.logical_and(g.set_mask, g.output_mask)
        indices = np.argwhere(transductive_node_mask).squeeze()
        np.random.shuffle(indices)
        non_transductive_number = int(np.ceil(np.sum(transductive_node_mask) * (1 - transductive_rate)))
        transductive_node_mask[indices[:non_transductive_number]] = False
        transductive_target_mask = transductive_node_mask[g.output_mask]

        length = g.arcs.shape[0] if focus == 'a' else g.nodes.shape[0]
        labelplus = np.zeros((length, g.DIM_TARGET), dtype=dtype)
        labelplus[transductive_node_mask] = g.targets[transductive_target_mask]
        nodes_new = np.concatenate([g.nodes, labelplus], axis=1)
        target_new = g.targets[np.logical_not(transductive_target_mask)]
        d0 = int(np.asarray(g.DIM_NODE_LABEL).reshape(-1)[0])
        dim_node_label_new = (d0, d0 + g.DIM_TARGET)

        type_mask = np.zeros((g.nodes.shape[0], 2), dtype=bool)
        type_mask[transductive_node_mask, 1] = True
        type_mask[:, 0] = np.logical_not(type_mask[:, 1])
        output_mask_new = g.output_mask.copy()
        output_mask_new[transductive_node_mask] = False
        return CompositeGraphObject(arcs=g.getArcs(), nodes=nodes_new, targets=target_new, type_mask=type_mask,
                                    dim_node_label=dim_node_label_new, focus=focus, set_mask=g.getSetMask(),
                                    output_mask=output_mask_new)


class TransductiveSingleGraphSequencer(TransductiveMultiGraphSequencer, CompositeSingleGraphSequencer):
    """Sequencer for one homogeneous graph, re-typed as transductive / non-transductive (reference :100-153)."""

    def __init__(self, graph: GraphObject, focus: str, transductive_rate: float = 0.5, batch_size: int = 32,
                 shuffle: bool = True, device=None):
        self.graph_object = graph
        self.transductive_rate = transductive_rate
        g = self.get_transduction(graph, transductive_rate, focus, 'float32')
        CompositeSingleGraphSequencer.__init__(self, g, focus, batch_size, shuffle, device=device)

    def copy(self):
        # the reference reads `self.trasductive_rate` here (typo, AttributeError at HEAD — SURVEY Q12)
        new_gen = self.__class__(self.graph_object.copy(), self.focus, self.transductive_rate, self.batch_size, False)
        new_gen.shuffle = self.shuffle
        return new_gen

    def get_config(self):
        return {"graph": self.graph_object, "focus": self.focus, "transductive_rate": self.transductive_rate,
                "batch_size": self.batch_size, "shuffle": self.shuffle}

    def __repr__(self):
        problem = {'a': 'edge', 'n': 'node', 'g': 'graph'}[self.focus]
        return f"transductive_graph_sequencer(type=single {problem}-focused, " \
               f"len={len(self)}, transductive_rate={self.transductive_rate}, " \
               f"batch_size={self.batch_size}, shuffle={self.shuffle})"

    __str__ = __repr__

    def on_epoch_end(self):
        """New transductive split of the graph + reshuffled set_mask batches (reference :149-153)."""
        g = self.get_transduction(self.graph_object, self.transductive_rate, self.focus, self.dtype)
        self.data = g
        self.graph_tensor = self.to_graph_tensor(g, self.device)
        self.set_mask_idx = np.argwhere(self.data.set_mask).reshape(-1)
        if self.shuffle: np.random.shuffle(self.set_mask_idx)
        self.build_batches()
