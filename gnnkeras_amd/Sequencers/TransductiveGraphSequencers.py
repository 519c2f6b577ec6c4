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
        config.pop("assemble", None)                               # (composite batches are merged on the host)
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
        """Heterogeneous version of `g` with non-transductive (type 0) / transductive (type 1) nodes (reference :62-95).

        The supervised nodes (`set_mask & output_mask`) are drawn in ONE `np.random.shuffle` of their ascending index list -
        the same consumption of numpy's global stream as the reference, so a seeded run draws the same split; the first
        ceil(n * (1 - rate)) of the shuffled list stay supervised, the others turn transductive: their target row moves behind
        their label, they get type 1 and leave the output mask."""
        supervised = np.flatnonzero(np.logical_and(g.set_mask, g.output_mask))
        np.random.shuffle(supervised)
        n_stay = int(np.ceil(len(supervised) * (1 - transductive_rate)))
        is_trans = np.zeros(len(g.set_mask), dtype=bool)
        is_trans[supervised[n_stay:]] = True
        # targets have one row per output_mask entry, in node order
        target_is_trans = is_trans[g.output_mask]

        n_rows = g.arcs.shape[0] if focus == 'a' else g.nodes.shape[0]
        L, T = g.nodes.shape[1], int(g.DIM_TARGET)
        labels = np.zeros((n_rows, L + T), dtype=np.result_type(g.nodes.dtype, dtype))
        labels[:, :L] = g.nodes
        labels[is_trans, L:] = g.targets[target_is_trans]

        d0 = int(np.asarray(g.DIM_NODE_LABEL).reshape(-1)[0])
        return CompositeGraphObject(arcs=g.getArcs(), nodes=labels, targets=g.targets[~target_is_trans],
                                    type_mask=np.stack([~is_trans, is_trans], axis=1), dim_node_label=(d0, d0 + T), focus=focus,
                                    set_mask=g.getSetMask(), output_mask=np.logical_and(g.output_mask, ~is_trans))


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
