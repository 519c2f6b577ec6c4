"""Batch assembly for the MI355X loop: merge B graphs into one block-diagonal graph, once, and keep it in HBM.

Mirror of the reference's `GNN/Sequencers/GraphSequencers.py` (`tf.keras.utils.Sequence` subclasses): same
constructors, `__len__`, `__getitem__ -> (x_list, targets, sample_weight)`, `on_epoch_end`, `set_batch_size`,
`get_batch`, `copy`, `get_config/from_config`. `x_list` has the reference's layout
(`GraphSequencers.py:108-120`, `:239-244`):

  homogeneous : [nodes, arcs, dim_node_label(1,1), set_mask(N,1), output_mask(N,1), Adjacency, ArcNode, NodeGraph]
  composite   : [nodes, arcs, dim_node_label(T,1), type_mask(T,N,1), set_mask, output_mask, [CompositeAdjacency]*T,
                 Adjacency, ArcNode, NodeGraph]

with each sparse matrix as the `(indices, values[...,None], dense_shape)` triple — a `SparseTriple` that also carries
the by-destination CSR built at merge time, so the model never re-derives it (the reference rebuilds three
`tf.SparseTensor`s per call, `GNN.py:192`).
"""
from __future__ import annotations

import numpy as np
import torch

from ..composite_graph_class import CompositeGraphObject, CompositeGraphTensor
from ..graph_class import GraphObject, GraphTensor
from ..sparse import default_device


class MultiGraphSequencer:
    """Sequencer for a dataset of many homogeneous graphs (reference GraphSequencers.py:12-127)."""

    merge = classmethod(lambda cls, *a, **k: GraphObject.merge(*a, **k))
    to_graph_tensor = classmethod(lambda cls, g, device=None: GraphTensor.fromGraphObject(g, device))

    def __init__(self, graphs, focus: str, aggregation_mode: str, batch_size: int = 32, shuffle: bool = True,
                 device=None, assemble: str = 'auto'):
        """`assemble` (additive): where batches are merged — 'host' (numpy `GraphObject.merge`, then upload), 'device' (the data
        set is uploaded once, a batch is ONE ragged-copy launch: `gnnkeras_amd/device_batch.py`) or 'auto' = device on a GPU
        for homogeneous data sets it supports, host otherwise. Same arrays either way."""
        self.data = graphs if isinstance(graphs, list) else [graphs]
        self.focus = focus
        self.aggregation_mode = aggregation_mode
        self.batch_size = int(batch_size)
        self.shuffle = shuffle
        self.dtype = 'float32'
        self.device = torch.device(device) if device is not None else default_device()
        if assemble not in ('auto', 'host', 'device'): raise ValueError("assemble must be 'auto', 'host' or 'device'")
        self.assemble = assemble
        self._dataset = None
        self.build_batches()

    def _device_dataset(self):
        """The device-resident data set, or None when batches are merged on the host."""
        if self.assemble == 'host' or self.device.type != 'cuda' or type(self).merge.__func__ is not MultiGraphSequencer.merge.__func__:
            if self.assemble == 'device': raise ValueError('device assembly needs a GPU and a homogeneous data set')
            return None
        # the data set on the device is valid as long as the graphs still hold the arrays it was built from (LGNN's serial fit
        # REPLACES the label arrays of a sequencer's graphs: reference LGNN.py:318-333); edits made IN PLACE inside an array are
        # not seen - call refresh() after such an edit
        arrays = lambda g: (id(g.nodes), id(g.arcs), id(g.targets), id(g.set_mask), id(g.output_mask))
        # (the signature is keyed by graph OBJECT: order-independent, so shuffling keeps it valid; a list that holds the same
        # object twice has fewer keys than entries, which is fine - compare against the number of distinct objects)
        stale = self._dataset is None or len(self._dataset[1]) != len({id(g) for g in self.data}) or \
            any(self._dataset[1].get(id(g)) != arrays(g) for g in self.data)
        if stale:
            sig = {id(g): arrays(g) for g in self.data}
            from ..device_batch import DeviceDataset
            try:
                ds = DeviceDataset(self.data, self.focus, self.aggregation_mode, self.device)
            except ValueError:
                if self.assemble == 'device': raise
                ds = None
            if ds is not None and ds.hub:
                if self.assemble == 'device': raise ValueError('hub rows (in-degree > 512) need the host-side split')
                ds = None
            self._dataset = (ds, sig, {id(g): i for i, g in enumerate(self.data)})
        return self._dataset[0]

    def refresh(self):
        """Rebuild everything from the current contents of `self.data` (after editing a graph's arrays in place)."""
        self._dataset = None
        self.build_batches()

    def build_batches(self):
        """Slice the graph list by batch_size, merge each slice, have it on the device (reference :42-46)."""
        ds = self._device_dataset()
        if ds is not None:
            index = self._dataset[2]
            order = [index[id(g)] for g in self.data]
            if ds is not None:
                self.graph_tensors = ds.assemble_many([order[i * self.batch_size: (i + 1) * self.batch_size] for i in range(len(self))])
                self._items = [None] * len(self.graph_tensors)
                return
        graphs = [self.merge(self.data[i * self.batch_size: (i + 1) * self.batch_size], focus=self.focus,
                             aggregation_mode=self.aggregation_mode) for i in range(len(self))]
        self.graph_tensors = [self.to_graph_tensor(g, self.device) for g in graphs]
        self._items = [None] * len(self.graph_tensors)

    def merged_batches(self, i0: int, i1: int = None):
        """(x_list, node_begin) of batches i0 .. i1-1 - or of the batches in the list `i0` - merged into ONE graph, graphs in the
        same order (additive; the reference has no counterpart).  `model.call(x_list, groups=node_begin)` runs those batches as
        independent loops of one launch - each batch converges and stops on its own, as if called alone - which is how
        predict() / evaluate() fill the GPU with small batches.  None when a bigger merge would change the operands:
        'normalized' divides by the number of arcs of the merged graph (graph_class.py buildArcNode).  Cached
        until the batches are rebuilt."""
        if self.aggregation_mode == 'normalized' or type(self).merge.__func__ is not MultiGraphSequencer.merge.__func__: return None
        batches = [int(b) for b in i0] if i1 is None else list(range(int(i0), int(i1)))
        key = tuple(batches)                                    # (a range and the list of its members are the same merge)
        cache = self.__dict__.setdefault('_merged', {})
        if cache.get('owner') is not self.graph_tensors: cache.clear(); cache['owner'] = self.graph_tensors
        if key not in cache:
            of = lambda b: self.data[b * self.batch_size: (b + 1) * self.batch_size]
            g = self._assemble_graphs([g_ for b in batches for g_ in of(b)])
            sizes = [sum(int(g_.nodes.shape[0]) for g_ in of(b)) for b in batches]
            cache[key] = (self._x_list(g), [0] + [int(v) for v in np.cumsum(sizes)])
        return cache[key]

    def _assemble_graphs(self, graphs):
        """The merged graph of `graphs` (in that order) on the device: one ragged-copy launch where the data set lives on the
        device, numpy merge + upload otherwise."""
        ds = self._device_dataset()
        if ds is not None:
            index = self._dataset[2]
            return ds.assemble([index[id(g_)] for g_ in graphs])
        return self.to_graph_tensor(self.merge(graphs, focus=self.focus, aggregation_mode=self.aggregation_mode), self.device)

    def shard_item(self, index: int, rank: int, world_size: int):
        """(x_list, targets, sample_weight) of rank `rank`'s share of batch `index` - a contiguous, balanced run of WHOLE graphs
        of the batch, merged like any batch (additive; what `gnnkeras_amd.data_parallel.DataParallel` trains on: a merged batch is
        block-diagonal, reference graph_class.py:399-408, so it shards by graph with no halo).  The shares of all ranks, in rank
        order, are the batch.  'normalized' aggregation cannot be sharded: its weights are 1 / #arcs of the whole batch."""
        if self.aggregation_mode == 'normalized':
            raise ValueError("'normalized' aggregation divides by the arc count of the whole merged batch: not shardable by graph")
        graphs = self.data[index * self.batch_size: (index + 1) * self.batch_size]
        lo, hi = len(graphs) * rank // world_size, len(graphs) * (rank + 1) // world_size
        if hi <= lo: raise ValueError(f'batch {index} has {len(graphs)} graphs: nothing left for rank {rank} of {world_size}')
        cache = self.__dict__.setdefault('_merged', {})
        if cache.get('owner') is not self.graph_tensors: cache.clear(); cache['owner'] = self.graph_tensors
        key = ('shard', int(index), int(rank), int(world_size))
        if key not in cache:
            g = self._assemble_graphs(graphs[lo:hi])
            if self.focus == 'g':
                cache[key] = (self._x_list(g), g.targets, g.sample_weight)
            else:
                mask = g.set_mask[g.output_mask]
                cache[key] = (self._x_list(g), g.targets[mask], g.sample_weight[mask])
        return cache[key]

    def copy(self):
        config = self.get_config()
        config["graphs"] = [g.copy() for g in config["graphs"]]
        return self.from_config(config)

    def _view(self):
        """A second sequencer over the SAME graph objects (its own list, batch size and order): for a consumer that re-batches and shuffles
        but does not edit the graphs - `fit()` of a GNN - where the reference hands over `copy()` (LGNN.py:312-313)."""
        config = self.get_config()
        config["graphs"] = list(config["graphs"])
        return self.from_config(config)

    def get_config(self):
        return {"graphs": self.data, "focus": self.focus, "aggregation_mode": self.aggregation_mode,
                "batch_size": self.batch_size, "shuffle": self.shuffle, "assemble": self.assemble}

    @classmethod
    def from_config(cls, config, **kwargs):
        return cls(**config)

    def __repr__(self):
        problem = {'a': 'edge', 'n': 'node', 'g': 'graph'}[self.focus]
        return f"graph_sequencer(type=multiple {problem}-focused, len={len(self)}, " \
               f"aggregation='{self.aggregation_mode}', batch_size={self.batch_size}, shuffle={self.shuffle})"

    __str__ = __repr__

    def set_batch_size(self, new_batch_size):
        self.batch_size = new_batch_size
        self.build_batches()

    def get_batch(self, index):
        g = self.graph_tensors[index]
        return g, g.set_mask

    def __len__(self):
        return int(np.ceil(len(self.data) / self.batch_size))

    def _x_list(self, g):
        newaxis = lambda x: x[..., None]
        return [g.nodes, g.arcs] + [newaxis(i) for i in [g.DIM_NODE_LABEL, g.set_mask, g.output_mask]] + \
               [m.triple(self.device) for m in [g.Adjacency, g.ArcNode, g.NodeGraph]]

    def __getitem__(self, index):
        """(x_list, targets, sample_weight) of batch `index`; device tensors, built once per batch and reused."""
        if self._items[index] is not None:
            return self._items[index]
        g, set_mask = self.get_batch(index)
        out = self._x_list(g)
        if self.focus == 'g':                   # every target row (reference :112); no boolean indexing = no host synchronisation
            item = (out, g.targets, g.sample_weight)
        else:
            mask = set_mask[g.output_mask]
            item = (out, g.targets[mask], g.sample_weight[mask])
        if set_mask is g.set_mask: self._items[index] = item
        return item

    def on_epoch_end(self):
        """Reshuffle the graph list and re-merge every batch (reference :123-127)."""
        if self.shuffle:
            np.random.shuffle(self.data)
            self.build_batches()


class SingleGraphSequencer(MultiGraphSequencer):
    """Sequencer for a dataset made of one homogeneous graph: a batch is a subset of set_mask
    (reference GraphSequencers.py:133-208). As in the reference, `x` carries the graph's full set_mask while targets are
    filtered with the batch mask — kept faithfully."""

    def __init__(self, graph: GraphObject, focus: str, batch_size: int = 32, shuffle: bool = True, device=None):
        self.data = graph
        self.device = torch.device(device) if device is not None else default_device()
        self.graph_tensor = self.to_graph_tensor(graph, self.device)
        self.focus = focus
        self.batch_size = batch_size
        self.shuffle = shuffle
        self.dtype = 'float32'
        self.aggregation_mode = getattr(graph, 'aggregation_mode', None)
        self.assemble = 'host'
        self.set_mask_idx = np.argwhere(self.data.set_mask).reshape(-1)
        self.build_batches()

    def merged_batches(self, i0, i1=None):
        """Every batch of a single-graph sequencer IS the whole graph (with another target subset): nothing to merge."""
        return None

    def build_batches(self):
        self.batch_masks = np.zeros((len(self), len(self.data.set_mask)), dtype=bool)
        for i in range(len(self)):
            self.batch_masks[i, self.set_mask_idx[i * self.batch_size: (i + 1) * self.batch_size]] = True
        self._items = [None] * len(self)

    def copy(self):
        config = self.get_config()
        config["graph"] = config["graph"].copy()
        return self.from_config(config)

    def get_config(self):
        return {"graph": self.data, "focus": self.focus, "batch_size": self.batch_size, "shuffle": self.shuffle}

    def __repr__(self):
        problem = {'a': 'edge', 'n': 'node', 'g': 'graph'}[self.focus]
        return f"graph_sequencer(type=single {problem}-focused, " \
               f"len={len(self)}, batch_size={self.batch_size}, shuffle={self.shuffle})"

    __str__ = __repr__

    def get_batch(self, index):
        return self.graph_tensor, torch.as_tensor(self.batch_masks[index], dtype=torch.bool, device=self.device)

    def __len__(self):
        return int(np.ceil(np.sum(self.data.set_mask) / self.batch_size))

    def on_epoch_end(self):
        if self.shuffle:
            np.random.shuffle(self.set_mask_idx)
            self.build_batches()


class CompositeMultiGraphSequencer(MultiGraphSequencer):
    """Sequencer for many heterogeneous graphs (reference GraphSequencers.py:214-245)."""

    merge = classmethod(lambda cls, *a, **k: CompositeGraphObject.merge(*a, **k))
    to_graph_tensor = classmethod(lambda cls, g, device=None: CompositeGraphTensor.fromGraphObject(g, device))

    def __init__(self, graphs, *args, **kwargs):
        super().__init__(graphs, *args, **kwargs)

    def __repr__(self):
        return f"composite_{super().__repr__()}"

    __str__ = __repr__

    def _x_list(self, g):
        out = super()._x_list(g)
        out.insert(3, g.type_mask[..., None])
        out.insert(-3, [ca.triple(self.device) for ca in g.CompositeAdjacencies])
        return out


class CompositeSingleGraphSequencer(SingleGraphSequencer, CompositeMultiGraphSequencer):
    """Sequencer for one heterogeneous graph (reference GraphSequencers.py:252-266)."""

    def __init__(self, graph: CompositeGraphObject, *args, **kwargs):
        SingleGraphSequencer.__init__(self, graph, *args, **kwargs)

    def __repr__(self):
        return f"composite_{super().__repr__()}"

    __str__ = __repr__
