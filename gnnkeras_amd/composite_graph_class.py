"""Heterogeneous (typed-node) graph containers for the per-type state networks of the Composite GNN loop.

Host-side mirror of the reference's `GNN/composite_graph_class.py` (same names / arguments / errors). On top of the
homogeneous containers it adds: `type_mask` (N, T) one-hot node types, per-type label widths `DIM_NODE_LABEL`, the
list of T `CompositeAdjacencies` (Adjacency restricted to sources of type t) and the 'composite_average' aggregation.
"""
from __future__ import annotations

import numpy as np
import torch
from scipy.sparse import coo_matrix

from .graph_class import GraphObject, GraphTensor
from .sparse import SparseMatrix


class CompositeGraphObject(GraphObject):
    """Heterogeneous graph (reference composite_graph_class.py:14-185)."""

    def __init__(self, nodes, arcs, targets, type_mask, dim_node_label, *args, **kwargs):
        # type_mask[:, t] marks nodes whose label has width dim_node_label[t]; needed by buildArcNode, so set first.
        self.type_mask = np.asarray(type_mask).astype(bool)
        super().__init__(nodes, arcs, targets, *args, **kwargs)
        self.DIM_NODE_LABEL = np.array(dim_node_label, ndmin=1, dtype=int)
        self.CompositeAdjacencies = self.buildCompositeAdjacency()

    def _source_type_masks(self):
        """bool (T, E): arc e leaves a node of type t."""
        return self.type_mask[self.arc_ids[:, 0]].transpose()

    def buildCompositeAdjacency(self):
        """CA[t][i, j] = Adjacency[i, j] iff node i has type t (reference composite_graph_class.py:57-70);
        explicit zeros are dropped like `eliminate_zeros()` there."""
        n = self.nodes.shape[0]
        out = []
        for src_is_t in self._source_type_masks():
            keep = src_is_t & (self.Adjacency.data != 0)
            out.append(coo_matrix((self.Adjacency.data[keep], (self.arc_ids[keep, 0], self.arc_ids[keep, 1])),
                                  shape=(n, n), dtype=self.dtype))
        return out

    def buildArcNode(self, aggregation_mode):
        """Adds 'composite_average': w_e = 1 / #(in-neighbours of dst_e having the type of src_e)
        (reference composite_graph_class.py:73-103)."""
        if aggregation_mode not in ['normalized', 'average', 'sum', 'composite_average']:
            raise ValueError("ERROR: Unknown aggregation mode")
        if aggregation_mode != 'composite_average':
            return super().buildArcNode(aggregation_mode)
        matrix = super().buildArcNode('sum')
        dst = self.arc_ids[:, 1]
        n = self.nodes.shape[0]
        for src_is_t in self._source_type_masks():
            if not np.any(src_is_t): continue
            per_dst = np.bincount(dst[src_is_t], minlength=n)
            matrix.data[src_is_t] /= per_dst[dst[src_is_t]]
        return matrix

    def copy(self):
        return CompositeGraphObject(arcs=self.getArcs(), nodes=self.getNodes(), targets=self.getTargets(),
                                    set_mask=self.getSetMask(), output_mask=self.getOutputMask(),
                                    sample_weight=self.getSampleWeights(), NodeGraph=self.getNodeGraph(),
                                    aggregation_mode=self.aggregation_mode, dim_node_label=self.DIM_NODE_LABEL,
                                    type_mask=self.getTypeMask())

    def __repr__(self):
        return f"composite_{super().__repr__()}"

    __str__ = __repr__

    def setAggregation(self, aggregation_mode: str):
        super().setAggregation(aggregation_mode)
        self.CompositeAdjacencies = self.buildCompositeAdjacency()

    def getTypeMask(self):
        return self.type_mask.copy()

    def get_dict_data(self):
        data = super().get_dict_data()
        data['type_mask'] = self.type_mask
        data['dim_node_label'] = self.DIM_NODE_LABEL
        return data

    @classmethod
    def merge(cls, glist, focus: str, aggregation_mode: str, dtype='float32'):
        """Block-diagonal merge of typed graphs (reference composite_graph_class.py:142-167)."""
        dims = set(tuple(int(d) for d in g.DIM_NODE_LABEL) for g in glist)
        assert len(dims) == 1, "DIM_NODE_LABEL not unique among graphs in :param glist:"
        nodes, arcs, targets, set_mask, output_mask, sample_weight, nodegraph = cls._merge_arrays(glist, dtype)
        type_mask = np.concatenate([g.type_mask for g in glist], axis=0, dtype=bool)
        return CompositeGraphObject(arcs=arcs, nodes=nodes, targets=targets, type_mask=type_mask,
                                    dim_node_label=dims.pop(), focus=focus, set_mask=set_mask,
                                    output_mask=output_mask, sample_weight=sample_weight, NodeGraph=nodegraph,
                                    aggregation_mode=aggregation_mode)

    @classmethod
    def fromGraphTensor(cls, g, focus: str):
        nodegraph = g.NodeGraph.to_scipy() if focus == 'g' else None
        t = lambda x: x.detach().cpu().numpy()
        return cls(arcs=t(g.arcs), nodes=t(g.nodes), targets=t(g.targets), dim_node_label=g.DIM_NODE_LABEL.numpy(),
                   type_mask=t(g.type_mask).transpose(), set_mask=t(g.set_mask), output_mask=t(g.output_mask),
                   sample_weight=t(g.sample_weight), NodeGraph=nodegraph, aggregation_mode=g.aggregation_mode,
                   focus=focus)


class CompositeGraphTensor(GraphTensor):
    """Device-resident CompositeGraphObject (reference composite_graph_class.py:188-264).
    `type_mask` is stored transposed, (T, N), as the reference does at `:263`."""

    def __init__(self, *args, type_mask, CompositeAdjacencies, **kwargs):
        super().__init__(*args, **kwargs)
        tm = type_mask if isinstance(type_mask, torch.Tensor) else torch.as_tensor(np.ascontiguousarray(type_mask))
        self.type_mask = tm.to(self.device, torch.bool)
        self.CompositeAdjacencies = [SparseMatrix.from_triple(i) for i in CompositeAdjacencies]

    def copy(self):
        return CompositeGraphTensor(nodes=self.nodes.clone(), dim_node_label=self.DIM_NODE_LABEL.numpy(),
                                    arcs=self.arcs.clone(), targets=self.targets.clone(),
                                    set_mask=self.set_mask.clone(), output_mask=self.output_mask.clone(),
                                    sample_weight=self.sample_weight.clone(), Adjacency=self.Adjacency.copy(),
                                    ArcNode=self.ArcNode.copy(), NodeGraph=self.NodeGraph.copy(),
                                    aggregation_mode=self.aggregation_mode, type_mask=self.type_mask.clone(),
                                    CompositeAdjacencies=[i.copy() for i in self.CompositeAdjacencies],
                                    device=self.device)

    def __repr__(self):
        return f"composite_{super().__repr__()}"

    __str__ = __repr__

    @staticmethod
    def save_graph(graph_npz_path: str, g, compressed: bool = False, **kwargs) -> None:
        data = {'type_mask': g.type_mask.detach().cpu().numpy()}
        for idx, mat in enumerate(g.CompositeAdjacencies):
            data[f"CompositeAdjacencies_{idx}"] = np.concatenate([mat.values[:, None], mat.indices.astype(np.float32)],
                                                                 axis=1)
        GraphTensor.save_graph(graph_npz_path, g, compressed, **data, **kwargs)

    @classmethod
    def load(cls, graph_npz_path, device=None, **kwargs):
        if '.npz' not in graph_npz_path: graph_npz_path += '.npz'
        data = dict(np.load(graph_npz_path, **kwargs))
        data['aggregation_mode'] = str(data['aggregation_mode'])
        for i in ['Adjacency', 'ArcNode', 'NodeGraph']:
            data[i] = SparseMatrix(data[i][:, 1:].astype(np.int64), data[i][:, 0], data.pop(i + '_shape'))
        CA = [data.pop(f"CompositeAdjacencies_{idx}") for idx, _ in enumerate(data['dim_node_label'])]
        CA = [SparseMatrix(adj[:, 1:].astype(np.int64), adj[:, 0], data['Adjacency'].shape) for adj in CA]
        return cls(**data, CompositeAdjacencies=CA, device=device)

    @classmethod
    def fromGraphObject(cls, g: CompositeGraphObject, device=None):
        return cls(nodes=g.nodes, dim_node_label=g.DIM_NODE_LABEL, arcs=g.arcs, targets=g.targets,
                   set_mask=g.set_mask, output_mask=g.output_mask, sample_weight=g.sample_weight,
                   Adjacency=cls.COO2SparseTensor(g.Adjacency), ArcNode=cls.COO2SparseTensor(g.ArcNode),
                   NodeGraph=cls.COO2SparseTensor(g.NodeGraph), aggregation_mode=g.aggregation_mode,
                   type_mask=g.type_mask.transpose(),
                   CompositeAdjacencies=[cls.COO2SparseTensor(i) for i in g.CompositeAdjacencies], device=device)
