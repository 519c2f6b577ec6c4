"""Homogeneous graph containers feeding the MI355X message-passing loop.

Host-side mirror of the reference's `GNN/graph_class.py` interface (same class / method / attribute names, argument
meaning and error behaviour), re-designed for the HIP path:

* `GraphObject`  — numpy + scipy COO, like the reference (`graph_class.py:13-427`), but node ids are kept as int64
  next to the float `arcs` matrix (the reference stores ids in float32 columns, exact only below 2**24 — SURVEY Q4),
  and every constructor is vectorised (the reference walks arcs with Python `zip`, `graph_class.py:87`, `:554`).
* `GraphTensor`  — device-resident mirror (`graph_class.py:433-560`): dense labels / masks as torch tensors in HBM,
  the three sparse operators as `SparseMatrix` (COO triple + cached by-destination CSR, see `sparse.py`).
"""
from __future__ import annotations

import os
import shutil

import numpy as np
import torch
from scipy.sparse import coo_matrix, block_diag

from .sparse import SparseMatrix, default_device

FLOATX = 'float32'   # tf.keras.backend.floatx() default, reference graph_class.py:43


def _unique_rows(arcs: np.ndarray) -> np.ndarray:
    """`np.unique(arcs, axis=0)` (reference graph_class.py:47): rows sorted lexicographically, duplicates dropped.
    Fast path: rows whose (src, dst) pairs are already strictly increasing are returned untouched."""
    arcs = np.asarray(arcs)
    if arcs.shape[0] < 2:
        return arcs.copy()
    a, b = arcs[:-1, :2], arcs[1:, :2]
    if np.all((a[:, 0] < b[:, 0]) | ((a[:, 0] == b[:, 0]) & (a[:, 1] < b[:, 1]))):
        return arcs.copy()
    order = np.lexsort(arcs.T[::-1])
    s = arcs[order]
    keep = np.ones(len(s), dtype=bool)
    keep[1:] = np.any(s[1:] != s[:-1], axis=1)
    return s[keep]


def _coo_clone(m):
    """A copy of a COO matrix without the constructor's index validation (its arrays are copied; 1 us against 30)."""
    c = m.copy() if not isinstance(m, coo_matrix) else None
    if c is not None: return c
    try:
        c = coo_matrix.__new__(coo_matrix)
        c.__dict__.update(m.__dict__)
        c.data = m.data.copy()
        c.coords = tuple(x.copy() for x in m.coords)
        return c
    except Exception:                     # (another scipy layout: the public way)
        return m.copy()


class GraphObject:
    """Homogeneous graph: nodes (N, L) labels, arcs (E, 2 + A) = [src id | dst id | arc label], targets.

    Same constructor contract as the reference (`graph_class.py:17-79`)."""

    def __init__(self, nodes, arcs, targets, focus: str = 'n', set_mask=None, output_mask=None, sample_weight=1,
                 ArcNode=None, NodeGraph=None, aggregation_mode: str = 'sum'):
        self.dtype = FLOATX
        nodes, arcs, targets = np.asarray(nodes), np.asarray(arcs), np.asarray(targets)

        uarcs = _unique_rows(arcs)
        self.nodes = nodes.astype(self.dtype)
        self.arc_ids = uarcs[:, :2].astype(np.int64)            # exact ids (the reference keeps float32 only)
        self.arcs = uarcs.astype(self.dtype)
        self.targets = targets.astype(self.dtype)
        self.sample_weight = sample_weight * np.ones(self.targets.shape[0])

        self.DIM_NODE_LABEL = np.array(nodes.shape[1], ndmin=1, dtype=int)
        self.DIM_ARC_LABEL = arcs.shape[1] - 2
        self.DIM_TARGET = targets.shape[1]

        # mask length follows the focus; for 'a' it is the number of arcs as passed in (graph_class.py:57)
        lenMask = {'n': nodes.shape[0], 'a': arcs.shape[0], 'g': nodes.shape[0]}
        self.set_mask = np.ones(lenMask[focus], dtype=bool) if set_mask is None else np.asarray(set_mask).astype(bool)
        self.output_mask = np.ones(len(self.set_mask), dtype=bool) if output_mask is None \
            else np.asarray(output_mask).astype(bool)
        if len(self.set_mask) != len(self.output_mask):
            raise ValueError('Error - len(<set_mask>) != len(<output_mask>)')

        self.aggregation_mode = str(aggregation_mode)
        self.ArcNode = self.buildArcNode(self.aggregation_mode) if ArcNode is None \
            else coo_matrix(ArcNode, dtype=self.dtype)
        self.Adjacency = self.buildAdjacency()
        self.NodeGraph = self.buildNodeGraph(focus) if NodeGraph is None else coo_matrix(NodeGraph, dtype=self.dtype)

    # ------------------------------------------------------------------------------------------------------------------
    def buildAdjacency(self):
        """ADJ[src_e, dst_e] = ArcNode value of arc e (reference graph_class.py:82-88)."""
        n = self.nodes.shape[0]
        return coo_matrix((self.ArcNode.data, (self.arc_ids[:, 0], self.arc_ids[:, 1])), shape=(n, n),
                          dtype=self.dtype)

    def buildArcNode(self, aggregation_mode):
        """AN[e, dst_e] = w_e with w = 1 ('sum'), 1/#arcs ('normalized', sic: arcs not nodes — SURVEY Q5) or
        1/in-degree(dst_e) ('average')  (reference graph_class.py:91-124)."""
        if aggregation_mode not in ['sum', 'normalized', 'average']:
            raise ValueError("ERROR: Unknown aggregation mode")
        col = self.arc_ids[:, 1]
        row = np.arange(0, len(col))
        values_vector = np.ones(len(col))
        if aggregation_mode == 'normalized':
            values_vector = values_vector * float(1 / len(col))
        elif aggregation_mode == 'average':
            in_degree = np.bincount(col, minlength=self.nodes.shape[0])
            values_vector = values_vector / in_degree[col]
        return coo_matrix((values_vector, (row, col)), shape=(self.arcs.shape[0], self.nodes.shape[0]),
                          dtype=self.dtype)

    def buildNodeGraph(self, focus: str):
        """N x 1 column of 1/N for graph focus, empty otherwise (reference graph_class.py:127-138)."""
        if focus == 'g':
            data = np.ones((self.nodes.shape[0], 1)) * (1 / self.nodes.shape[0])
        else:
            data = np.array([], ndmin=2)
        return coo_matrix(data, dtype=self.dtype)

    # ------------------------------------------------------------------------------------------------------------------
    def copy(self):
        """A deep copy THROUGH THE CONSTRUCTOR, as in the reference (graph_class.py:141-146): arcs de-duplicated and ordered again, ArcNode and
        Adjacency rebuilt from `aggregation_mode`, the DIM_* fields read from the arrays' shapes.  When the constructor would find everything
        as it is - the arcs still the unique ordered rows their ids say, ArcNode what `buildArcNode` builds - the same object is made field by
        field without it (a serial LGNN `fit()` copies every graph of a sequencer a dozen times per epoch: 230 -> 25 us a graph)."""
        fast = self._copy_fields()
        if fast is not None: return fast
        return GraphObject(nodes=self.getNodes(), arcs=self.getArcs(), targets=self.getTargets(),
                           set_mask=self.getSetMask(), output_mask=self.getOutputMask(),
                           sample_weight=self.getSampleWeights(), NodeGraph=self.getNodeGraph(),
                           aggregation_mode=self.aggregation_mode)

    def _copy_fields(self):
        if type(self) is not GraphObject: return None
        arcs, ids, an = self.arcs, self.arc_ids, self.ArcNode
        E, N = arcs.shape[0], self.nodes.shape[0]
        if arcs.ndim != 2 or ids.shape != (E, 2) or an.shape != (E, N) or an.data.shape[0] != E or self.Adjacency.shape != (N, N): return None
        if self.nodes.dtype != self.dtype or arcs.dtype != self.dtype or self.targets.dtype != self.dtype: return None
        if self.set_mask.dtype != bool or self.output_mask.dtype != bool or len(self.set_mask) != len(self.output_mask): return None
        if not np.array_equal(arcs[:, :2], ids): return None                                   # the ids the arcs carry are the ids the operators were built on
        if E > 1:                                                                               # ... and the rows are unique and in np.unique's order
            d = np.diff(arcs, axis=0)
            first = (d != 0).argmax(axis=1)
            if not ((d != 0).any(axis=1) & (d[np.arange(E - 1), first] > 0)).all(): return None
        mode = self.aggregation_mode
        if mode not in ('sum', 'normalized', 'average'): return None
        col = ids[:, 1]
        want = np.ones(E)
        if mode == 'normalized': want = want * float(1 / max(E, 1))
        elif mode == 'average': want = want / np.bincount(col, minlength=N)[col]
        if not (np.array_equal(an.row, np.arange(E)) and np.array_equal(an.col, col) and np.array_equal(an.data, want.astype(self.dtype))): return None
        adj = self.Adjacency
        if not (np.array_equal(adj.row, ids[:, 0]) and np.array_equal(adj.col, col) and np.array_equal(adj.data, an.data)): return None
        ng = self.NodeGraph
        if ng.dtype != self.dtype: return None
        g = object.__new__(GraphObject)
        g.dtype = self.dtype
        g.nodes, g.arc_ids, g.arcs, g.targets = self.nodes.copy(), ids.copy(), arcs.copy(), self.targets.copy()
        g.sample_weight = np.asarray(self.sample_weight) * np.ones(self.targets.shape[0])
        g.DIM_NODE_LABEL = np.array(self.nodes.shape[1], ndmin=1, dtype=int)
        g.DIM_ARC_LABEL, g.DIM_TARGET = arcs.shape[1] - 2, self.targets.shape[1]
        g.set_mask, g.output_mask = self.set_mask.copy(), self.output_mask.copy()
        g.aggregation_mode = str(mode)
        g.ArcNode, g.Adjacency, g.NodeGraph = _coo_clone(an), _coo_clone(adj), _coo_clone(ng)
        return g

    def __repr__(self):
        set_mask_type = 'all' if np.all(self.set_mask) else 'mixed'
        return f"graph(n={self.nodes.shape[0]}, a={self.arcs.shape[0]}, " \
               f"ndim={self.DIM_NODE_LABEL}, adim={self.DIM_ARC_LABEL}, tdim={self.DIM_TARGET}, " \
               f"set={set_mask_type}, mode={self.aggregation_mode})"

    __str__ = __repr__

    def setAggregation(self, aggregation_mode: str) -> None:
        self.ArcNode = self.buildArcNode(aggregation_mode)
        self.Adjacency = self.buildAdjacency()
        self.aggregation_mode = aggregation_mode

    # getters return deep copies, as in the reference (graph_class.py:170-197)
    def getArcs(self): return self.arcs.copy()
    def getNodes(self): return self.nodes.copy()
    def getTargets(self): return self.targets.copy()
    def getSetMask(self): return self.set_mask.copy()
    def getOutputMask(self): return self.output_mask.copy()
    def getAdjacency(self): return self.Adjacency.copy()
    def getArcNode(self): return self.ArcNode.copy()
    def getNodeGraph(self): return self.NodeGraph.copy()
    def getSampleWeights(self): return self.sample_weight.copy()

    # ------------------------------------------------------------------------------------------------------------------
    # on-disk formats: same keys / file names as the reference (graph_class.py:200-382) for interchange
    def get_dict_data(self):
        data = {'nodes': self.nodes, 'arcs': self.arcs, 'targets': self.targets}
        if not all(self.set_mask): data['set_mask'] = self.set_mask
        if not all(self.output_mask): data['output_mask'] = self.output_mask
        if np.any(self.sample_weight != 1): data['sample_weight'] = self.sample_weight
        if self.NodeGraph.size > 0 and self.NodeGraph.shape[1] > 1:
            data['NodeGraph'] = np.stack([self.NodeGraph.data, self.NodeGraph.row, self.NodeGraph.col]).transpose()
        return data

    def save(self, graph_npz_path: str, **kwargs) -> None:
        self.save_graph(graph_npz_path, self, False, **kwargs)

    def save_compressed(self, graph_npz_path, **kwargs) -> None:
        self.save_graph(graph_npz_path, self, True, **kwargs)

    def savetxt(self, graph_folder_path: str, format: str = '%.10g', **kwargs) -> None:
        self.save_txt(graph_folder_path, self, format, **kwargs)

    @staticmethod
    def save_graph(graph_npz_path: str, g, compressed: bool = False, **kwargs) -> None:
        (np.savez_compressed if compressed else np.savez)(graph_npz_path, **g.get_dict_data(), **kwargs)

    @staticmethod
    def save_txt(graph_folder_path: str, g, fmt: str = '%.10g', **kwargs) -> None:
        if graph_folder_path[-1] != '/': graph_folder_path += '/'
        if os.path.exists(graph_folder_path): shutil.rmtree(graph_folder_path)
        os.makedirs(graph_folder_path)
        for name, arr in g.get_dict_data().items():
            np.savetxt(f"{graph_folder_path}{name}.txt", np.asarray(arr), fmt=fmt, **kwargs)

    @classmethod
    def _from_dict(cls, data, focus, aggregation_mode):
        data = dict(data)
        if 'NodeGraph' in data:
            ng = np.asarray(data.pop('NodeGraph'))
            data['NodeGraph'] = coo_matrix((ng[:, 0], (ng[:, 1].astype(int), ng[:, 2].astype(int))))
        return cls(focus=focus, aggregation_mode=aggregation_mode, **data)

    @classmethod
    def load(cls, graph_npz_path, focus, aggregation_mode, **kwargs):
        if '.npz' not in graph_npz_path: graph_npz_path += '.npz'
        return cls._from_dict(np.load(graph_npz_path, **kwargs), focus, aggregation_mode)

    @classmethod
    def load_txt(cls, graph_folder_path: str, focus, aggregation_mode, **kwargs):
        if graph_folder_path[-1] != '/': graph_folder_path += '/'
        files = [f for f in os.listdir(graph_folder_path) if f.endswith('.txt')]
        data = {f[:-4]: np.loadtxt(f"{graph_folder_path}{f}", ndmin=2, **kwargs) for f in files}
        for key in ('set_mask', 'output_mask', 'sample_weight'):
            if key in data: data[key] = data[key].reshape(-1)
        return cls._from_dict(data, focus, aggregation_mode)

    # datasets: a folder of 'g<idx>.npz' files / of 'g<idx>/' txt folders (reference graph_class.py:276-302, :358-382).
    # Entries are listed in natural order (g0, g1, g2, ..., g10): the reference iterates os.listdir() as the OS returns it.
    @staticmethod
    def _dataset_entries(folder):
        import re
        key = lambda name: [int(t) if t.isdigit() else t for t in re.split(r'(\d+)', name)]
        return sorted(os.listdir(folder), key=key)

    @staticmethod
    def save_dataset(folder, glist, compressed=False, **kwargs) -> None:
        if folder[-1] != '/': folder += '/'
        if os.path.exists(folder): shutil.rmtree(folder)
        os.makedirs(folder)
        for idx, g in enumerate(glist): type(g).save_graph(f"{folder}g{idx}", g, compressed, **kwargs)

    @staticmethod
    def save_dataset_txt(folder, glist, **kwargs) -> None:
        if folder[-1] != '/': folder += '/'
        if os.path.exists(folder): shutil.rmtree(folder)
        os.makedirs(folder)
        for idx, g in enumerate(glist): type(g).save_txt(f"{folder}g{idx}", g, **kwargs)

    @classmethod
    def load_dataset(cls, folder, focus, aggregation_mode, **kwargs):
        """Folder of npz graph files, as written by `save_dataset`."""
        return [cls.load(os.path.join(folder, g), focus, aggregation_mode, **kwargs) for g in cls._dataset_entries(folder)]

    @classmethod
    def load_dataset_txt(cls, folder, focus, aggregation_mode, **kwargs):
        """Folder of graph folders of txt files, as written by `save_dataset_txt`."""
        return [cls.load_txt(os.path.join(folder, g), focus, aggregation_mode, **kwargs) for g in cls._dataset_entries(folder)]

    # ------------------------------------------------------------------------------------------------------------------
    @staticmethod
    def _merge_arrays(glist, dtype):
        """Block-diagonal concatenation of a list of graphs (reference graph_class.py:394-408); ids are offset in
        int64 (the reference offsets them inside the float32 arcs matrix, `:399`)."""
        nodes_lens = np.array([g.nodes.shape[0] for g in glist], dtype=np.int64)
        offsets = np.concatenate([[0], np.cumsum(nodes_lens)[:-1]])
        arcs = []
        for g, off in zip(glist, offsets):
            a = g.getArcs().astype(np.float64)
            a[:, :2] = g.arc_ids + off
            arcs.append(a)
        arcs = np.concatenate(arcs, axis=0)
        nodes = np.concatenate([g.nodes for g in glist], axis=0, dtype=dtype)
        targets = np.concatenate([g.targets for g in glist], axis=0, dtype=dtype)
        set_mask = np.concatenate([g.set_mask for g in glist], axis=0, dtype=bool)
        output_mask = np.concatenate([g.output_mask for g in glist], axis=0, dtype=bool)
        sample_weight = np.concatenate([g.sample_weight for g in glist], axis=0, dtype=dtype)
        nodegraph = block_diag([g.NodeGraph for g in glist], dtype=dtype)
        return nodes, arcs, targets, set_mask, output_mask, sample_weight, nodegraph

    @classmethod
    def merge(cls, glist: list, focus: str, aggregation_mode: str, dtype='float32'):
        """Merge graphs into one block-diagonal graph; ArcNode / Adjacency are rebuilt for `aggregation_mode` on the
        merged graph, NodeGraph becomes (N, #graphs) (reference graph_class.py:386-413)."""
        nodes, arcs, targets, set_mask, output_mask, sample_weight, nodegraph = cls._merge_arrays(glist, dtype)
        return GraphObject(arcs=arcs, nodes=nodes, targets=targets, focus=focus, set_mask=set_mask,
                           output_mask=output_mask, sample_weight=sample_weight, NodeGraph=nodegraph,
                           aggregation_mode=aggregation_mode)

    @classmethod
    def fromGraphTensor(cls, g, focus: str):
        nodegraph = g.NodeGraph.to_scipy() if focus == 'g' else None
        t = lambda x: x.detach().cpu().numpy()
        return cls(arcs=t(g.arcs), nodes=t(g.nodes), targets=t(g.targets), set_mask=t(g.set_mask),
                   output_mask=t(g.output_mask), sample_weight=t(g.sample_weight), NodeGraph=nodegraph,
                   aggregation_mode=g.aggregation_mode, focus=focus)


class GraphTensor:
    """Device-resident version of a GraphObject (reference graph_class.py:433-560).

    Dense members are torch tensors on `device` (HBM on a GPU box); Adjacency / ArcNode / NodeGraph are
    `SparseMatrix` objects carrying the by-destination CSR that the kernels walk."""

    def __init__(self, nodes, dim_node_label, arcs, targets, set_mask, output_mask, sample_weight,
                 Adjacency, ArcNode, NodeGraph, aggregation_mode, device=None):
        self.dtype = FLOATX
        self.device = torch.device(device) if device is not None else default_device()
        self.aggregation_mode = aggregation_mode
        as_t = lambda x, dt: (x.to(self.device, dt) if isinstance(x, torch.Tensor)
                              else torch.as_tensor(np.ascontiguousarray(x), dtype=dt).to(self.device))

        self.DIM_ARC_LABEL = arcs.shape[1] - 2
        self.DIM_TARGET = targets.shape[1]
        self.DIM_NODE_LABEL = torch.as_tensor(np.array(dim_node_label, ndmin=1), dtype=torch.int32)
        self.nodes = as_t(nodes, torch.float32)
        self.arcs = as_t(arcs, torch.float32)
        self.targets = as_t(targets, torch.float32)
        self.sample_weight = as_t(sample_weight, torch.float32)
        self.set_mask = as_t(set_mask, torch.bool)
        self.output_mask = as_t(output_mask, torch.bool)

        self.Adjacency = SparseMatrix.from_triple(Adjacency)
        self.ArcNode = SparseMatrix.from_triple(ArcNode)
        self.NodeGraph = SparseMatrix.from_triple(NodeGraph)

    def copy(self):
        return GraphTensor(nodes=self.nodes.clone(), dim_node_label=self.DIM_NODE_LABEL.numpy(), arcs=self.arcs.clone(),
                           targets=self.targets.clone(), set_mask=self.set_mask.clone(),
                           output_mask=self.output_mask.clone(), sample_weight=self.sample_weight.clone(),
                           Adjacency=self.Adjacency.copy(), ArcNode=self.ArcNode.copy(),
                           NodeGraph=self.NodeGraph.copy(), aggregation_mode=self.aggregation_mode, device=self.device)

    def __repr__(self):
        set_mask_type = 'all' if bool(torch.all(self.set_mask)) else 'mixed'
        return f"graph_tensor(n={self.nodes.shape[0]}, a={self.arcs.shape[0]}, " \
               f"ndim={self.DIM_NODE_LABEL.tolist()}, adim={self.DIM_ARC_LABEL}, tdim={self.DIM_TARGET}, " \
               f"set={set_mask_type}, mode={self.aggregation_mode}, dtype={self.dtype})"

    __str__ = __repr__

    # npz layout of the reference (graph_class.py:503-535): sparse matrices as [value, row, col] + '<name>_shape'
    def save(self, graph_npz_path, **kwargs) -> None:
        self.save_graph(graph_npz_path, self, False, **kwargs)

    def save_compressed(self, graph_npz_path, **kwargs) -> None:
        self.save_graph(graph_npz_path, self, True, **kwargs)

    @staticmethod
    def _sparse_dict(g):
        sparse_data = {'aggregation_mode': np.array(g.aggregation_mode)}
        for name in ['Adjacency', 'ArcNode', 'NodeGraph']:
            mat = getattr(g, name)
            sparse_data[name] = np.concatenate([mat.values[:, None], mat.indices.astype(np.float32)], axis=1)
            sparse_data[name + '_shape'] = np.array(mat.shape)
        return sparse_data

    @staticmethod
    def save_graph(graph_npz_path: str, g, compressed: bool = False, **kwargs) -> None:
        t = lambda x: x.detach().cpu().numpy()
        (np.savez_compressed if compressed else np.savez)(
            graph_npz_path, dim_node_label=g.DIM_NODE_LABEL.numpy(), nodes=t(g.nodes), arcs=t(g.arcs),
            targets=t(g.targets), sample_weight=t(g.sample_weight), set_mask=t(g.set_mask),
            output_mask=t(g.output_mask), **GraphTensor._sparse_dict(g), **kwargs)

    @classmethod
    def load(cls, graph_npz_path, device=None, **kwargs):
        if '.npz' not in graph_npz_path: graph_npz_path += '.npz'
        data = dict(np.load(graph_npz_path, **kwargs))
        data['aggregation_mode'] = str(data['aggregation_mode'])
        for i in ['Adjacency', 'ArcNode', 'NodeGraph']:
            data[i] = SparseMatrix(data[i][:, 1:].astype(np.int64), data[i][:, 0], data.pop(i + '_shape'))
        return cls(**data, device=device)

    @classmethod
    def fromGraphObject(cls, g: GraphObject, device=None):
        return cls(nodes=g.nodes, dim_node_label=g.DIM_NODE_LABEL, arcs=g.arcs, targets=g.targets,
                   set_mask=g.set_mask, output_mask=g.output_mask, sample_weight=g.sample_weight,
                   NodeGraph=cls.COO2SparseTensor(g.NodeGraph), Adjacency=cls.COO2SparseTensor(g.Adjacency),
                   ArcNode=cls.COO2SparseTensor(g.ArcNode), aggregation_mode=g.aggregation_mode, device=device)

    @staticmethod
    def COO2SparseTensor(coo) -> SparseMatrix:
        """scipy COO -> canonical row-major `SparseMatrix` (reference graph_class.py:551-560)."""
        return SparseMatrix.from_scipy(coo)
