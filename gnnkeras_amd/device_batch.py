"""Batch assembly on the device: `GraphObject.merge` + `GraphTensor.fromGraphObject` (reference `GNN/graph_class.py:386-413`,
`:539-560`, driven by `MultiGraphSequencer.build_batches / on_epoch_end`, `GNN/Sequencers/GraphSequencers.py:42-46, :123-127`)
without the host in the data path.

The reference re-merges every batch in numpy at each `on_epoch_end` (reshuffle) and converts it to tensors; the host port of
that costs ≈1.3 ms per MUTAG batch (merge + upload + CSR build) = 0.18 s per epoch next to 0.05 s of forwards. Here the
DATASET is uploaded once: all graphs concatenated in dataset order, with node / arc ids made graph-local, together with
everything a merged batch needs that is a per-graph property (by-destination CSR pieces of Adjacency and ArcNode,
their by-source forms for the backward pass, 'average' row scales). A merged batch is then the block-diagonal
concatenation of its graphs: each of its arrays is a run of per-graph segments with a per-segment offset added to the ids.
`DeviceDataset.assemble(ids)` builds the table of segment operations with vectorised numpy (≈0.5 k descriptors for 32 graphs),
uploads it with ONE small copy and runs ONE `gnn_ragged_copy` launch; the by-source operands (training) and the COO triples
(API compatibility, tests) are assembled / materialised lazily, only when somebody asks.

What is batch-dependent rather than per-graph: 'normalized' weights (1 / #arcs of the MERGED graph, `graph_class.py:110`)
become one constant row scale per batch; NodeGraph (`:127-138`, block_diag at `:407`) is generated (iota / fills).
Results are identical to the host path — same arrays, same CSR order, same weights — `tests/test_gpu_batch.py` compares
them array by array and through the model."""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _native as nat
from .graph_class import GraphObject
from .sparse import CSRByDestination, SparseMatrix


class DeviceBatch:
    """What `GraphTensor` is to the host path: the merged batch, resident in HBM (same attribute names)."""

    def __init__(self, **kw):
        self.__dict__.update(kw)

    def __repr__(self):
        return f"graph_tensor(n={self.nodes.shape[0]}, a={self.arcs.shape[0]}, ndim={self.DIM_NODE_LABEL.tolist()}, " \
               f"adim={self.DIM_ARC_LABEL}, tdim={self.DIM_TARGET}, mode={self.aggregation_mode}, assembled on {self.device})"


class DeviceDataset:
    """All graphs of a homogeneous dataset, concatenated, on the device; `assemble(ids)` merges any subset in one launch."""

    def __init__(self, graphs, focus: str, aggregation_mode: str, device):
        if aggregation_mode not in ('sum', 'average', 'normalized'):
            raise ValueError("device assembly supports 'sum', 'average' and 'normalized' aggregation")
        if any(not type(g) is GraphObject for g in graphs): raise ValueError('device assembly is built for homogeneous GraphObjects')
        self.focus, self.mode, self.device = focus, aggregation_mode, torch.device(device)
        self.G = len(graphs)
        L = {g.nodes.shape[1] for g in graphs}; A = {g.arcs.shape[1] for g in graphs}; T = {g.targets.shape[1] for g in graphs}
        if len(L) != 1 or len(A) != 1 or len(T) != 1: raise ValueError('graphs of one dataset must share label / target widths')
        self.L, self.W, self.T = L.pop(), A.pop(), T.pop()                      # W = 2 + dim_arc_label
        n = np.array([g.nodes.shape[0] for g in graphs], dtype=np.int64)
        e = np.array([g.arcs.shape[0] for g in graphs], dtype=np.int64)
        t = np.array([g.targets.shape[0] for g in graphs], dtype=np.int64)
        m = np.array([len(g.set_mask) for g in graphs], dtype=np.int64)          # mask length: nodes, or arcs for arc focus
        self.n, self.e, self.t, self.m = n, e, t, m
        off = lambda c: np.concatenate([[0], np.cumsum(c)]).astype(np.int64)
        self.noff, self.eoff, self.toff, self.moff = off(n), off(e), off(t), off(m)
        N, E = int(n.sum()), int(e.sum())
        # the whole dataset as ONE block-diagonal graph (ids global), then made graph-local again where they are ids
        nodes = np.concatenate([g.nodes for g in graphs]).astype(np.float32)
        arcs = np.concatenate([g.arcs for g in graphs]).astype(np.float32)       # local ids (float, like the reference)
        gid_arc = np.repeat(np.arange(self.G), e)
        src = np.concatenate([g.arc_ids[:, 0] for g in graphs]).astype(np.int64) + self.noff[gid_arc]
        dst = np.concatenate([g.arc_ids[:, 1] for g in graphs]).astype(np.int64) + self.noff[gid_arc]
        # by destination: Adjacency (rows = source nodes) and ArcNode (rows = arc ids); arcs are sorted by (src, dst) inside a
        # graph and graphs are contiguous, so a stable sort by destination keeps ascending source inside a destination
        indeg = np.bincount(dst, minlength=N)
        order = np.argsort(dst, kind='stable')
        rowptr_g = np.concatenate([[0], np.cumsum(indeg)])[:-1]                  # global exclusive prefix
        gid_node = np.repeat(np.arange(self.G), n)
        self._up = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(self.device)
        up = self._up
        self.d_nodes, self.d_arcs = up(nodes), up(arcs)
        self.d_targets = up(np.concatenate([g.targets for g in graphs]).astype(np.float32))
        self.d_sw = up(np.concatenate([np.asarray(g.sample_weight, dtype=np.float32).reshape(-1) for g in graphs]))
        self.d_set = up(np.concatenate([g.set_mask for g in graphs]).astype(np.uint8))
        self.d_out = up(np.concatenate([g.output_mask for g in graphs]).astype(np.uint8))
        self.d_rowptr = up((rowptr_g - self.eoff[gid_node]).astype(np.int32))                     # graph-local row pointers
        self.d_adj_src = up((src[order] - self.noff[gid_arc[order]]).astype(np.int32))            # graph-local source ids by destination
        self.d_an_src = up((order - self.eoff[gid_arc[order]]).astype(np.int32))                  # graph-local arc ids by destination
        self.d_scale = None
        if aggregation_mode == 'average':                                        # 1 / in-degree of the destination (graph_class.py:116-121)
            self.d_scale = up(np.where(indeg > 0, 1.0 / np.maximum(indeg, 1), 1.0).astype(np.float32))
        self.d_asrc, self.d_adst = up(src.astype(np.int32) - self.noff[gid_arc].astype(np.int32)), \
            up(dst.astype(np.int32) - self.noff[gid_arc].astype(np.int32))      # arc end points (arc focus: GNN.py:322-325)
        self._host = dict(src=src, dst=dst, indeg=indeg, gid_arc=gid_arc, gid_node=gid_node)
        self._by_source_ready = False
        self.hub = bool(indeg.max(initial=0) > 512)                              # hub rows need the host-side split (sparse.split_heavy)
        # staging for the descriptor tables
        self._pinned, self._copy_done = None, None

    def _prepare_by_source(self):
        """Adjacency by SOURCE (the transposed aggregate of the backward pass): built on first use."""
        if self._by_source_ready: return
        h, up = self._host, self._up
        N = int(self.n.sum())
        outdeg = np.bincount(h['src'], minlength=N)
        order = np.argsort(h['src'], kind='stable')                               # arcs are already (src, dst)-sorted: identity inside a graph
        rowptr = np.concatenate([[0], np.cumsum(outdeg)])[:-1]
        self.d_t_rowptr = up((rowptr - self.eoff[h['gid_node']]).astype(np.int32))
        self.d_t_dst = up((h['dst'][order] - self.noff[h['gid_arc'][order]]).astype(np.int32))
        self.d_t_w = None
        if self.mode == 'average':                                               # weight of arc e = 1 / in-degree(dst_e): not uniform per source
            self.d_t_w = up((1.0 / h['indeg'][h['dst'][order]]).astype(np.float32))
        self._by_source_ready = True

    # ------------------------------------------------------------------------------------------------------------------
    def _run(self, descs):
        """descs: list of (src tensor | None, src element offset, dst tensor, dst element offset, count, kind, iadd, fval, width)
        given as parallel numpy arrays per array family; executes them in one launch."""
        n = sum(len(d['count']) for d in descs)
        if n == 0: return
        tab = np.zeros(n, dtype=[('src', '<u8'), ('dst', '<u8'), ('count', '<i8'), ('kind', '<i4'), ('iadd', '<i4'), ('fval', '<f4'), ('width', '<i4')])
        assert tab.dtype.itemsize == C.sizeof(nat.RaggedDesc)
        pos = 0
        for d in descs:
            k = len(d['count'])
            sl = tab[pos:pos + k]
            esz = d['esize']
            sl['src'] = 0 if d['src'] is None else np.uint64(d['src'].data_ptr()) + np.asarray(d['src_off']).astype(np.uint64) * np.uint64(esz)
            sl['dst'] = np.uint64(d['dst'].data_ptr()) + np.asarray(d['dst_off']).astype(np.uint64) * np.uint64(esz)
            sl['count'], sl['kind'] = d['count'], d['kind']
            sl['iadd'], sl['fval'], sl['width'] = d.get('iadd', 0), d.get('fval', 0.0), d.get('width', 0)
            pos += k
        tab = tab[tab['count'] > 0]
        n = len(tab)
        if n == 0: return
        blocks = (tab['count'] + nat.RC_CHUNK - 1) // nat.RC_CHUNK
        blk = np.concatenate([[0], np.cumsum(blocks)]).astype(np.int32)
        nbytes = tab.nbytes + blk.nbytes
        if self._copy_done is not None: self._copy_done.synchronize()           # the previous table has left the staging buffer
        if self._pinned is None or self._pinned.numel() < nbytes:
            self._pinned = torch.empty(max(nbytes * 2, 1 << 16), dtype=torch.uint8).pin_memory()
        stage = self._pinned[:nbytes].numpy()
        stage[:tab.nbytes] = tab.view(np.uint8).reshape(-1)
        stage[tab.nbytes:] = blk.view(np.uint8)
        dev = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
        dev.copy_(self._pinned[:nbytes], non_blocking=True)
        if self._copy_done is None: self._copy_done = torch.cuda.Event()
        self._copy_done.record(torch.cuda.current_stream(self.device))
        nat.check(nat.lib().gnn_ragged_copy(C.c_void_p(dev.data_ptr()), n, C.c_void_p(dev.data_ptr() + tab.nbytes), int(blk[-1]),
                                            nat.current_stream(self.device)))
        self._keep = dev                              # until the next assemble on this stream (the launch reads it)

    def assemble(self, ids) -> DeviceBatch:
        """The merged batch of graphs `ids` (dataset indices, in batch order)."""
        ids = np.asarray(ids, dtype=np.int64)
        B = len(ids)
        dev = self.device
        n, e, t, m = self.n[ids], self.e[ids], self.t[ids], self.m[ids]
        cum = lambda c: np.concatenate([[0], np.cumsum(c)]).astype(np.int64)
        bn, be, bt, bm = cum(n), cum(e), cum(t), cum(m)                          # offsets inside the batch
        N, E, Tn, Mn = int(bn[-1]), int(be[-1]), int(bt[-1]), int(bm[-1])
        f32 = lambda *s: torch.empty(s, dtype=torch.float32, device=dev)
        i32 = lambda *s: torch.empty(s, dtype=torch.int32, device=dev)
        nodes, arcs, targets, sw = f32(N, self.L), f32(E, self.W), f32(Tn, self.T), f32(Tn)
        set_mask, out_mask = torch.empty(Mn, dtype=torch.uint8, device=dev), torch.empty(Mn, dtype=torch.uint8, device=dev)
        rowptr, adj_src, an_src = i32(N + 1), i32(E), i32(E)
        asrc, adst = (i32(E), i32(E)) if self.focus == 'a' else (None, None)
        scale = f32(N) if self.mode in ('average', 'normalized') else None
        no, eo, to_, mo = self.noff[ids], self.eoff[ids], self.toff[ids], self.moff[ids]
        K = nat
        D = [dict(src=self.d_nodes, src_off=no * self.L, dst=nodes, dst_off=bn[:-1] * self.L, count=n * self.L, kind=K.RC_COPY_F32, esize=4),
             dict(src=self.d_arcs, src_off=eo * self.W, dst=arcs, dst_off=be[:-1] * self.W, count=e * self.W, kind=K.RC_COPY_ROWS_ADD2,
                  fval=bn[:-1].astype(np.float32), width=self.W, esize=4),
             dict(src=self.d_targets, src_off=to_ * self.T, dst=targets, dst_off=bt[:-1] * self.T, count=t * self.T, kind=K.RC_COPY_F32, esize=4),
             dict(src=self.d_sw, src_off=to_, dst=sw, dst_off=bt[:-1], count=t, kind=K.RC_COPY_F32, esize=4),
             dict(src=self.d_set, src_off=mo, dst=set_mask, dst_off=bm[:-1], count=m, kind=K.RC_COPY_U8, esize=1),
             dict(src=self.d_out, src_off=mo, dst=out_mask, dst_off=bm[:-1], count=m, kind=K.RC_COPY_U8, esize=1),
             dict(src=self.d_rowptr, src_off=no, dst=rowptr, dst_off=bn[:-1], count=n, kind=K.RC_COPY_I32_ADD, iadd=be[:-1].astype(np.int32), esize=4),
             dict(src=None, src_off=np.zeros(1, np.int64), dst=rowptr, dst_off=np.array([N]), count=np.array([1]), kind=K.RC_FILL_I32,
                  iadd=np.array([E], np.int32), esize=4),
             dict(src=self.d_adj_src, src_off=eo, dst=adj_src, dst_off=be[:-1], count=e, kind=K.RC_COPY_I32_ADD, iadd=bn[:-1].astype(np.int32), esize=4),
             dict(src=self.d_an_src, src_off=eo, dst=an_src, dst_off=be[:-1], count=e, kind=K.RC_COPY_I32_ADD, iadd=be[:-1].astype(np.int32), esize=4)]
        if self.mode == 'average':
            D.append(dict(src=self.d_scale, src_off=no, dst=scale, dst_off=bn[:-1], count=n, kind=K.RC_COPY_F32, esize=4))
        elif self.mode == 'normalized':                                           # 1 / #arcs of the merged graph (graph_class.py:110)
            D.append(dict(src=None, src_off=np.zeros(1, np.int64), dst=scale, dst_off=np.zeros(1, np.int64), count=np.array([N]),
                          kind=K.RC_FILL_F32, fval=np.array([np.float32(1.0) / np.float32(E)], np.float32), esize=4))
        if self.focus == 'a':
            D += [dict(src=self.d_asrc, src_off=eo, dst=asrc, dst_off=be[:-1], count=e, kind=K.RC_COPY_I32_ADD, iadd=bn[:-1].astype(np.int32), esize=4),
                  dict(src=self.d_adst, src_off=eo, dst=adst, dst_off=be[:-1], count=e, kind=K.RC_COPY_I32_ADD, iadd=bn[:-1].astype(np.int32), esize=4)]
        ng = None
        if self.focus == 'g':                                                     # NodeGraph[n, g] = 1 / |V_g| (graph_class.py:127-138, :407)
            ng_rowptr, ng_src, ng_scale = i32(B + 1), i32(N), f32(B)
            D += [dict(src=None, src_off=np.zeros(B + 1, np.int64), dst=ng_rowptr, dst_off=np.arange(B + 1), count=np.ones(B + 1, np.int64),
                       kind=K.RC_FILL_I32, iadd=bn.astype(np.int32), esize=4),
                  dict(src=None, src_off=np.zeros(1, np.int64), dst=ng_src, dst_off=np.zeros(1, np.int64), count=np.array([N]), kind=K.RC_IOTA_I32,
                       iadd=np.zeros(1, np.int32), esize=4),
                  dict(src=None, src_off=np.zeros(B, np.int64), dst=ng_scale, dst_off=np.arange(B), count=np.ones(B, np.int64), kind=K.RC_FILL_F32,
                       fval=(np.float32(1.0) / n.astype(np.float32)), esize=4)]
        self._run(D)
        csr = dict(rowptr=rowptr, w=None, row_scale=scale, n_dst=N, nnz=E, max_degree=0)
        batch = self
        adj_endpoints = (asrc, adst) if self.focus == 'a' else None

        def adjacency_by_source():
            batch._prepare_by_source()
            t_rowptr, t_dst = i32(N + 1), i32(E)
            t_w = f32(E) if batch.mode == 'average' else None
            t_scale = None
            DD = [dict(src=batch.d_t_rowptr, src_off=no, dst=t_rowptr, dst_off=bn[:-1], count=n, kind=K.RC_COPY_I32_ADD, iadd=be[:-1].astype(np.int32), esize=4),
                  dict(src=None, src_off=np.zeros(1, np.int64), dst=t_rowptr, dst_off=np.array([N]), count=np.array([1]), kind=K.RC_FILL_I32,
                       iadd=np.array([E], np.int32), esize=4),
                  dict(src=batch.d_t_dst, src_off=eo, dst=t_dst, dst_off=be[:-1], count=e, kind=K.RC_COPY_I32_ADD, iadd=bn[:-1].astype(np.int32), esize=4)]
            if batch.mode == 'average':
                DD.append(dict(src=batch.d_t_w, src_off=eo, dst=t_w, dst_off=be[:-1], count=e, kind=K.RC_COPY_F32, esize=4))
            elif batch.mode == 'normalized':
                t_scale = torch.full((N,), float(np.float32(1.0) / np.float32(E)), dtype=torch.float32, device=dev)
            batch._run(DD)
            return dict(rowptr=t_rowptr, src=t_dst, w=t_w, row_scale=t_scale, n_src=N, n_dst=N, nnz=E)

        adjacency = _LazySparse.make((N, N), dict(csr, src=adj_src, n_src=N), dev, endpoints=adj_endpoints, by_source=adjacency_by_source)
        arcnode = _LazySparse.make((E, N), dict(csr, src=an_src, n_src=E), dev,
                              by_source=lambda: dict(rowptr=torch.arange(E + 1, dtype=torch.int32, device=dev), src=adst_of(), w=arc_w(), row_scale=None,
                                                     n_src=N, n_dst=E, nnz=E))

        def adst_of():                       # destination node of every arc, in arc order (ArcNode by source: one entry per arc)
            return arcs[:, 1].to(torch.int32).contiguous()

        def arc_w():                         # its weight: the destination's row scale
            return None if scale is None else scale[arcs[:, 1].long()].contiguous()

        if self.focus == 'g':
            nodegraph = _LazySparse.make((N, B), dict(rowptr=ng_rowptr, src=ng_src, w=None, row_scale=ng_scale, n_src=N, n_dst=B, nnz=N, max_degree=0), dev,
                                    by_source=lambda: dict(rowptr=torch.arange(N + 1, dtype=torch.int32, device=dev),
                                                           src=torch.repeat_interleave(torch.arange(B, dtype=torch.int32, device=dev),
                                                                                       torch.from_numpy(n).to(dev)),
                                                           w=torch.repeat_interleave(ng_scale, torch.from_numpy(n).to(dev)), row_scale=None,
                                                           n_src=B, n_dst=N, nnz=N))
        else:
            nodegraph = SparseMatrix(np.zeros((0, 2), np.int64), np.zeros(0, np.float32), (1, 0))     # reference: empty matrix
        return DeviceBatch(nodes=nodes, arcs=arcs, targets=targets, sample_weight=sw, set_mask=set_mask.view(torch.bool),
                           output_mask=out_mask.view(torch.bool), DIM_NODE_LABEL=torch.tensor([self.L], dtype=torch.int32),
                           DIM_ARC_LABEL=self.W - 2, DIM_TARGET=self.T, Adjacency=adjacency, ArcNode=arcnode, NodeGraph=nodegraph,
                           aggregation_mode=self.mode, device=dev, dtype='float32')


class _LazySparse(SparseMatrix):
    """Device-only `SparseMatrix` whose by-source form is assembled on first request (training only)."""

    @classmethod
    def make(cls, dense_shape, csr, device, endpoints=None, by_source=None):
        m = cls.device_only(dense_shape, csr, device, endpoints=endpoints)
        m._by_source_thunk = by_source
        return m

    def by_source(self, device):
        key = ('by_source', str(torch.device(device)))
        if key not in self._dev:
            self._dev[key] = self._by_source_thunk()
        return self._dev[key]
