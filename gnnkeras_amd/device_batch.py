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
import weakref

import numpy as np
import torch

from . import _native as nat
from .graph_class import GraphObject
from .sparse import CSRByDestination, SparseMatrix, canonical_device


# (set_mask ptr, output_mask ptr, length) -> (out_index, set_mask, output_mask): the model's `nonzero(set & out)` (GNN.py:269,
# one host synchronisation per new batch) answered from the assembly instead.  The masks are kept alive by the entry, so a
# pointer cannot be recycled while it is registered; entries go when their batch is garbage-collected.
_OUT_INDEX = {}


def lookup_out_index(set_mask, output_mask):
    hit = _OUT_INDEX.get((set_mask.data_ptr(), output_mask.data_ptr(), set_mask.shape[0]))
    # an in-place edit of either mask since the assembly (views share the version counter of their base) makes the entry stale:
    # the caller then derives the index from the masks as they are now
    if hit is None or set_mask._version != hit[3] or output_mask._version != hit[4]: return None
    return hit[0]


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
        # graph focus: the assembled NodeGraph pools ONE graph per data-set entry (a column of 1 / n); an entry that is itself a merge of
        # graphs (its NodeGraph has several columns, reference graph_class.py:407) keeps the host path, which block-diagonalises them
        if focus == 'g' and any(g.NodeGraph.shape[1] != 1 for g in graphs):
            raise ValueError('device assembly pools one graph per data-set entry')
        self.focus, self.mode, self.device = focus, aggregation_mode, canonical_device(device)
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
        both = np.concatenate([np.logical_and(g.set_mask, g.output_mask) for g in graphs])         # rows the output network sees (GNN.py:269)
        gid_mask = np.repeat(np.arange(self.G), m)
        self.oc = np.bincount(gid_mask[both], minlength=self.G).astype(np.int64)
        self.ooff = off(self.oc)
        self.d_oidx = up((np.flatnonzero(both) - self.moff[gid_mask[both]]).astype(np.int32))    # graph-local positions
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
        return self.assemble_many([ids])[0]

    def assemble_many(self, batches) -> list:
        """Every batch of an epoch at once: `batches` = list of lists of dataset indices.  All batches live in epoch-wide arrays
        (a batch is a run of rows of each) filled by ONE descriptor upload and ONE launch; returns the `DeviceBatch` views."""
        nb = len(batches)
        if nb == 0: return []
        sizes = np.array([len(b) for b in batches], dtype=np.int64)
        ids = np.concatenate([np.asarray(b, dtype=np.int64) for b in batches]) if sizes.sum() else np.zeros(0, np.int64)
        bidx = np.repeat(np.arange(nb), sizes)                                   # batch of every graph
        first = np.concatenate([[0], np.cumsum(sizes)])[:-1]                     # position of each batch's first graph
        dev = self.device
        n, e, t, m = self.n[ids], self.e[ids], self.t[ids], self.m[ids]
        excl = lambda c: np.cumsum(c) - c                                        # exclusive prefix over all graphs = offsets in the epoch arrays
        gn, ge, gt, gm = excl(n), excl(e), excl(t), excl(m)
        tot = lambda c: np.add.reduceat(c, first) if len(c) else np.zeros(nb, np.int64)      # per batch totals (every batch has >= 1 graph)
        Nb, Eb, Tb, Mb = tot(n), tot(e), tot(t), tot(m)
        base = lambda g_: g_[first]                                              # epoch offset of each batch
        bN, bE, bT, bM = base(gn), base(ge), base(gt), base(gm)
        ln, le = gn - bN[bidx], ge - bE[bidx]                                    # node / arc offset of every graph INSIDE its batch
        N, E, Tn, Mn = int(n.sum()), int(e.sum()), int(t.sum()), int(m.sum())
        f32 = lambda *s_: torch.empty(s_, dtype=torch.float32, device=dev)
        i32 = lambda *s_: torch.empty(s_, dtype=torch.int32, device=dev)
        nodes, arcs, targets, sw = f32(N, self.L), f32(E, self.W), f32(Tn, self.T), f32(Tn)
        set_mask, out_mask = torch.empty(Mn, dtype=torch.uint8, device=dev), torch.empty(Mn, dtype=torch.uint8, device=dev)
        rowptr, adj_src, an_src = i32(N + nb), i32(E), i32(E)                    # batch b's row pointers start at bN[b] + b
        asrc, adst = (i32(E), i32(E)) if self.focus == 'a' else (None, None)
        scale = f32(N) if self.mode in ('average', 'normalized') else None
        no, eo, to_, mo = self.noff[ids], self.eoff[ids], self.toff[ids], self.moff[ids]
        K = nat
        oc = self.oc[ids]
        go = excl(oc)
        Ob, bO = tot(oc), base(go)
        out_index = i32(int(oc.sum()))
        lm = gm - bM[bidx]                                                        # mask offset of every graph inside its batch
        D = [dict(src=self.d_nodes, src_off=no * self.L, dst=nodes, dst_off=gn * self.L, count=n * self.L, kind=K.RC_COPY_F32, esize=4),
             dict(src=self.d_arcs, src_off=eo * self.W, dst=arcs, dst_off=ge * self.W, count=e * self.W, kind=K.RC_COPY_ROWS_ADD2,
                  fval=ln.astype(np.float32), width=self.W, esize=4),
             dict(src=self.d_targets, src_off=to_ * self.T, dst=targets, dst_off=gt * self.T, count=t * self.T, kind=K.RC_COPY_F32, esize=4),
             dict(src=self.d_sw, src_off=to_, dst=sw, dst_off=gt, count=t, kind=K.RC_COPY_F32, esize=4),
             dict(src=self.d_set, src_off=mo, dst=set_mask, dst_off=gm, count=m, kind=K.RC_COPY_U8, esize=1),
             dict(src=self.d_out, src_off=mo, dst=out_mask, dst_off=gm, count=m, kind=K.RC_COPY_U8, esize=1),
             dict(src=self.d_rowptr, src_off=no, dst=rowptr, dst_off=gn + bidx, count=n, kind=K.RC_COPY_I32_ADD, iadd=le.astype(np.int32), esize=4),
             dict(src=None, src_off=np.zeros(nb, np.int64), dst=rowptr, dst_off=bN + Nb + np.arange(nb), count=np.ones(nb, np.int64),
                  kind=K.RC_FILL_I32, iadd=Eb.astype(np.int32), esize=4),
             dict(src=self.d_adj_src, src_off=eo, dst=adj_src, dst_off=ge, count=e, kind=K.RC_COPY_I32_ADD, iadd=ln.astype(np.int32), esize=4),
             dict(src=self.d_an_src, src_off=eo, dst=an_src, dst_off=ge, count=e, kind=K.RC_COPY_I32_ADD, iadd=le.astype(np.int32), esize=4)]
        D.append(dict(src=self.d_oidx, src_off=self.ooff[ids], dst=out_index, dst_off=go, count=oc, kind=K.RC_COPY_I32_ADD, iadd=lm.astype(np.int32), esize=4))
        if self.mode == 'average':
            D.append(dict(src=self.d_scale, src_off=no, dst=scale, dst_off=gn, count=n, kind=K.RC_COPY_F32, esize=4))
        elif self.mode == 'normalized':                                           # 1 / #arcs of the merged graph (graph_class.py:110)
            D.append(dict(src=None, src_off=np.zeros(nb, np.int64), dst=scale, dst_off=bN, count=Nb, kind=K.RC_FILL_F32,
                          fval=(np.float32(1.0) / Eb.astype(np.float32)), esize=4))
        if self.focus == 'a':
            D += [dict(src=self.d_asrc, src_off=eo, dst=asrc, dst_off=ge, count=e, kind=K.RC_COPY_I32_ADD, iadd=ln.astype(np.int32), esize=4),
                  dict(src=self.d_adst, src_off=eo, dst=adst, dst_off=ge, count=e, kind=K.RC_COPY_I32_ADD, iadd=ln.astype(np.int32), esize=4)]
        if self.focus == 'g':                                                     # NodeGraph[n, g] = 1 / |V_g| (graph_class.py:127-138, :407)
            G = len(ids)
            ng_rowptr, ng_src, ng_scale = i32(G + nb), i32(N), f32(G)
            gpos = np.arange(G) + bidx                                            # slot of graph j's row pointer (batch b's start at first[b] + b)
            D += [dict(src=None, src_off=np.zeros(G, np.int64), dst=ng_rowptr, dst_off=gpos, count=np.ones(G, np.int64), kind=K.RC_FILL_I32,
                       iadd=ln.astype(np.int32), esize=4),
                  dict(src=None, src_off=np.zeros(nb, np.int64), dst=ng_rowptr, dst_off=first + sizes + np.arange(nb), count=np.ones(nb, np.int64),
                       kind=K.RC_FILL_I32, iadd=Nb.astype(np.int32), esize=4),
                  dict(src=None, src_off=np.zeros(nb, np.int64), dst=ng_src, dst_off=bN, count=Nb, kind=K.RC_IOTA_I32, iadd=np.zeros(nb, np.int32), esize=4),
                  dict(src=None, src_off=np.zeros(G, np.int64), dst=ng_scale, dst_off=np.arange(G), count=np.ones(G, np.int64), kind=K.RC_FILL_F32,
                       fval=(np.float32(1.0) / n.astype(np.float32)), esize=4)]
        self._run(D)

        # the by-source operands of the whole epoch, assembled together on the first request (training only)
        epoch = {}

        def by_source_all():
            if 'rowptr' not in epoch:
                self._prepare_by_source()
                t_rowptr, t_dst = i32(N + nb), i32(E)
                t_w = f32(E) if self.mode == 'average' else None
                DD = [dict(src=self.d_t_rowptr, src_off=no, dst=t_rowptr, dst_off=gn + bidx, count=n, kind=K.RC_COPY_I32_ADD, iadd=le.astype(np.int32), esize=4),
                      dict(src=None, src_off=np.zeros(nb, np.int64), dst=t_rowptr, dst_off=bN + Nb + np.arange(nb), count=np.ones(nb, np.int64),
                           kind=K.RC_FILL_I32, iadd=Eb.astype(np.int32), esize=4),
                      dict(src=self.d_t_dst, src_off=eo, dst=t_dst, dst_off=ge, count=e, kind=K.RC_COPY_I32_ADD, iadd=ln.astype(np.int32), esize=4)]
                if self.mode == 'average':
                    DD.append(dict(src=self.d_t_w, src_off=eo, dst=t_w, dst_off=ge, count=e, kind=K.RC_COPY_F32, esize=4))
                self._run(DD)
                epoch.update(rowptr=t_rowptr, dst=t_dst, w=t_w)
            return epoch

        out = []
        dim_node_label = torch.tensor([self.L], dtype=torch.int32)               # one (read-only) tensor for every batch of the epoch
        for b in range(nb):
            n0, n1, e0, e1 = int(bN[b]), int(bN[b] + Nb[b]), int(bE[b]), int(bE[b] + Eb[b])
            t0, t1, m0, m1 = int(bT[b]), int(bT[b] + Tb[b]), int(bM[b]), int(bM[b] + Mb[b])
            Nn, Ee, B = n1 - n0, e1 - e0, int(sizes[b])
            b_scale = None if scale is None else scale[n0:n1]
            b_arcs = arcs[e0:e1]
            csr = dict(rowptr=rowptr[n0 + b:n1 + b + 1], w=None, row_scale=b_scale, n_dst=Nn, nnz=Ee, max_degree=0)

            def adjacency_by_source(b=b, n0=n0, n1=n1, e0=e0, e1=e1, Nn=Nn, Ee=Ee):
                ep = by_source_all()
                sc = None
                if self.mode == 'normalized':
                    sc = torch.full((Nn,), float(np.float32(1.0) / np.float32(Ee)), dtype=torch.float32, device=dev)
                return dict(rowptr=ep['rowptr'][n0 + b:n1 + b + 1], src=ep['dst'][e0:e1], w=None if ep['w'] is None else ep['w'][e0:e1],
                            row_scale=sc, n_src=Nn, n_dst=Nn, nnz=Ee)

            def arcnode_by_source(b_arcs=b_arcs, b_scale=b_scale, Nn=Nn, Ee=Ee):       # one entry per arc: its destination, its weight
                dst_of = b_arcs[:, 1].to(torch.int32).contiguous()
                return dict(rowptr=torch.arange(Ee + 1, dtype=torch.int32, device=dev), src=dst_of,
                            w=None if b_scale is None else b_scale[dst_of.long()].contiguous(), row_scale=None, n_src=Nn, n_dst=Ee, nnz=Ee)

            adjacency = _LazySparse.make((Nn, Nn), dict(csr, src=adj_src[e0:e1], n_src=Nn), dev,
                                         endpoints=(asrc[e0:e1], adst[e0:e1]) if self.focus == 'a' else None, by_source=adjacency_by_source)
            arcnode = _LazySparse.make((Ee, Nn), dict(csr, src=an_src[e0:e1], n_src=Ee), dev, by_source=arcnode_by_source)
            g0_ = int(first[b])
            adjacency._blocks = np.concatenate([[0], np.cumsum(n[g0_:g0_ + B])]).astype(np.int64)      # one diagonal block per graph
            if self.focus == 'g':
                g0 = int(first[b])
                b_ngs = ng_scale[g0:g0 + B]

                def nodegraph_by_source(rp=ng_rowptr[g0 + b:g0 + b + B + 1], b_ngs=b_ngs, Nn=Nn, B=B):
                    # graph of every node = the segment of the batch's (device) row pointers it falls into: no host data, no
                    # synchronisation (repeat_interleave with device repeats waits for its output size: 2 x 50 us per training step)
                    gid = torch.bucketize(torch.arange(Nn, dtype=torch.int32, device=dev), rp[1:], right=True)
                    return dict(rowptr=torch.arange(Nn + 1, dtype=torch.int32, device=dev), src=gid.to(torch.int32),
                                w=b_ngs[gid].contiguous(), row_scale=None, n_src=B, n_dst=Nn, nnz=Nn)

                nodegraph = _LazySparse.make((Nn, B), dict(rowptr=ng_rowptr[g0 + b:g0 + b + B + 1], src=ng_src[n0:n1], w=None, row_scale=b_ngs,
                                                           n_src=Nn, n_dst=B, nnz=Nn, max_degree=0), dev, by_source=nodegraph_by_source)
            else:
                nodegraph = SparseMatrix(np.zeros((0, 2), np.int64), np.zeros(0, np.float32), (1, 0))     # reference: empty matrix
            b_set, b_out = set_mask[m0:m1].view(torch.bool), out_mask[m0:m1].view(torch.bool)
            key = (b_set.data_ptr(), b_out.data_ptr(), m1 - m0)
            _OUT_INDEX[key] = (out_index[int(bO[b]):int(bO[b] + Ob[b])], b_set, b_out, b_set._version, b_out._version)
            out.append(DeviceBatch(nodes=nodes[n0:n1], arcs=b_arcs, targets=targets[t0:t1], sample_weight=sw[t0:t1],
                                   set_mask=b_set, output_mask=b_out,
                                   DIM_NODE_LABEL=dim_node_label, DIM_ARC_LABEL=self.W - 2, DIM_TARGET=self.T,
                                   Adjacency=adjacency, ArcNode=arcnode, NodeGraph=nodegraph, aggregation_mode=self.mode, device=dev,
                                   dtype='float32'))
            weakref.finalize(out[-1], _OUT_INDEX.pop, key, None)
        return out


class _LazySparse(SparseMatrix):
    """Device-only `SparseMatrix` whose by-source form is assembled on first request (training only)."""

    @classmethod
    def make(cls, dense_shape, csr, device, endpoints=None, by_source=None):
        m = cls.device_only(dense_shape, csr, device, endpoints=endpoints)
        m._by_source_thunk = by_source
        return m

    def by_source(self, device):
        key = ('by_source', str(canonical_device(device)))
        if key not in self._dev:
            self._dev[key] = self._by_source_thunk()
        return self._dev[key]
