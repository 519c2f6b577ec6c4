"""Node-range sharding of the message-passing loop over the GPUs of one node.

One process per GPU, `torch.distributed` over RCCL (backend "nccl" on ROCm). The reference has nothing like this
(single process, single device, eager — SURVEY §2); the algorithm it must reproduce is the same `Loop`
(reference GNN/Models/GNN.py:245-274) on the whole graph.

Layout. With R ranks and chunk = ceil(N / R), rank r owns destination nodes [r·chunk, min(N, (r+1)·chunk)). The
exchanged state lives in two *full* buffers of R slices; a slice is `chunk` state rows followed by ONE flag row whose
first word is that rank's "some node of mine still moves" flag, so the global `reduce_any` of the reference's
`condition` (GNN.py:212) travels inside the same all-gather as the states:

    full buffer = [ slice_0 | slice_1 | ... | slice_{R-1} ],   slice_r = [ chunk rows of SP floats | flag row ]
    padded row of global node g = (g // chunk) * (chunk + 1) + g % chunk

Per iteration rank r runs ONE fused kernel over its destinations (reading any row of the current full buffer, writing
its slice of the other one) and ONE in-place `all_gather_into_tensor`. Every rank also holds the labels of all nodes
in padded-row order (the one-off halo of the neighbour-label aggregate, GNN.py:258); arcs are stored with their
destination, so the ArcNode scatter-add (GNN.py:254) is local. No host synchronisation anywhere in the loop.

Overlap (`overlap=True`): a rank's arcs are split by where their source row lives — *own-range* arcs (sources
the rank itself wrote: final as soon as its own kernel ends) and *halo* arcs (sources received from peers). The exchange
of iteration i is started asynchronously; while it is in flight the rank sums the own-range arcs of iteration i+1
(`gnn_shard_partial`); once the collective has landed, the fused kernel walks only the halo arcs, starting every row's sum
from that partial (`gnn_shard_iteration_split`). Summation order becomes "own-range arcs, then halo arcs" (float32
re-association of the single-GPU result, inside the 1e-5 bar). No host synchronisation is added: the collective's
completion is a stream dependency.

`ShardedLoop` exchanges whole slices with one all-gather. `HaloShardedLoop` exchanges *only the rows a peer actually
reads* (its halo) with one `all_to_all_single` of uneven splits — every pair talks over its own direct xGMI link, and
block-diagonal batches (MUTAG: shard by graph) exchange nothing but the flag rows. On ER graphs, which have no
locality, a peer still needs ~(1 - e^{-deg/R}) of a slice (71 % at R = 8, in-degree 10).
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch
import torch.distributed as dist

from . import _native as nat
from .graph_class import GraphObject
from .sparse import CSRByDestination, HEAVY_THRESHOLD, split_heavy


def partition(n_nodes: int, world_size: int):
    """(chunk, [(lo, hi)] per rank): contiguous node ranges of equal nominal size."""
    chunk = -(-n_nodes // world_size)
    return chunk, [(min(n_nodes, r * chunk), min(n_nodes, (r + 1) * chunk)) for r in range(world_size)]


def padded_row(g, chunk):
    g = np.asarray(g, dtype=np.int64)
    return (g // chunk) * (chunk + 1) + g % chunk


def split_csr(c: CSRByDestination, own_lo: int, own_hi: int):
    """(own, halo): the entries of `c` whose source row lies in [own_lo, own_hi) and the others, both as CSRs over the same
    destinations, entry order inside a row preserved.  `own` carries no row scale (it yields the UN-SCALED partial sum);
    `halo` keeps `c.row_scale`, which the iteration applies once to partial + halo sum."""
    own = (c.src >= own_lo) & (c.src < own_hi)
    row_of = np.repeat(np.arange(c.n_dst), np.diff(c.rowptr.astype(np.int64)))
    out = []
    for sel, scale in ((own, None), (~own, c.row_scale)):
        rowptr = np.zeros(c.n_dst + 1, dtype=np.int64)
        np.cumsum(np.bincount(row_of[sel], minlength=c.n_dst), out=rowptr[1:])
        out.append(CSRByDestination(rowptr.astype(np.int32), np.ascontiguousarray(c.src[sel]),
                                    None if c.w is None else np.ascontiguousarray(c.w[sel]), scale, c.n_src, c.n_dst))
    return out[0], out[1]


class GraphSlice:
    """What ONE rank needs of a graph - nothing that grows with the other ranks' arcs:

        nodes [N, L]                      every node's label (the one-off halo of the neighbour-label aggregate, GNN.py:258)
        arc_src / arc_dst / arc_labels    the arcs whose DESTINATION lies in [lo, hi), in the graph's arc order (sorted by (src, dst))
        values                            their ArcNode / Adjacency entries (reference graph_class.py:105-121: 1, 1 / #arcs of the
                                          whole graph, or 1 / in-degree of the destination - all computable from the slice)
        set_mask / output_mask            of the own nodes
        arc_mask                          arc-focused graphs: set_mask & output_mask of those arcs (the reference's masks run over arcs there)
        nodegraph                         graph-focused graphs: (own node, graph, value) triples of NodeGraph's own rows, and #graphs
    plus, for heterogeneous graphs, the own rows of type_mask and the own columns of every CompositeAdjacency.

    `from_graph` cuts it out of a replicated `GraphObject`; a generator that can produce a rank's arcs directly
    (`synth.er_graph_slice`) never builds the whole graph's matrices at all."""

    def __init__(self, n_nodes, nodes, lo, hi, arc_src, arc_dst, arc_labels, values, set_mask, output_mask, arc_index=None,
                 composite=None, focus='n', arc_mask=None, nodegraph=None):
        self.n_nodes, self.lo, self.hi = int(n_nodes), int(lo), int(hi)
        self.nodes = np.ascontiguousarray(nodes, dtype=np.float32)
        self.arc_src, self.arc_dst = np.asarray(arc_src, dtype=np.int64), np.asarray(arc_dst, dtype=np.int64)
        self.arc_labels = np.ascontiguousarray(arc_labels, dtype=np.float32)
        self.values = np.asarray(values, dtype=np.float32)
        self.set_mask, self.output_mask = np.asarray(set_mask, dtype=bool), np.asarray(output_mask, dtype=bool)
        self.arc_index = None if arc_index is None else np.asarray(arc_index, dtype=np.int64)
        self.composite = composite      # None | dict(type_mask [n_local, T], dim_node_label [T], adjacencies [(row, col, data)] * T)
        self.focus = focus
        self.arc_mask = None if arc_mask is None else np.asarray(arc_mask, dtype=bool)
        self.nodegraph = nodegraph      # None | (row_local i64, graph i64, value f32, n_graphs)
        if focus == 'a' and (self.arc_mask is None or len(self.arc_mask) != len(self.arc_dst)):
            raise ValueError('an arc-focused slice needs one mask entry per local arc')
        if focus == 'g' and nodegraph is None: raise ValueError('a graph-focused slice needs its rows of NodeGraph')
        if len(self.arc_dst) and (self.arc_dst.min() < self.lo or self.arc_dst.max() >= self.hi):
            raise ValueError('a GraphSlice holds the arcs whose destination lies in its own node range')

    @classmethod
    def from_graph(cls, graph: GraphObject, lo: int, hi: int, focus: str = 'n'):
        """`focus`: what the graph was built for ('a': its masks run over arcs; 'g': NodeGraph is pooled over) - a `GraphObject` does
        not remember it (nor does the reference's)."""
        dst = graph.arc_ids[:, 1]
        mine = (dst >= lo) & (dst < hi)
        comp = None
        if hasattr(graph, 'type_mask'):
            cas = []
            for ca in graph.CompositeAdjacencies:
                ca = ca.tocoo()
                keep = (ca.col >= lo) & (ca.col < hi)
                cas.append((ca.row[keep].astype(np.int64), ca.col[keep].astype(np.int64), ca.data[keep].astype(np.float32)))
            comp = dict(type_mask=graph.type_mask[lo:hi], dim_node_label=[int(d) for d in graph.DIM_NODE_LABEL], adjacencies=cas)
        set_mask, output_mask, arc_mask, nodegraph = graph.set_mask[lo:hi], graph.output_mask[lo:hi], None, None
        if focus == 'a':                 # masks run over ARCS (reference GNN.py:317-330): the nodes carry none
            arc_mask = (np.asarray(graph.set_mask, dtype=bool) & np.asarray(graph.output_mask, dtype=bool))[mine]
            set_mask = output_mask = np.ones(hi - lo, dtype=bool)
        elif focus == 'g':
            ng = graph.NodeGraph.tocsr()[lo:hi].tocoo()
            nodegraph = (ng.row.astype(np.int64), ng.col.astype(np.int64), ng.data.astype(np.float32), int(graph.NodeGraph.shape[1]))
        return cls(graph.nodes.shape[0], graph.nodes, lo, hi, graph.arc_ids[mine, 0], dst[mine], graph.arcs[mine][:, 2:],
                   graph.ArcNode.data[mine],                       # aggregation weights computed on the WHOLE graph
                   set_mask, output_mask, arc_index=np.flatnonzero(mine), composite=comp, focus=focus, arc_mask=arc_mask,
                   nodegraph=nodegraph)


class ShardPlan:
    """Host-side (numpy) description of one rank's shard: local CSR operators in padded-row space.  `graph`: the replicated
    `GraphObject`, or just this rank's `GraphSlice`."""

    def __init__(self, graph, rank: int, world_size: int, focus: str = 'n'):
        N = graph.n_nodes if isinstance(graph, GraphSlice) else graph.nodes.shape[0]
        self.N, self.rank, self.world_size = N, rank, world_size
        self.chunk, self.ranges = partition(N, world_size)
        self.lo, self.hi = self.ranges[rank]
        self.n_local = self.hi - self.lo
        self.rows_per_slice = self.chunk + 1
        self.n_rows_full = world_size * self.rows_per_slice
        self.row_base = rank * self.rows_per_slice
        gs = graph if isinstance(graph, GraphSlice) else GraphSlice.from_graph(graph, self.lo, self.hi, focus)
        if (gs.lo, gs.hi) != (self.lo, self.hi):
            raise ValueError(f'rank {rank} of {world_size} owns nodes [{self.lo}, {self.hi}); the slice covers [{gs.lo}, {gs.hi})')
        self.L, self.A = gs.nodes.shape[1], gs.arc_labels.shape[1]
        self.arc_index = gs.arc_index                               # global arc ids of the local (incoming) arcs, when known
        self.e_local = len(gs.arc_dst)
        values = gs.values
        src_rows = padded_row(gs.arc_src, self.chunk)
        dst_local = gs.arc_dst - self.lo
        # COO (rows = sources, cols = local destinations) -> by-destination CSR, ascending source inside a row
        self.adjacency = CSRByDestination.from_coo(src_rows, dst_local, values, (self.n_rows_full, self.n_local))
        self.own_rows = (self.row_base, self.row_base + self.n_local)     # rows of the exchanged buffer this rank writes itself
        self.arcnode = CSRByDestination.from_coo(np.arange(self.e_local), dst_local, values,
                                                 (self.e_local, self.n_local))
        self.arc_labels = gs.arc_labels
        self.nodes_local = np.ascontiguousarray(gs.nodes[self.lo:self.hi])
        nodes_full = np.zeros((self.n_rows_full, self.L), dtype=np.float32)
        nodes_full[padded_row(np.arange(N), self.chunk)] = gs.nodes
        self.nodes_full = nodes_full
        self.out_index = np.flatnonzero(gs.set_mask & gs.output_mask).astype(np.int32)
        self.focus = gs.focus
        if gs.focus == 'g':
            # per-node outputs of EVERY own node (the reference's pooling matmul fails when a node is masked out, GNN.py:345), pooled
            # with the own rows of NodeGraph; the per-graph partial sums of the ranks are added by one all-reduce
            if len(self.out_index) != self.n_local: raise ValueError('graph focus: every node must pass set_mask & output_mask')
            row, col, val, self.n_graphs = gs.nodegraph
            self.nodegraph = CSRByDestination.from_coo(row, col, val, (self.n_local, self.n_graphs))
        elif gs.focus == 'a':
            # arcs live with their destination: the own arcs that pass the arc masks, in the graph's arc order; their end points
            # as rows of the exchanged state buffer (sources may be anybody's) - no per-node outputs at all
            self.out_index = np.zeros(0, dtype=np.int32)
            self.arc_out = np.flatnonzero(gs.arc_mask)
            self.arc_out_src_rows = src_rows[self.arc_out]
            self.arc_out_dst_rows = self.row_base + dst_local[self.arc_out]
            self.arc_out_index = None if gs.arc_index is None else gs.arc_index[self.arc_out]      # global arc ids of the output rows
        self.per_arc_weights = self.adjacency.w is not None
        # heterogeneous graphs: local node ids grouped by type + the per-source-type adjacencies of the label aggregate
        self.composite = gs.composite is not None
        if self.composite:
            tm = gs.composite['type_mask']                          # (n_local, T)
            if not np.all(tm.sum(1) == 1): raise ValueError('type_mask must be one-hot: every node needs exactly one type')
            types = tm.argmax(1)
            order = np.argsort(types, kind='stable')
            self.type_nodes = order.astype(np.int32)
            self.type_offsets = np.concatenate([[0], np.cumsum(np.bincount(types, minlength=tm.shape[1]))]).astype(np.int64)
            self.dim_node_label = list(gs.composite['dim_node_label'])
            self.composite_adjacency = [CSRByDestination.from_coo(padded_row(row, self.chunk), col - self.lo, data,
                                                                  (self.n_rows_full, self.n_local))
                                        for row, col, data in gs.composite['adjacencies']]

    def pad_state(self, state: np.ndarray, SP: int) -> np.ndarray:
        """[N, S] -> full padded buffer [n_rows_full, SP] (flag rows and padding zero)."""
        full = np.zeros((self.n_rows_full, SP), dtype=np.float32)
        full[padded_row(np.arange(self.N), self.chunk), :state.shape[1]] = state
        return full


class ShardedLoop:
    """Forward pass of a GNN on one rank's node range.

        sl = ShardedLoop(model, graph, rank, world_size, device)      # graph replicated on the hosts, or the rank's GraphSlice
        k, state_local, out_local = sl.forward(state0_full)           # collective: every rank calls it

    `state_local` covers the rank's own nodes [lo, hi).  `out_local`: node focus - the masked own nodes; ARC focus (reference
    GNN.py:317-330, CompositeGNN.py:315-327) - the masked arcs whose destination the rank owns, in arc order (`plan.arc_out_index` = their
    global arc ids: the ranks' rows interleave in the whole graph's order); GRAPH focus (GNN.py:341-346) - the pooled [#graphs, T]
    outputs, complete on every rank (per-graph partial sums over the own nodes, one all-reduce)."""

    def __init__(self, model, graph, rank: int, world_size: int, device, group=None, overlap: bool = False):
        """`graph`: the replicated `GraphObject` / `CompositeGraphObject`, or this rank's `GraphSlice` (a rank never needs more)."""
        self.composite = isinstance(model.net_state, (list, tuple))
        self.focus = model._focus
        if self.focus != 'n' and type(self)._layout != 'allgather':
            raise NotImplementedError('arc- / graph-focused sharding runs on the all-gather layout (the compacted halo layout is node-focused)')
        self.model, self.group = model, group
        self.rank, self.world_size = rank, world_size
        self.device = torch.device(device)
        self.overlap = bool(overlap)
        self.plan = p = ShardPlan(graph, rank, world_size, self.focus)
        if p.focus != self.focus: raise ValueError(f"the model is '{self.focus}'-focused, the graph '{p.focus}'-focused")
        if self.composite != p.composite:
            raise ValueError('composite models need CompositeGraphObject graphs (and vice versa)')
        self.n_local, self.e_local, self.per_arc_weights = p.n_local, p.e_local, p.per_arc_weights
        self.L, self.A = p.L, p.A
        self.S = model.state_vect_dim if model.state_vect_dim > 0 else self.L
        self.SP = self._state_ld(self.S)
        self.n_virtual_rows = 0                       # hub segments: virtual state rows behind the exchanged ones (set by _upload)
        self._upload()
        self.buf = [torch.zeros((p.n_rows_full + self.n_virtual_rows, self.SP), dtype=torch.float32, device=self.device) for _ in range(2)]
        self._iter_events = None

    _layout = 'allgather'
    pipeline_chunks = 1               # > 1: the pipelined exchange (set_pipeline); make_sharded_loop(pipeline='auto') arms the measured choice
    _tune_pipeline_pending = False    # ... which the first forward() makes (_tune_pipeline)
    PIPELINE_MARGIN = 0.05            # ... and which leaves the un-chunked exchange only for a count that is at least this much faster

    # ---- device-specific pieces (the gloo/CPU tests override these with numpy stand-ins) ------------------------------------
    def _pool(self, out_nodes):
        """Graph focus: [#graphs, T] partial sums of NodeGraph^T . out over the own nodes."""
        p, dev = self.plan, self.device
        if not hasattr(self, 'd_ng'):
            up = lambda a: None if a is None else torch.from_numpy(np.ascontiguousarray(a)).to(dev)
            c = p.nodegraph
            self.d_ng = dict(rowptr=up(c.rowptr), src=up(c.src), w=up(c.w), row_scale=up(c.row_scale), n_src=c.n_src, n_dst=c.n_dst, nnz=c.nnz)
        T = out_nodes.shape[1]
        pooled = torch.zeros((p.n_graphs, T), dtype=torch.float32, device=dev)
        if p.n_local > 0:
            csr = nat.make_csr(self.d_ng)
            nat.check(nat.lib().gnn_aggregate(C.byref(csr), nat.ptr(out_nodes), T, T, nat.ptr(pooled), T, nat.current_stream(dev)))
        return pooled

    def _arc_outputs(self, k):
        """Arc focus: net_output([state_src | labels_src | state_dst | labels_dst | arc label]) of the own masked arcs (the label
        columns only when the state is not the labels themselves, GNN.py:239-242; heterogeneous models filter on the state alone,
        CompositeGNN.py:315-327), from the exchanged buffer the loop ended on."""
        p, dev, m = self.plan, self.device, self.model
        if not hasattr(self, 'd_arc_rows'):
            self.d_arc_rows = (torch.from_numpy(p.arc_out_src_rows).to(dev), torch.from_numpy(p.arc_out_dst_rows).to(dev),
                               torch.from_numpy(np.ascontiguousarray(p.arc_labels[p.arc_out])).to(dev))
        rs, rd, lab = self.d_arc_rows
        buf = self.buf[int(k) & 1]
        parts = []
        for rows in (rs, rd):
            parts.append(buf[rows, :self.S])
            if m.state_vect_dim > 0 and not self.composite: parts.append(self.d_nodes_full[rows])
        parts.append(lab)
        x = torch.cat(parts, dim=1)
        if x.shape[0] == 0: return torch.zeros((0, m.net_output.units[-1]), dtype=torch.float32, device=dev)
        return m.net_output.to(dev)(x)

    def _finish(self, k, state_local, out_nodes):
        """Node focus: the masked own nodes' outputs as they are; arc / graph focus: see the class docstring."""
        if self.focus == 'g':
            pooled = self._pool(out_nodes)
            if self.world_size > 1: dist.all_reduce(pooled, group=self.group)
            return k, state_local, pooled
        if self.focus == 'a':
            return k, state_local, self._arc_outputs(float(k))
        return k, state_local, out_nodes

    def _state_ld(self, S):
        return int(nat.lib().gnn_state_ld(S))

    def _n_src_rows(self):
        return self.plan.n_rows_full

    def _upload(self):
        p, dev = self.plan, self.device
        up = lambda a: None if a is None else torch.from_numpy(np.ascontiguousarray(a)).to(dev)
        csr = lambda c: dict(rowptr=up(c.rowptr), src=up(c.src), w=up(c.w), row_scale=up(c.row_scale), n_src=c.n_src,
                             n_dst=c.n_dst, nnz=c.nnz)
        self.d_adj, self.d_an = csr(p.adjacency), csr(p.arcnode)
        self.d_nodes, self.d_nodes_full = up(p.nodes_local), up(p.nodes_full)
        self.d_arc_labels, self.d_out_index = up(p.arc_labels), up(p.out_index)
        m = self.model
        a = nat.LoopArgs()
        a.abi_version, a.composite = nat.GNN_ABI_VERSION, 0
        a.n_nodes, a.n_arcs, a.dim_node_label, a.dim_arc_label = p.n_local, p.e_local, self.L, self.A
        a.nodes, a.ld_nodes = nat.ptr(self.d_nodes), self.L
        a.nodes_src, a.ld_nodes_src = nat.ptr(self.d_nodes_full), self.L
        a.arc_labels, a.ld_arcs = nat.ptr(self.d_arc_labels), max(self.A, 1)
        a.adjacency, a.arcnode = nat.make_csr(self.d_adj), nat.make_csr(self.d_an)
        if p.adjacency.max_degree > HEAVY_THRESHOLD:
            # hub rows (sparse.split_heavy, as on one GPU): whole workgroups sum their segments into virtual state rows behind the
            # n_rows_full exchanged ones before every iteration; the iterations walk the light operator
            light, heavy = split_heavy(p.adjacency)
            self.d_adj_light = csr(light)
            self.d_heavy = (up(heavy['seg_beg']), up(heavy['seg_end']))
            a.adjacency_light = nat.make_csr(self.d_adj_light)
            a.heavy_seg_beg, a.heavy_seg_end, a.n_heavy_segments = nat.ptr(self.d_heavy[0]), nat.ptr(self.d_heavy[1]), heavy['n_seg']
            self.n_virtual_rows = heavy['n_seg']
        if self.composite:
            a.composite, a.n_types = 1, len(p.dim_node_label)
            if a.n_types != len(m.net_state): raise ValueError('one state network per node type is required')
            self.d_type_nodes = up(p.type_nodes)
            a.type_nodes = nat.ptr(self.d_type_nodes)
            self.d_ca = [csr(c) for c in p.composite_adjacency]
            for t in range(a.n_types):
                a.type_dim_label[t] = p.dim_node_label[t]
                a.type_offsets[t] = int(p.type_offsets[t])
                a.composite_adjacency[t] = nat.make_csr(self.d_ca[t])
                a.net_state[t] = m.net_state[t].to(dev).native()
            a.type_offsets[a.n_types] = int(p.type_offsets[a.n_types])
        else:
            a.n_types = 1
            a.net_state[0] = m.net_state.to(dev).native()
        a.net_output = m.net_output.to(dev).native()
        a.state_dim, a.max_iteration, a.state_threshold = m.state_vect_dim, m.max_iteration, float(m.state_threshold)
        a.focus = nat.FOCUS['n']
        if self.focus == 'a':
            # the library only runs the loop here (n_out = 0: its output stage has nothing to do); the arc-shaped output network is
            # applied by _arc_outputs.  The focus tells the argument check which input width to expect.
            a.focus = nat.FOCUS['a']
            self.d_arc_ends = torch.zeros(max(p.e_local, 1), dtype=torch.int32, device=dev)
            a.arc_src = a.arc_dst = nat.ptr(self.d_arc_ends)
        a.n_out, a.out_index = len(p.out_index), nat.ptr(self.d_out_index)
        a.flags = m.native_flags
        self.k = torch.zeros((), dtype=torch.float32, device=dev)
        self.state_local = torch.empty((p.n_local, self.S), dtype=torch.float32, device=dev)
        self.out_local = torch.empty((len(p.out_index), m.net_output.units[-1]), dtype=torch.float32, device=dev)
        a.k_out, a.state_out, a.out = nat.ptr(self.k), nat.ptr(self.state_local), nat.ptr(self.out_local)
        # state0 only matters for validation in make_plan: point it at the own rows of buffer 0 later
        nbytes = nat.lib().gnn_loop_workspace_bytes(C.byref(a))
        if nbytes == 0: nat.check(1)
        self._ws = torch.empty(nbytes + 256, dtype=torch.uint8, device=dev)
        aligned = (self._ws.data_ptr() + 255) & ~255
        a.workspace, a.workspace_bytes = C.c_void_p(aligned), nbytes
        self.args = a
        # own-range / halo split of the adjacency (overlap of the exchange with own-range work); only where the library runs
        # this shard on the kernel that can start a row's sum from a partial (else: plain iterations, no overlap)
        if self.overlap and nat.lib().gnn_shard_can_split(C.byref(a)):
            own, halo = split_csr(p.adjacency, *p.own_rows)
            self.d_adj_own, self.d_adj_halo = csr(own), csr(halo)
            self.c_adj_own, self.c_adj_halo = nat.make_csr(self.d_adj_own), nat.make_csr(self.d_adj_halo)
            self.agg_partial = torch.zeros((max(p.n_local, 1), self.SP), dtype=torch.float32, device=dev)
            self.e_own = own.nnz
        else:
            self.overlap = False

    def _setup(self):
        a = self.args
        a.stream = nat.current_stream(self.device)
        a.state0 = nat.ptr(self.buf[0])                      # non-NULL placeholder for the pointer validation
        nat.check(nat.lib().gnn_shard_setup(C.byref(a)))

    def _initial_flags(self):
        """flag of slice r in buffer 0 = predicate(state0 rows of r, ones): every rank evaluates all slices itself, so
        the loop starts without an exchange (GNN.py:261: state_old_0 = ones)."""
        p, lib = self.plan, nat.lib()
        thr = float(self.model.state_threshold)
        for r, (lo, hi) in enumerate(p.ranges):
            base = r * p.rows_per_slice
            rows = self.buf[0][base:base + (hi - lo)]
            flag = self.buf[0][base + p.chunk]               # the slice's flag row; word 0 reinterpreted as int32
            nat.check(lib.gnn_converged(nat.ptr(rows), None, hi - lo, self.S, self.SP, thr, nat.ptr(flag),
                                        nat.current_stream(self.device)))

    def _iteration(self, it: int):
        p = self.plan
        src, dst = self.buf[it & 1], self.buf[(it + 1) & 1]
        gate = src.data_ptr() + 4 * p.chunk * self.SP        # flag word of slice 0
        flag_out = dst.data_ptr() + 4 * (p.row_base + p.chunk) * self.SP
        nat.check(nat.lib().gnn_shard_iteration(C.byref(self.args), nat.ptr(src), nat.ptr(dst), p.row_base,
                                                C.c_void_p(gate), self.world_size, p.rows_per_slice * self.SP,
                                                C.c_void_p(flag_out), it))

    def _gate_args(self, it):
        """(gate pointer, n_gate, stride in floats) of iteration `it`: the flag words of every slice of the buffer it reads."""
        p = self.plan
        return C.c_void_p(self.buf[it & 1].data_ptr() + 4 * p.chunk * self.SP), self.world_size, p.rows_per_slice * self.SP

    def _flag_out(self, it):
        p = self.plan
        return C.c_void_p(self.buf[(it + 1) & 1].data_ptr() + 4 * (p.row_base + p.chunk) * self.SP)

    def _partial(self, it: int):
        """Phase A of iteration `it`: un-scaled sums of the own-range arcs (reads only rows this rank wrote itself)."""
        self.args.stream = nat.current_stream(self.device)
        nat.check(nat.lib().gnn_shard_partial(C.byref(self.args), C.byref(self.c_adj_own), nat.ptr(self.buf[it & 1]),
                                              nat.ptr(self.agg_partial)))

    def _iteration_split(self, it: int):
        """Phase B of iteration `it`: halo arcs on top of the partial sums, dense layer, predicate (after the exchange landed)."""
        gate, n_gate, stride = self._gate_args(it)
        self.args.stream = nat.current_stream(self.device)
        nat.check(nat.lib().gnn_shard_iteration_split(C.byref(self.args), C.byref(self.c_adj_halo), nat.ptr(self.agg_partial),
                                                      nat.ptr(self.buf[it & 1]), nat.ptr(self.buf[(it + 1) & 1]),
                                                      self.plan.row_base, gate, n_gate, stride, self._flag_out(it), it))

    def _output(self):
        nat.check(nat.lib().gnn_shard_output(C.byref(self.args), nat.ptr(self.buf[0]), nat.ptr(self.buf[1]),
                                             self.plan.row_base))
        return self.k, self.state_local, self.out_local

    # ---- pipelined exchange: the halo kernel chunk by chunk, every chunk's rows on their way to the peers while the next is computed ----
    def pipeline_supported(self) -> bool:
        """Chunked launches need the own-range / halo split (the kernel that starts from partial sums), a homogeneous model, a state
        width of at most 64 and no hub rows."""
        return bool(self.overlap) and not self.composite and self.SP <= 64 and self.n_virtual_rows == 0 and type(self)._layout == 'allgather'

    def set_pipeline(self, chunks: int):
        """Cut the slice's rows into `chunks` tile-aligned ranges (the same on every rank: they are ranges of the nominal slice).  With
        more than one chunk the exchange of an iteration is `chunks` point-to-point rounds, each started as soon as its rows are
        written (`forward`), so it runs under the remaining chunks' kernels instead of behind the whole iteration kernel - at the
        price of one kernel ramp-up per chunk.  Which count wins is a property of the machine: `make_sharded_loop(exchange='auto')`
        measures it."""
        chunks = int(chunks)
        if chunks > 1 and not self.pipeline_supported(): chunks = 1
        self.pipeline_chunks = max(1, chunks)
        step = -(-self.plan.chunk // self.pipeline_chunks)
        step = -(-step // 16) * 16                                   # whole 16-node tiles of the iteration kernel
        bounds = list(range(0, self.plan.chunk, step)) + [self.plan.chunk]
        self._chunk_rows = [(bounds[i], bounds[i + 1]) for i in range(len(bounds) - 1)]
        if self.pipeline_chunks > 1: self.transport = 'direct'        # (pieces of a slice travel as point-to-point messages)
        return self.pipeline_chunks

    def _iota(self):
        if not hasattr(self, 'd_iota'): self.d_iota = torch.arange(max(self.n_local, 1), dtype=torch.int32, device=self.device)
        return self.d_iota

    def _iteration_split_rows(self, it: int, lo: int, hi: int, first: bool):
        """Phase B of iteration `it` for the local rows [lo, hi) only (clipped to the rows this rank has)."""
        hi = min(hi, self.n_local); lo = min(lo, hi)
        gate, n_gate, stride = self._gate_args(it)
        self.args.stream = nat.current_stream(self.device)
        ids = self._iota()
        nat.check(nat.lib().gnn_shard_iteration_split_rows(C.byref(self.args), C.byref(self.c_adj_halo), nat.ptr(self.agg_partial),
                                                           nat.ptr(self.buf[it & 1]), nat.ptr(self.buf[(it + 1) & 1]), self.plan.row_base,
                                                           gate, n_gate, stride, self._flag_out(it), it,
                                                           C.c_void_p(ids.data_ptr() + 4 * lo), hi - lo, int(bool(first))))

    def _exchange_rows(self, buf: torch.Tensor, lo: int, hi: int, last: bool):
        """Rows [lo, hi) of every rank's slice of `buf` to everybody (asynchronous; `last`: through the end of the slice - the padding
        rows of the short last rank and the FLAG row, complete once the rank's last chunk has run).  Returns the work handles."""
        if self.world_size == 1: return []
        p = self.plan
        if last: hi = p.rows_per_slice
        return list(dist.batch_isend_irecv(self._p2p_ops(buf, lo, hi)))

    def _p2p_ops(self, buf: torch.Tensor, lo: int, hi: int):
        """The send / receive descriptors of rows [lo, hi) of every slice of `buf`, built once per (buffer, range) and reused by every
        iteration (2 (R - 1) objects and views per exchange otherwise: tens of microseconds of interpreter time per iteration at R = 8,
        where an iteration has ~250 us).  Staggered peer order: rank r starts with r + 1."""
        cache = self.__dict__.setdefault('_p2p_cache', {})
        key = (buf.data_ptr(), lo, hi)
        ops = cache.get(key)
        if ops is None:
            flat = buf.view(-1)
            n = self.plan.rows_per_slice * self.SP
            a, b = lo * self.SP, hi * self.SP
            ops = []
            for off in range(1, self.world_size):
                to, frm = (self.rank + off) % self.world_size, (self.rank - off) % self.world_size
                ops.append(dist.P2POp(dist.isend, flat[self.rank * n + a:self.rank * n + b], to, group=self.group))
                ops.append(dist.P2POp(dist.irecv, flat[frm * n + a:frm * n + b], frm, group=self.group))
            cache[key] = ops
        return ops

    def _pipelined_iteration(self, it: int, partial_next: bool):
        nxt = self.buf[(it + 1) & 1]
        works = []
        for ci, (lo, hi) in enumerate(self._chunk_rows):
            self._iteration_split_rows(it, lo, hi, first=ci == 0)
            works += self._exchange_rows(nxt, lo, hi, last=ci == len(self._chunk_rows) - 1)
        if partial_next: self._partial(it + 1)                          # reads only the rows this rank has just written
        self._exchange_finish(works, nxt, it)

    def _tune_pipeline(self, state0_full, candidates=(1, 2, 4), reps: int = 3):
        """Measured choice of the chunk count (collective): `reps` iterations of each candidate on the real buffers with the gates
        forced open, the slowest rank's time decides, every rank keeps the same count.  Chunk count 1 runs with the transport
        `pick_transport` chose - and stays unless a chunked count is faster by a clear margin (PIPELINE_MARGIN: three wall-clock
        repetitions do not resolve less; the un-chunked exchange is the path every other test and measurement has seen)."""
        import time
        m = self.model
        base_transport = self.transport
        flags, m.native_flags = m.native_flags, m.native_flags | nat.FLAG_NO_EARLY_EXIT
        if hasattr(self, 'args'): self.args.flags = m.native_flags
        times = {}
        try:
            for c in candidates:
                self.transport = base_transport
                if self.set_pipeline(c) != c and c != 1: continue
                self._partial(0)
                def one():
                    if self.pipeline_chunks > 1: self._pipelined_iteration(0, True)
                    else:
                        self._iteration_split(0)
                        w = self._exchange(self.buf[1], 0, async_op=True)
                        self._partial(0)
                        self._exchange_finish(w, self.buf[1], 0)
                one()
                self._sync()
                if self.world_size > 1: dist.barrier(group=self.group)
                t0 = time.perf_counter()
                for _ in range(reps): one()
                self._sync()
                t = torch.tensor([(time.perf_counter() - t0) / reps], dtype=torch.float64, device=self.device)
                if self.world_size > 1: dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
                times[c] = float(t)
        finally:
            m.native_flags = flags
            if hasattr(self, 'args'): self.args.flags = flags
        best = min(times, key=times.get)
        if best != 1 and 1 in times and times[best] > (1.0 - self.PIPELINE_MARGIN) * times[1]: best = 1
        self.transport = base_transport
        self.set_pipeline(best)
        self.pipeline_times = times
        self._prepare(state0_full)                                      # the timed iterations overwrote the buffers
        return best

    def _sync(self):
        if self.device.type == 'cuda': torch.cuda.synchronize(self.device)

    # ---- the loop driven from native code (csrc/shard_loop.hpp) ------------------------------------------------------
    force_exchange = False            # True: issue the collective even at world size 1 (what it costs the HOST can then be measured on one GPU)
    native_loop = False               # True: forward() issues ALL iterations with one `gnn_shard_loop` call (enable_native_loop)
    _comm = None

    def enable_native_loop(self, emulated: bool = False, with_comm: bool = False):
        """Drive the iterations from the library instead of from here: one C call issues every iteration's launches and - over the
        RCCL C API, on an exchange stream of the library's own - its exchange, in the order `forward()` issues them from the
        interpreter (same launches, same bits; ~10 us of host time per iteration instead of the interpreter's 100 .. 200).
        Collective at world size > 1: rank 0 draws the communicator's id, everybody receives it over `self.group` and joins.
        `emulated`: one rank's launches of an R-rank job on one GPU without any exchange (bench.py --emulate-shard: timing, and the
        tests' R-ranks-on-one-device harness); `with_comm`: a communicator even at world size 1 (the exchange then really goes
        through RCCL - an all-gather of the one slice).
        Whole-slice layouts on HIP devices; the interpreter-driven loop stays the default."""
        if self.device.type != 'cuda' or type(self)._layout != 'allgather':
            raise NotImplementedError('the native loop drives whole-slice layouts on HIP devices')
        lib = nat.lib()
        if (self.world_size > 1 or with_comm) and not emulated and self._comm is None:
            uid = torch.zeros(128, dtype=torch.uint8)
            if self.rank == 0:
                raw = (C.c_char * 128)()
                nat.check(lib.gnn_comm_unique_id(C.cast(raw, C.c_void_p)))
                uid = torch.frombuffer(bytearray(raw.raw), dtype=torch.uint8).clone()
            box = [uid.numpy().tobytes()]
            if self.world_size > 1:
                dist.broadcast_object_list(box, src=dist.get_global_rank(self.group, 0) if self.group is not None else 0, group=self.group)
            handle = C.c_void_p(0)
            with torch.cuda.device(self.device):
                nat.check(lib.gnn_comm_create(self.world_size, self.rank, C.c_char_p(box[0]), C.byref(handle)))
            self._comm = handle
        self._emulated = bool(emulated)
        self.native_loop = True
        return self

    def close(self, collective: bool = True):
        """Release the native loop's communicator (collective at world size > 1) and the peer mappings / raw buffers of the peer exchange.
        `collective=False` (the finalizer): local releases only, no barrier."""
        if self._comm is not None:
            nat.lib().gnn_comm_destroy(self._comm)
            self._comm = None
        self.native_loop = False
        if self.peer_exchange:
            torch.cuda.synchronize(self.device)
            if self.world_size > 1 and collective: dist.barrier(group=self.group)      # nobody writes a buffer that is about to go
            for opened in self._peers.values():
                for q in opened: nat.lib().gnn_ipc_close(C.c_void_p(q))
            self._peers, self.peer_exchange = {}, False
            shape = tuple(self.buf[0].shape)
            self.buf = [torch.zeros(shape, dtype=torch.float32, device=self.device) for _ in range(2)]
            self.arrive = None
            for q in self._raw: nat.lib().gnn_device_free(C.c_void_p(q))
            self._raw = []

    def __del__(self):                   # (the native communicator / the peer mappings must not outlive the loop object)
        try:
            if getattr(self, '_comm', None) is not None or getattr(self, 'peer_exchange', False): self.close(collective=False)
        except Exception:
            pass

    def __enter__(self): return self

    def __exit__(self, *exc): self.close()

    def _kernels_only(self):
        """One iteration's launches without the exchange although a communicator exists (profile_iteration's `kernel_s`)."""
        comm, self._comm, em = self._comm, None, getattr(self, '_emulated', False)
        self._emulated = True
        try: self._native_iterations()
        finally: self._comm, self._emulated = comm, em

    def _native_iterations(self, first_iteration: int = 0, n_iterations: int = -1):
        """Iterations [first, first + n) of forward() in one call (after _prepare; n < 0: all): plain, own-range / halo split, or the
        split in chunk launches."""
        p = self.plan
        sa = nat.ShardLoopArgs()
        self.args.stream = nat.current_stream(self.device)
        sa.loop = C.pointer(self.args)
        if self.overlap:
            sa.adjacency_own, sa.adjacency_halo = C.pointer(self.c_adj_own), C.pointer(self.c_adj_halo)
            sa.agg_partial = nat.ptr(self.agg_partial)
        sa.buf[0], sa.buf[1] = self.buf[0].data_ptr(), self.buf[1].data_ptr()
        sa.row_base, sa.rows_per_slice, sa.chunk = p.row_base, p.rows_per_slice, p.chunk
        sa.world_size, sa.rank, sa.SP = self.world_size, self.rank, self.SP
        sa.first_iteration, sa.n_iterations = int(first_iteration), int(n_iterations)
        sa.transport = 1 if self.transport == 'direct' else 0
        keep = None
        if self.overlap and self.pipeline_chunks > 1:
            bounds = [lo for lo, _ in self._chunk_rows] + [self._chunk_rows[-1][1]]
            keep = (C.c_int32 * len(bounds))(*bounds)
            sa.n_chunks, sa.chunk_begin, sa.node_iota = len(bounds) - 1, keep, nat.ptr(self._iota())
        else:
            sa.n_chunks = 1
        sa.emulated = 1 if getattr(self, '_emulated', False) else 0
        sa.comm = self._comm
        nat.check(nat.lib().gnn_shard_loop(C.byref(sa)))

    # ---- backend-independent orchestration ------------------------------------------------------------------------------
    # ---- the exchange inside the iteration kernel: peer stores over IPC-mapped buffers (include/gnnloop.h: gnn_shard_iteration_peers) ----------
    # SURVEY 8e "each rank writes its slice to all 7 peers, 1 hop": the iteration kernel's epilogue stores every new row to the own full buffer
    # and to the peers' full buffers (mapped with hipIpcOpenMemHandle), so the all-gather no longer FOLLOWS the kernel - expected cost per
    # iteration max(kernel, exchange) instead of their sum.  Opt-in (`enable_peer_exchange()`), the RCCL transports stay the default: this
    # path has only ever run between two processes sharing one GPU (tests/test_gpu_peer.py) - no performance claim attaches to it.
    peer_exchange = False

    def enable_peer_exchange(self):
        """Collective over `self.group` (any backend: only picklable handles travel).  Re-allocates the two full state buffers and an
        arrival array as plain device allocations (an IPC handle names an allocation's BASE pointer; the caching allocator hands out
        interior pointers), exchanges their handles and maps every peer's."""
        if type(self)._layout != 'allgather' or self.overlap or self.composite or self.n_virtual_rows or self.per_arc_weights:
            raise NotImplementedError('peer stores: homogeneous all-gather shards without the own-range split, hub rows or per-arc weights')
        if self.SP not in (32, 64): raise NotImplementedError('peer stores: state widths 17 .. 64')
        if not (1 <= self.world_size <= nat.GNN_MAX_PEERS + 1): raise ValueError('peer stores: 1 .. 8 ranks')
        lib = nat.lib()
        self._raw = []

        def raw(shape, dtype):
            n = int(np.prod(shape)) * torch.empty((), dtype=dtype).element_size()
            ptr = C.c_void_p(0)
            nat.check(lib.gnn_device_malloc(C.byref(ptr), n))
            self._raw.append(ptr.value)
            holder = type('RawDeviceBuffer', (), {})()
            holder.__cuda_array_interface__ = {'shape': tuple(shape), 'typestr': '<f4' if dtype == torch.float32 else '<i4', 'data': (ptr.value, False), 'version': 2}
            t = torch.as_tensor(holder, device=self.device)
            t._holder = holder
            return t, ptr.value
        shape = tuple(self.buf[0].shape)
        (b0, p0), (b1, p1) = raw(shape, torch.float32), raw(shape, torch.float32)
        self.buf = [b0, b1]
        self.arrive, pa = raw((max(self.world_size, 8),), torch.int32)
        handles = []
        for ptr_ in (p0, p1, pa):
            h = (C.c_char * 64)()
            nat.check(lib.gnn_ipc_export(C.c_void_p(ptr_), h))
            handles.append(bytes(h))
        every = [None] * self.world_size
        if self.world_size > 1: dist.all_gather_object(every, handles, group=self.group)
        else: every[0] = handles
        self._peers = {}
        for r, hs in enumerate(every):
            if r == self.rank: continue
            opened = []
            for hb in hs:
                q = C.c_void_p(0)
                nat.check(lib.gnn_ipc_open(C.c_char_p(hb), C.byref(q)))
                opened.append(q.value)
            self._peers[r] = opened                       # [buf 0, buf 1, arrive] of rank r in THIS process' address space
        self._arrive_base = 0
        self.peer_exchange = True
        if self.world_size > 1: dist.barrier(group=self.group)          # every rank has mapped every buffer before anyone writes

    def _peer_set(self, which_buf: int):
        ps = nat.PeerSet()
        ps.n_peers = len(self._peers)
        for i, r in enumerate(sorted(self._peers)):
            ps.state_out_full[i] = self._peers[r][which_buf]
            ps.arrive[i] = self._peers[r][2]
        return ps

    def _peer_forward(self):
        """The loop with the exchange in the kernel's epilogue: wait(i) -> iteration i with peer stores -> publish(i + 1)."""
        m, p, lib = self.model, self.plan, nat.lib()
        base, K = self._arrive_base, m.max_iteration
        st = nat.current_stream(self.device)
        self.args.stream = st
        none = nat.PeerSet(); none.n_peers = 0
        sets = [self._peer_set(0), self._peer_set(1)]
        # "ready": this rank's previous forward is over and state_0 / the initial flags are in place (stream order) - peers may write buffer 1
        nat.check(lib.gnn_peer_publish(C.byref(sets[0]), nat.ptr(self.arrive), None, 0, 0, self.rank, base, st))
        for it in range(K):
            nat.check(lib.gnn_peer_wait(nat.ptr(self.arrive), self.world_size, self.rank, base + it, nat.ptr(self.k), st))
            src, dst = self.buf[it & 1], self.buf[(it + 1) & 1]
            gate, n_gate, stride = self._gate_args(it)
            flag_out = self._flag_out(it)
            ps = sets[(it + 1) & 1]
            nat.check(lib.gnn_shard_iteration_peers(C.byref(self.args), nat.ptr(src), nat.ptr(dst), p.row_base, gate, n_gate, stride, flag_out, it, C.byref(ps)))
            nat.check(lib.gnn_peer_publish(C.byref(ps), nat.ptr(self.arrive), flag_out, (p.row_base + p.chunk) * self.SP, self.SP, self.rank, base + it + 1, st))
        nat.check(lib.gnn_peer_wait(nat.ptr(self.arrive), self.world_size, self.rank, base + K, nat.ptr(self.k), st))
        self._arrive_base = base + K + 1

    def _load_state0(self, state0_full):
        p = self.plan
        if isinstance(state0_full, torch.Tensor):
            s0 = state0_full.to(self.device, torch.float32)
            rows = torch.from_numpy(padded_row(np.arange(p.N), p.chunk)).to(self.device)
            self.buf[0].zero_()
            self.buf[0][rows, :s0.shape[1]] = s0                  # (virtual hub rows behind n_rows_full stay zero until the pre-pass)
        else:
            self.buf[0][:p.n_rows_full].copy_(torch.from_numpy(p.pad_state(np.asarray(state0_full, dtype=np.float32), self.SP)))

    transport = 'ring'      # 'ring': RCCL all_gather_into_tensor; 'direct': one send + one receive per peer, all at once

    def _exchange(self, buf: torch.Tensor, it: int = 0, async_op: bool = False):
        """In-place all-gather of the slices of `buf` (states + flag rows): the one exchange step per iteration.
        Two transports for the same data movement: RCCL's all-gather collective ('ring': R - 1 steps, each bound by one xGMI
        link) or R - 1 concurrent point-to-point pairs ('direct': every peer's slice travels over its own link of the fully
        connected xGMI mesh - SURVEY §8e's one-hop all-gather).  `async_op`: returns work handle(s); `_exchange_finish` makes
        the current stream wait for them."""
        if self.world_size == 1 and not self.force_exchange:
            return None
        p = self.plan
        flat = buf.view(-1)
        n = p.rows_per_slice * self.SP
        if self.transport == 'direct':
            ops = self._p2p_ops(buf, 0, p.rows_per_slice)
            if not ops: return None                              # (one rank: nobody to pair with)
            works = dist.batch_isend_irecv(ops)
            if async_op: return works
            for w in works: w.wait()
            return None
        return dist.all_gather_into_tensor(flat, flat[self.rank * n:(self.rank + 1) * n], group=self.group, async_op=async_op)

    def _exchange_finish(self, work, buf: torch.Tensor, it: int = 0):
        if work is None: return
        for w in (work if isinstance(work, (list, tuple)) else [work]): w.wait()     # stream dependencies, not host waits (RCCL)

    def exchange_bytes(self) -> int:
        """Bytes this rank RECEIVES per iteration."""
        return (self.world_size - 1) * self.plan.rows_per_slice * self.SP * 4

    def _prepare(self, state0_full):
        m = self.model
        if m.state_vect_dim > 0:
            if state0_full is None: raise ValueError('state0 (all nodes) is required when state_vect_dim > 0')
            self._load_state0(state0_full)
        else:
            self._load_state0(self.plan_nodes_as_state())
        self._setup()
        self._initial_flags()

    def forward(self, state0_full=None):
        m = self.model
        self._prepare(state0_full)
        if self._tune_pipeline_pending and self.world_size > 1:
            self._tune_pipeline_pending = False
            # chunk rounds are sized messages between every pair of ranks: ALL ranks pipeline, or none does (a rank with hub rows or
            # without the split kernel cannot) - agreed on collectively before anything is timed
            ok = torch.tensor([1 if (self.overlap and self.pipeline_supported()) else 0], dtype=torch.int32, device=self.device)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=self.group)
            if int(ok) == 1: self._tune_pipeline(state0_full)
        if self.native_loop:
            self._native_iterations()
            return self._finish(*self._output())
        if self.peer_exchange:
            self._peer_forward()
            return self._finish(*self._output())
        if not self.overlap:
            for it in range(m.max_iteration):
                self._iteration(it)
                self._exchange(self.buf[(it + 1) & 1], it)
            return self._finish(*self._output())
        # own-range arcs of iteration it+1 are summed while the exchange of iteration it is in flight
        if m.max_iteration > 0: self._partial(0)                        # state_0 is complete on every rank
        for it in range(m.max_iteration):
            if self.pipeline_chunks > 1:
                self._pipelined_iteration(it, it + 1 < m.max_iteration)
                continue
            self._iteration_split(it)
            work = self._exchange(self.buf[(it + 1) & 1], it, async_op=True)
            if it + 1 < m.max_iteration: self._partial(it + 1)          # reads only the rows this rank has just written
            self._exchange_finish(work, self.buf[(it + 1) & 1], it)
        return self._finish(*self._output())

    def plan_nodes_as_state(self):
        """state_vect_dim == 0: state0 = node labels (GNN.py:259)."""
        return np.ascontiguousarray(self._graph_nodes())

    def _graph_nodes(self):
        p = self.plan
        full = p.nodes_full.reshape(self.world_size, p.rows_per_slice, -1)[:, :p.chunk].reshape(-1, p.nodes_full.shape[1])
        return full[:p.N]

    def profile_iteration(self, state0_full=None, reps: int = 10, collective: bool = True) -> dict:
        """Per-iteration device times of this rank, HIP events on the launch stream (collective call: every rank runs it):
        `kernel_s` = the iteration's launches alone (partial + split kernel, or the one fused kernel), no collective;
        `exchange_s` = the collective alone; `iteration_s` = one iteration of `forward()` with both in flight.  Gates are
        forced open for the measurement (the states it leaves behind are meaningless).
        `collective=False`: only `kernel_s`, no process group needed - one rank's share of an R-rank job measured on a single
        GPU (`bench.py --emulate-shard r/R`)."""
        m = self.model
        self._prepare(state0_full)
        flags, m.native_flags = m.native_flags, m.native_flags | nat.FLAG_NO_EARLY_EXIT
        self.args.flags = m.native_flags
        ev = lambda: torch.cuda.Event(enable_timing=True)
        t = {}
        try:
            def timed(fn):
                torch.cuda.synchronize(self.device)
                if self.world_size > 1 and collective: dist.barrier(group=self.group)
                a, b = ev(), ev()
                a.record()
                for _ in range(reps): fn()
                b.record()
                torch.cuda.synchronize(self.device)
                return 1e-3 * a.elapsed_time(b) / reps

            def host_issue(fn):
                """Wall time the HOST needs to issue `fn` once, the device idle when it starts and never waited for: the median of 7
                (at most ~50 iterations' packets are queued at a time - a longer run would fill the HIP queue and measure the device)."""
                import time as _t
                ts = []
                for _ in range(7):
                    torch.cuda.synchronize(self.device)
                    t0 = _t.perf_counter()
                    for _ in range(1 if self.native_loop else 20): fn()
                    ts.append((_t.perf_counter() - t0) / (1 if self.native_loop else 20))
                torch.cuda.synchronize(self.device)
                return float(np.median(ts))

            K_loop = max(int(m.max_iteration), 1)
            per_call = K_loop if self.native_loop else 1           # (the native driver is timed over its whole loop: K iterations per call)

            def kernels():
                if self.native_loop: return self._native_iterations() if not getattr(self, '_comm', None) else self._kernels_only()
                if self.overlap and self.pipeline_chunks > 1:
                    self._partial(0)
                    for ci, (lo, hi) in enumerate(self._chunk_rows): self._iteration_split_rows(0, lo, hi, first=ci == 0)
                elif self.overlap: self._partial(0); self._iteration_split(0)
                else: self._iteration(0)

            def exchange():
                self._exchange_finish(self._exchange(self.buf[1], 0, async_op=True), self.buf[1], 0)

            def both():
                if self.native_loop: return self._native_iterations()
                if self.overlap and self.pipeline_chunks > 1:
                    self._pipelined_iteration(0, True)
                elif self.overlap:
                    self._iteration_split(0)
                    w = self._exchange(self.buf[1], 0, async_op=True)
                    self._partial(0)
                    self._exchange_finish(w, self.buf[1], 0)
                else:
                    self._iteration(0); self._exchange(self.buf[1], 0)

            if self.overlap: self._partial(0)
            if not collective:
                kernels()                                          # warm-up
                t['kernel_s'] = timed(kernels) / per_call
                t['host_issue_s'] = host_issue(kernels) / per_call
                return t
            kernels(); exchange()                                  # warm-up
            t['kernel_s'], t['exchange_s'], t['iteration_s'] = timed(kernels) / per_call, timed(exchange), timed(both) / per_call
            t['host_issue_s'] = host_issue(both) / per_call
        finally:
            m.native_flags = flags
            self.args.flags = flags
        return t


# ----------------------------------------------------------------------------------------------------------------------
# compacted halo exchange
# ----------------------------------------------------------------------------------------------------------------------
class HaloShardPlan(ShardPlan):
    """Shard description whose exchanged buffer holds [own rows | halo rows of peer 0 + flag row | ... | own flag row].

    Source ids of the local CSR operators index that *local view*. `send_rows[p]` = my local rows peer p reads (sorted),
    `halo[q]` = the global ids I read from peer q (sorted). Every rank derives all lists from the replicated host graph,
    so no index exchange is needed."""

    def __init__(self, graph: GraphObject, rank: int, world_size: int):
        N = graph.nodes.shape[0]
        self.N, self.rank, self.world_size = N, rank, world_size
        self.chunk, self.ranges = partition(N, world_size)
        self.lo, self.hi = self.ranges[rank]
        self.n_local = self.hi - self.lo
        src, dst = graph.arc_ids[:, 0], graph.arc_ids[:, 1]
        owner = lambda ids: np.minimum(ids // self.chunk, world_size - 1)
        mine = (dst >= self.lo) & (dst < self.hi)
        self.arc_index = np.flatnonzero(mine)
        self.e_local = int(mine.sum())
        my_src = src[mine]
        src_owner = owner(my_src)
        # halo[q]: sorted unique sources owned by q that my destinations read; view layout
        self.halo, self.seg_start = [], []
        row = self.n_local
        for q in range(world_size):
            h = np.unique(my_src[src_owner == q]) if q != rank else np.zeros(0, dtype=np.int64)
            self.halo.append(h)
            self.seg_start.append(row)
            if q != rank: row += len(h) + 1                         # + the peer's flag row
        self.own_flag_row = row
        self.n_rows_view = row + 1
        self.n_rows_full = self.n_rows_view                         # rows the device operators may read
        self.row_base = 0
        self.recv_rows = [0 if q == rank else len(self.halo[q]) + 1 for q in range(world_size)]
        self.flag_rows = np.array([self.own_flag_row if q == rank else self.seg_start[q] + len(self.halo[q])
                                   for q in range(world_size)], dtype=np.int64)
        # what every peer p reads from me: sources in my range among arcs whose destination p owns
        from_me = (src >= self.lo) & (src < self.hi)
        dst_owner = owner(dst)
        self.send_rows = []
        for p_ in range(world_size):
            if p_ == rank: self.send_rows.append(np.zeros(0, dtype=np.int64)); continue
            self.send_rows.append(np.unique(src[from_me & (dst_owner == p_)]) - self.lo)
        self.send_counts = [0 if p_ == rank else len(self.send_rows[p_]) + 1 for p_ in range(world_size)]
        self.pack_index = np.concatenate([np.concatenate([self.send_rows[p_], [self.own_flag_row]])
                                          for p_ in range(world_size) if p_ != rank] or [np.zeros(0)]).astype(np.int32)
        # local operators in view-row space
        view_row = np.empty(len(my_src), dtype=np.int64)
        own = src_owner == rank
        view_row[own] = my_src[own] - self.lo
        for q in range(world_size):
            sel = src_owner == q
            if q != rank and sel.any():
                view_row[sel] = self.seg_start[q] + np.searchsorted(self.halo[q], my_src[sel])
        values = graph.ArcNode.data[mine]
        dst_local = dst[mine] - self.lo
        self.adjacency = CSRByDestination.from_coo(view_row, dst_local, values, (self.n_rows_view, self.n_local))
        self.own_rows = (0, self.n_local)
        self.arcnode = CSRByDestination.from_coo(np.arange(self.e_local), dst_local, values, (self.e_local, self.n_local))
        self.arc_labels = np.ascontiguousarray(graph.arcs[mine][:, 2:])
        self.nodes_local = np.ascontiguousarray(graph.nodes[self.lo:self.hi])
        # labels of every view row (own + halos): the one-off halo of the label aggregate
        self.view_global = np.full(self.n_rows_view, -1, dtype=np.int64)
        self.view_global[:self.n_local] = np.arange(self.lo, self.hi)
        for q in range(world_size):
            if q != rank: self.view_global[self.seg_start[q]:self.seg_start[q] + len(self.halo[q])] = self.halo[q]
        nodes_full = np.zeros((self.n_rows_view, graph.nodes.shape[1]), dtype=np.float32)
        valid = self.view_global >= 0
        nodes_full[valid] = graph.nodes[self.view_global[valid]]
        self.nodes_full = nodes_full
        sm, om = graph.set_mask[self.lo:self.hi], graph.output_mask[self.lo:self.hi]
        self.out_index = np.flatnonzero(sm & om).astype(np.int32)
        self.per_arc_weights = self.adjacency.w is not None
        self.composite = hasattr(graph, 'type_mask')
        if self.composite:
            tm = graph.type_mask[self.lo:self.hi]
            if not np.all(tm.sum(1) == 1): raise ValueError('type_mask must be one-hot: every node needs exactly one type')
            types = tm.argmax(1)
            self.type_nodes = np.argsort(types, kind='stable').astype(np.int32)
            self.type_offsets = np.concatenate([[0], np.cumsum(np.bincount(types, minlength=tm.shape[1]))]).astype(np.int64)
            self.dim_node_label = [int(d) for d in graph.DIM_NODE_LABEL]
            self.composite_adjacency = []
            lookup = np.full(N, -1, dtype=np.int64)
            lookup[self.view_global[valid]] = np.flatnonzero(valid)
            for ca in graph.CompositeAdjacencies:
                ca = ca.tocoo()
                keep = (ca.col >= self.lo) & (ca.col < self.hi)
                self.composite_adjacency.append(CSRByDestination.from_coo(lookup[ca.row[keep]], ca.col[keep] - self.lo,
                                                                          ca.data[keep], (self.n_rows_view, self.n_local)))

    def view_state(self, state: np.ndarray, SP: int) -> np.ndarray:
        """[N, S] -> this rank's local view buffer [n_rows_view, SP] (own rows + halo rows; flag rows zero)."""
        full = np.zeros((self.n_rows_view, SP), dtype=np.float32)
        valid = self.view_global >= 0
        full[valid, :state.shape[1]] = state[self.view_global[valid]]
        return full


class HaloShardedLoop(ShardedLoop):
    """`ShardedLoop` with the compacted halo exchange: per iteration one fused kernel, one row-gather that packs what
    each peer reads, one `all_to_all_single` (uneven splits, direct pair-wise transfers)."""

    _layout = 'halo'

    def __init__(self, model, graph: GraphObject, rank: int, world_size: int, device, group=None, overlap: bool = False):
        if model._focus != 'n':
            raise NotImplementedError('the compacted halo exchange is built for node-focused models; arc- / graph-focused models '
                                      'shard on the all-gather layout (ShardedLoop)')
        self.focus = 'n'
        self.composite = isinstance(model.net_state, (list, tuple))
        if self.composite != hasattr(graph, 'type_mask'):
            raise ValueError('composite models need CompositeGraphObject graphs (and vice versa)')
        self.model, self.group = model, group
        self.rank, self.world_size = rank, world_size
        self.device = torch.device(device)
        self.overlap = bool(overlap)
        self.plan = p = HaloShardPlan(graph, rank, world_size)
        self._graph_nodes_full = np.ascontiguousarray(graph.nodes, dtype=np.float32)
        self.n_local, self.e_local, self.per_arc_weights = p.n_local, p.e_local, p.per_arc_weights
        self.S = model.state_vect_dim if model.state_vect_dim > 0 else graph.nodes.shape[1]
        self.L, self.A = graph.nodes.shape[1], graph.arcs.shape[1] - 2
        self.SP = self._state_ld(self.S)
        self.n_virtual_rows = 0
        self._upload()
        self.buf = [torch.zeros((p.n_rows_view + self.n_virtual_rows, self.SP), dtype=torch.float32, device=self.device) for _ in range(2)]
        self._make_exchange_state()
        self._iter_events = None

    def _make_exchange_state(self):
        p, dev = self.plan, self.device
        self.d_pack_index = torch.from_numpy(p.pack_index).to(dev)
        self.sendbuf = torch.zeros((max(len(p.pack_index), 1), self.SP), dtype=torch.float32, device=dev)
        self.d_flag_rows = torch.from_numpy(p.flag_rows).to(dev)
        self.gates = [torch.zeros(self.world_size, dtype=torch.int32, device=dev) for _ in range(2)]
        self.in_splits = [c * self.SP for c in p.send_counts]
        self.out_splits = [c * self.SP for c in p.recv_rows]

    # ---- orchestration pieces that differ from the all-gather layout -----------------------------------------------------
    def _load_state0(self, state0_full):
        p = self.plan
        if isinstance(state0_full, torch.Tensor):
            s0 = state0_full.to(self.device, torch.float32)
            vg = torch.from_numpy(p.view_global).to(self.device)
            valid = vg >= 0
            self.buf[0].zero_()
            self.buf[0][:p.n_rows_view][valid, :s0.shape[1]] = s0[vg[valid]]
            self._state0_full = s0
        else:
            arr = np.asarray(state0_full, dtype=np.float32)
            self.buf[0][:p.n_rows_view].copy_(torch.from_numpy(p.view_state(arr, self.SP)))
            self._state0_full = torch.from_numpy(arr).to(self.device)

    def plan_nodes_as_state(self):
        return self._graph_nodes_full

    def _initial_flags(self):
        """The predicate on state_0 covers every node of the graph and every rank holds state_0: evaluate it globally
        once and open / close all R gates together (no exchange before the first iteration)."""
        s0 = self._state0_full.contiguous()
        flag = torch.zeros(1, dtype=torch.int32, device=self.device)
        nat.check(nat.lib().gnn_converged(nat.ptr(s0), None, s0.shape[0], self.S, s0.shape[1], float(self.model.state_threshold),
                                          nat.ptr(flag), nat.current_stream(self.device)))
        self.gates[0].copy_(flag.expand(self.world_size))

    def _iteration(self, it: int):
        p = self.plan
        src, dst = self.buf[it & 1], self.buf[(it + 1) & 1]
        nat.check(nat.lib().gnn_shard_iteration(C.byref(self.args), nat.ptr(src), nat.ptr(dst), 0, nat.ptr(self.gates[it & 1]),
                                                self.world_size, 1, self._flag_out(it), it))

    def _gate_args(self, it):
        return nat.ptr(self.gates[it & 1]), self.world_size, 1

    def _flag_out(self, it):
        return C.c_void_p(self.buf[(it + 1) & 1].data_ptr() + 4 * self.plan.own_flag_row * self.SP)

    def _pack(self, buf):
        n = len(self.plan.pack_index)
        if n:
            nat.check(nat.lib().gnn_gather_rows(nat.ptr(buf), self.SP, nat.ptr(self.d_pack_index), n, self.SP,
                                                nat.ptr(self.sendbuf), self.SP, nat.current_stream(self.device)))

    def _exchange(self, buf: torch.Tensor, it: int = 0, async_op: bool = False):
        """Pack the rows every peer reads (+ my flag row), swap them pair-wise, collect the R flag words for the next gate.
        `async_op`: the swap is only started; `_exchange_finish` waits for it (on the stream) and collects the flags."""
        p = self.plan
        work = None
        if self.world_size > 1:
            self._pack(buf)
            n_recv = sum(p.recv_rows)
            recv = buf[p.n_local:p.n_local + n_recv].view(-1)
            work = dist.all_to_all_single(recv, self.sendbuf[:len(p.pack_index)].view(-1), output_split_sizes=self.out_splits,
                                          input_split_sizes=self.in_splits, group=self.group, async_op=async_op)
        if not async_op: self.gates[(it + 1) & 1].copy_(self._flag_words(buf))
        return work

    def _exchange_finish(self, work, buf: torch.Tensor, it: int = 0):
        if work is not None: work.wait()
        self.gates[(it + 1) & 1].copy_(self._flag_words(buf))

    def exchange_bytes(self) -> int:
        return sum(self.plan.recv_rows) * self.SP * 4

    def _flag_words(self, buf):
        return buf.view(torch.int32)[self.d_flag_rows, 0]

    def _prepare(self, state0_full):
        m = self.model
        if m.state_vect_dim > 0:
            if state0_full is None: raise ValueError('state0 (all nodes) is required when state_vect_dim > 0')
            self._load_state0(state0_full)
        else:
            self._load_state0(self._graph_nodes_full)
        self._setup()
        self._initial_flags()


def choose_exchange(graph: GraphObject, world_size: int) -> str:
    """'halo' when the compacted halos move less than half of the all-gather volume for EVERY rank, else 'allgather'.
    Decided from the whole graph (every rank holds it on the host), never from the caller's own shard: all ranks must
    reach the same answer or they would wait in different collectives."""
    if isinstance(graph, GraphSlice): return 'allgather'           # a rank that only holds its slice cannot know what its peers read
    chunk, ranges = partition(graph.nodes.shape[0], world_size)
    src, dst = graph.arc_ids[:, 0], graph.arc_ids[:, 1]
    worst = 0.0
    for lo, hi in ranges:
        mine = (dst >= lo) & (dst < hi)
        remote = np.unique(src[mine & ((src < lo) | (src >= hi))])
        worst = max(worst, len(remote) / max((world_size - 1) * chunk, 1))
    return 'halo' if worst < 0.5 else 'allgather'


def make_sharded_loop(model, graph: GraphObject, rank: int, world_size: int, device, group=None, exchange: str = 'auto',
                      overlap: bool = True, measure: bool = True, pipeline='off'):
    """`exchange`:
        'allgather' whole slices with RCCL's all-gather collective;
        'direct'    whole slices as R - 1 concurrent point-to-point pairs (one-hop all-gather over the xGMI mesh);
        'halo'      only the rows each peer reads, one all-to-all of uneven splits;
        'peer'      no exchange step at all: the iteration kernel stores every new row into the peers' IPC-mapped buffers
                    (`ShardedLoop.enable_peer_exchange`; homogeneous models, widths 17 .. 64, up to 8 ranks of one node; validated between
                    processes on ONE GPU only - opt-in until a node has measured it against the transports above);
        'auto'      halo when it moves less than half of the all-gather volume for every rank (graphs with locality,
                    block-diagonal batches); otherwise the whole-slice layout, and - with `measure` and more than one rank - the
                    faster of its two transports on THIS machine: both are timed on the real buffers (3 exchanges each, the
                    slowest rank's time decides, so every rank picks the same one).  Which collective wins depends on how RCCL
                    maps it onto the xGMI links, which cannot be known from here: measured, not guessed.
    `pipeline` (whole-slice layouts; ADVICE r4: opt-in until a multi-GPU run has validated it against the un-chunked exchange):
        'off'   one halo-kernel launch and one exchange per iteration (default);
        'auto'  the chunk count (1 / 2 / 4 launches, every chunk's rows on the links while the next is computed) is measured at the
                first forward() on the real buffers and a chunked count kept only when it is > 5 % faster than one launch;
        int     that many chunks (`ShardedLoop.set_pipeline`).
    The environment variable GNN_EXCHANGE_PIPELINE (off / auto / a count) overrides the argument."""
    import os
    pipeline = os.environ.get('GNN_EXCHANGE_PIPELINE', pipeline)
    if isinstance(pipeline, str) and pipeline.isdigit(): pipeline = int(pipeline)
    if not (pipeline in ('off', 'auto') or (isinstance(pipeline, int) and pipeline >= 1)): raise ValueError("pipeline must be 'off', 'auto' or a chunk count")
    if exchange not in ('auto', 'allgather', 'direct', 'halo', 'peer'): raise ValueError('exchange must be auto, allgather, direct, halo or peer')
    if exchange == 'peer':
        sl = ShardedLoop(model, graph, rank, world_size, device, group=group, overlap=False)
        sl.enable_peer_exchange()
        return sl
    pick = exchange
    if exchange == 'auto': pick = choose_exchange(graph, world_size)
    if pick == 'halo':
        if isinstance(graph, GraphSlice): raise ValueError("exchange='halo' derives every peer's pack lists from the whole graph: pass the GraphObject")
        return HaloShardedLoop(model, graph, rank, world_size, device, group=group, overlap=overlap)
    sl = ShardedLoop(model, graph, rank, world_size, device, group=group, overlap=overlap)
    if pick == 'direct': sl.transport = 'direct'
    if exchange == 'auto' and measure and world_size > 1:
        sl.transport, sl.transport_times = pick_transport(sl)
    if pipeline == 'auto' and world_size > 1: sl._tune_pipeline_pending = True      # the chunk count is measured at the first forward()
    elif isinstance(pipeline, int) and pipeline > 1: sl.set_pipeline(pipeline)
    return sl


def pick_transport(sl: 'ShardedLoop', reps: int = 3):
    """Time the two whole-slice transports on `sl`'s own buffers; returns (name of the faster, {name: seconds per exchange}).
    Collective call. The decision uses the MAX over ranks, so all ranks agree."""
    import time
    times = {}
    sync = (lambda: torch.cuda.synchronize(sl.device)) if sl.device.type == 'cuda' else (lambda: None)
    for name in ('ring', 'direct'):
        sl.transport = name
        sl._exchange(sl.buf[1], 0)                                   # warm-up (connection set-up)
        sync(); dist.barrier(group=sl.group)
        t0 = time.perf_counter()
        for _ in range(reps): sl._exchange(sl.buf[1], 0)
        sync()
        t = torch.tensor([(time.perf_counter() - t0) / reps], dtype=torch.float64, device=sl.device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=sl.group)
        times[name] = float(t)
    return min(times, key=times.get), times
