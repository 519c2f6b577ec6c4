"""N > 1 path on CPU: world_size-2 `gloo` processes run the node-range sharded loop's orchestration (partition,
padded-row layout, local CSR operators, in-place all-gather of state slices with the piggy-backed convergence flags,
gated iterations, k) with the per-rank device step replaced by a NumPy stand-in built from the oracle, and must
reproduce the single-process oracle result. The stand-in is test code; the product path (gnnkeras_amd/distributed.py)
calls libgnnloop.so and has no CPU fallback."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from gnnkeras_amd import GraphObject
from gnnkeras_amd.distributed import ShardedLoop, ShardPlan, HaloShardedLoop, HaloShardPlan, partition, padded_row, choose_exchange, split_csr
from gnnkeras_amd.Models.MLP import MLP, get_inout_dims
from gnnkeras_amd.Models.GNN import GNNnodeBased, GNNarcBased, GNNgraphBased
from gnnkeras_amd.Sequencers.GraphSequencers import MultiGraphSequencer
from gnnkeras_amd.synth import er_graph
from oracle import gnn_oracle as O
from oracle.harness import oracle_loop, rel_err


def csr_to_coo(c):
    dst = np.repeat(np.arange(c.n_dst), np.diff(c.rowptr))
    w = np.ones(c.nnz, np.float32) if c.w is None else c.w
    if c.row_scale is not None: w = w * c.row_scale[dst]
    return np.stack([c.src.astype(np.int64), dst], 1), w, np.array([c.n_src, c.n_dst])


class OracleShardedLoop(ShardedLoop):
    """Device pieces replaced by the NumPy oracle's ops (float64); everything else is the product's orchestration."""
    dtype = np.float64

    def _state_ld(self, S):
        p = 16
        while p < S: p *= 2
        return p

    def _upload(self):
        self.k = torch.zeros((), dtype=torch.float32)

    def _flag(self, buf, row):
        return buf.view(torch.int32)[row, 0]

    def _setup(self):
        p = self.plan
        self.adj, self.an = csr_to_coo(p.adjacency), csr_to_coo(p.arcnode)
        self.agg_arcs = O.sparse_dense_matmul_adjoint(*self.an, p.arc_labels, self.dtype)
        self.agg_nodes = O.sparse_dense_matmul_adjoint(*self.adj, p.nodes_full, self.dtype) \
            if self.model.state_vect_dim > 0 else np.zeros((p.n_local, 0))
        self.k.zero_()

    def _initial_flags(self):
        p, m = self.plan, self.model
        b = self.buf[0].numpy()
        for r, (lo, hi) in enumerate(p.ranges):
            base = r * p.rows_per_slice
            s = b[base:base + hi - lo, :self.S].astype(self.dtype)
            self.buf[0].view(torch.int32)[base + p.chunk, 0] = int(O.condition(0, s, np.ones_like(s), 1, m.state_threshold, self.dtype))

    def _iteration(self, it):
        p, m = self.plan, self.model
        src, dst = self.buf[it & 1], self.buf[(it + 1) & 1]
        dst.view(torch.int32)[p.row_base + p.chunk, 0] = 0
        if not any(int(src.view(torch.int32)[r * p.rows_per_slice + p.chunk, 0]) for r in range(self.world_size)):
            return
        full = src.numpy()[:, :self.S].astype(self.dtype)
        own = full[p.row_base:p.row_base + p.n_local]
        agg = O.sparse_dense_matmul_adjoint(*self.adj, full, self.dtype)
        comps = [own, p.nodes_local, agg, self.agg_nodes, self.agg_arcs] if m.state_vect_dim > 0 else [own, agg, self.agg_arcs]
        new = O.mlp_apply(*m.net_state.spec(), np.concatenate(comps, axis=1), False, self.dtype)
        dst[p.row_base:p.row_base + p.n_local, :self.S] = torch.from_numpy(new.astype(np.float32))
        dst.view(torch.int32)[p.row_base + p.chunk, 0] = int(O.condition(0, new, own, 1, m.state_threshold, self.dtype))
        self.k.fill_(it + 1)

    # overlap path: phase A (own-range arcs, un-scaled) / phase B (halo arcs on top, row scale, dense, predicate)
    def _own_rows_of(self, buf):
        p = self.plan
        return buf.numpy()[p.own_rows[0]:p.own_rows[0] + p.n_local, :self.S].astype(self.dtype)

    def _partial(self, it):
        p = self.plan
        if not hasattr(self, 'adj_own'):
            own, halo = split_csr(p.adjacency, *p.own_rows)
            assert own.nnz + halo.nnz == p.adjacency.nnz and own.row_scale is None
            self.adj_own, self.adj_halo = csr_to_coo(own), csr_to_coo(halo)      # halo carries the row scale, own does not
            self.scale = np.ones(p.n_local) if p.adjacency.row_scale is None else p.adjacency.row_scale.astype(self.dtype)
        full = self.buf[it & 1].numpy()[:, :self.S].astype(self.dtype)
        # only rows this rank wrote itself may be read here: poison everything else to prove it
        lo = p.own_rows[0]
        poisoned = np.full_like(full, np.nan); poisoned[lo:lo + p.n_local] = full[lo:lo + p.n_local]
        part = np.zeros((p.n_local, self.S), self.dtype)
        idx, w, _ = self.adj_own
        np.add.at(part, idx[:, 1], w[:, None] * poisoned[idx[:, 0]])
        assert np.all(np.isfinite(part))
        self.partial = part

    def _iteration_split(self, it):
        p, m = self.plan, self.model
        src, dst = self.buf[it & 1], self.buf[(it + 1) & 1]
        gate_open = self._gate_open(it)
        self._clear_flag(dst)
        if not gate_open: return
        full = src.numpy()[:, :self.S].astype(self.dtype)
        own = self._own_rows_of(src)
        agg = self.partial * self.scale[:, None] + O.sparse_dense_matmul_adjoint(*self.adj_halo, full, self.dtype)
        comps = [own, p.nodes_local, agg, self.agg_nodes, self.agg_arcs] if m.state_vect_dim > 0 else [own, agg, self.agg_arcs]
        new = O.mlp_apply(*m.net_state.spec(), np.concatenate(comps, axis=1), False, self.dtype)
        lo = p.own_rows[0]
        dst[lo:lo + p.n_local, :self.S] = torch.from_numpy(new.astype(np.float32))
        self._set_flag(dst, int(O.condition(0, new, own, 1, m.state_threshold, self.dtype)))
        self.k.fill_(it + 1)

    def _iteration_split_rows(self, it, lo, hi, first):
        """The pipelined exchange's chunk launch: phase B for the local rows [lo, hi) only; the flag row is cleared by the first chunk
        and OR-ed into by every chunk; rows outside [lo, hi) of the destination buffer must not be touched."""
        p, m = self.plan, self.model
        src, dst = self.buf[it & 1], self.buf[(it + 1) & 1]
        hi = min(hi, p.n_local); lo = min(lo, hi)
        if first: self._clear_flag(dst)
        if not self._gate_open(it) or hi == lo: return
        full = src.numpy()[:, :self.S].astype(self.dtype)
        own = self._own_rows_of(src)
        agg = self.partial * self.scale[:, None] + O.sparse_dense_matmul_adjoint(*self.adj_halo, full, self.dtype)
        comps = [own, p.nodes_local, agg, self.agg_nodes, self.agg_arcs] if m.state_vect_dim > 0 else [own, agg, self.agg_arcs]
        new = O.mlp_apply(*m.net_state.spec(), np.concatenate(comps, axis=1)[lo:hi], False, self.dtype)
        r0 = p.own_rows[0]
        dst[r0 + lo:r0 + hi, :self.S] = torch.from_numpy(new.astype(np.float32))
        if int(O.condition(0, new, own[lo:hi], 1, m.state_threshold, self.dtype)): self._set_flag(dst, 1)
        self.k.fill_(it + 1)

    def _gate_open(self, it):
        p, src = self.plan, self.buf[it & 1]
        return any(int(src.view(torch.int32)[r * p.rows_per_slice + p.chunk, 0]) for r in range(self.world_size))

    def _clear_flag(self, dst):
        p = self.plan
        dst.view(torch.int32)[p.row_base + p.chunk, 0] = 0

    def _set_flag(self, dst, v):
        p = self.plan
        dst.view(torch.int32)[p.row_base + p.chunk, 0] = v

    def _output(self):
        p, m = self.plan, self.model
        buf = self.buf[int(self.k) & 1].numpy()
        state = buf[p.row_base:p.row_base + p.n_local, :self.S]
        inp = np.concatenate([state, p.nodes_local], 1) if m.state_vect_dim > 0 else state
        if getattr(self, 'focus', 'n') == 'a':                 # (arc focus: the library's output stage has no rows; _arc_outputs does the work)
            return self.k, torch.from_numpy(state.copy()), torch.zeros((0, m.net_output.units[-1]))
        out = O.mlp_apply(*m.net_output.spec(), inp[p.out_index], False, self.dtype)
        return self.k, torch.from_numpy(state.copy()), torch.from_numpy(out.astype(np.float32))


def _oracle_pool(self, out_nodes):
    p = self.plan
    idx, w, _ = csr_to_coo(p.nodegraph)                       # rows = own nodes, columns = graphs
    pooled = np.zeros((p.n_graphs, out_nodes.shape[1]), self.dtype)
    np.add.at(pooled, idx[:, 1], w[:, None] * out_nodes.numpy().astype(self.dtype)[idx[:, 0]])
    return torch.from_numpy(pooled.astype(np.float32))


def _oracle_arc_outputs(self, k):
    p, m = self.plan, self.model
    buf = self.buf[int(k) & 1].numpy()[:, :self.S]
    parts = []
    for rows in (p.arc_out_src_rows, p.arc_out_dst_rows):
        parts.append(buf[rows])
        if m.state_vect_dim > 0 and not getattr(self, 'composite', False): parts.append(p.nodes_full[rows])
    parts.append(p.arc_labels[p.arc_out])
    x = np.concatenate(parts, axis=1)
    return torch.from_numpy(O.mlp_apply(*m.net_output.spec(), x, False, self.dtype).astype(np.float32))


OracleShardedLoop._pool = _oracle_pool
OracleShardedLoop._arc_outputs = _oracle_arc_outputs


class OracleHaloShardedLoop(HaloShardedLoop):
    """Same stand-in for the compacted halo exchange: device pieces from the oracle, plan / pack lists / all_to_all real."""
    dtype = np.float64
    _state_ld = OracleShardedLoop._state_ld
    _upload = OracleShardedLoop._upload
    _setup = OracleShardedLoop._setup

    def _initial_flags(self):
        s = self._state0_full.numpy()[:, :self.S].astype(self.dtype)
        self.gates[0].fill_(int(O.condition(0, s, np.ones_like(s), 1, self.model.state_threshold, self.dtype)))

    def _iteration(self, it):
        p, m = self.plan, self.model
        src, dst = self.buf[it & 1], self.buf[(it + 1) & 1]
        dst.view(torch.int32)[p.own_flag_row, 0] = 0
        if not bool(self.gates[it & 1].any()): return
        full = src.numpy()[:, :self.S].astype(self.dtype)
        own = full[:p.n_local]
        agg = O.sparse_dense_matmul_adjoint(*self.adj, full, self.dtype)
        comps = [own, p.nodes_local, agg, self.agg_nodes, self.agg_arcs] if m.state_vect_dim > 0 else [own, agg, self.agg_arcs]
        new = O.mlp_apply(*m.net_state.spec(), np.concatenate(comps, axis=1), False, self.dtype)
        dst[:p.n_local, :self.S] = torch.from_numpy(new.astype(np.float32))
        dst.view(torch.int32)[p.own_flag_row, 0] = int(O.condition(0, new, own, 1, m.state_threshold, self.dtype))
        self.k.fill_(it + 1)

    _own_rows_of = OracleShardedLoop._own_rows_of
    _partial = OracleShardedLoop._partial
    _iteration_split = OracleShardedLoop._iteration_split

    def _gate_open(self, it):
        return bool(self.gates[it & 1].any())

    def _clear_flag(self, dst):
        dst.view(torch.int32)[self.plan.own_flag_row, 0] = 0

    def _set_flag(self, dst, v):
        dst.view(torch.int32)[self.plan.own_flag_row, 0] = v

    def _pack(self, buf):
        n = len(self.plan.pack_index)
        if n: self.sendbuf[:n].copy_(buf[self.d_pack_index.long()])

    def _output(self):
        p, m = self.plan, self.model
        state = self.buf[int(self.k) & 1].numpy()[:p.n_local, :self.S]
        inp = np.concatenate([state, p.nodes_local], 1) if m.state_vect_dim > 0 else state
        out = O.mlp_apply(*m.net_output.spec(), inp[p.out_index], False, self.dtype)
        return self.k, torch.from_numpy(state.copy()), torch.from_numpy(out.astype(np.float32))


def _problem(threshold, d=6, max_it=12):
    rng = np.random.default_rng(0)
    g = er_graph(203, 1500, dim_node_label=5, dim_arc_label=2, seed=7)          # 203 is not a multiple of 2 or 3
    om = rng.random(203) < 0.7
    g = GraphObject(g.nodes, g.arcs, rng.normal(size=(int(om.sum()), 2)), focus='n', set_mask=rng.random(203) < 0.8,
                    output_mask=om, aggregation_mode='average')
    inp, lay = get_inout_dims('state', 5, 2, 2, 'n', d)
    ns = MLP(inp[0], lay, 'tanh', 'lecun_normal', 'lecun_normal', rng=0, device='cpu')
    ns.set_weights([w * 0.4 if w.ndim == 2 else w for w in ns.get_weights()])
    inp, lay = get_inout_dims('output', 5, 2, 2, 'n', d)
    no = MLP(inp[0], lay, 'softmax', 'glorot_normal', 'glorot_normal', rng=1, device='cpu')
    model = GNNnodeBased(ns, no, d, max_it, threshold)
    s0 = rng.normal(0, 0.1, (203, d)).astype(np.float32) if d else None
    return g, model, s0


def _worker(rank, world, port, threshold, d, out_q, halo=False, overlap=False, transport=None, from_slice=False, chunks=1):
    os.environ['MASTER_ADDR'], os.environ['MASTER_PORT'] = '127.0.0.1', str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        g, model, s0 = _problem(threshold, d)
        if from_slice:                      # the rank holds nothing but its own destination range of the graph
            from gnnkeras_amd.distributed import GraphSlice
            g = GraphSlice.from_graph(g, *partition(g.nodes.shape[0], world)[1][rank])
        sl = (OracleHaloShardedLoop if halo else OracleShardedLoop)(model, g, rank, world, 'cpu', overlap=overlap)
        assert sl.overlap == overlap
        if transport == 'measured':
            from gnnkeras_amd.distributed import pick_transport
            sl.transport, times = pick_transport(sl, reps=2)
            assert set(times) == {'ring', 'direct'} and sl.transport in times
            names = [None] * world
            dist.all_gather_object(names, sl.transport)
            assert len(set(names)) == 1                              # every rank picked the same transport
        elif transport:
            sl.transport = transport
        if chunks == 'tuned':                   # the measured choice of the chunk count (collective), as make_sharded_loop(pipeline='auto') arms it
            sl._tune_pipeline_pending = True
        elif chunks > 1:
            assert sl.pipeline_supported() and sl.set_pipeline(chunks) == chunks and sl.transport == 'direct'
            assert len(sl._chunk_rows) >= 2 and sl._chunk_rows[0][0] == 0 and sl._chunk_rows[-1][1] == sl.plan.chunk
        k, state, out = sl.forward(s0)
        if chunks == 'tuned':
            assert set(sl.pipeline_times) == {1, 2, 4} and sl.pipeline_chunks in (1, 2, 4)
            counts = [None] * world
            dist.all_gather_object(counts, sl.pipeline_chunks)
            assert len(set(counts)) == 1                                 # every rank kept the same chunk count
        out_q.put((rank, float(k), state.numpy(), out.numpy(), sl.plan.lo, sl.plan.hi))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('overlap', [False, True])
@pytest.mark.parametrize('halo', [False, True])
@pytest.mark.parametrize('world', [2, 3])
@pytest.mark.parametrize('threshold,d', [(0.0, 6), (0.02, 6), (0.0, 0)])
def test_sharded_loop_matches_single_process_oracle(world, threshold, d, halo, overlap):
    g, model, s0 = _problem(threshold, d)
    seq = MultiGraphSequencer([g], 'n', 'average', 1, shuffle=False, device='cpu')
    k_ref, st_ref, out_ref = oracle_loop(model, seq[0][0], s0, np.float64)
    if threshold > 0: assert 1 < k_ref < model.max_iteration              # early exit really happens
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 29500 + (os.getpid() + world * 7 + int(threshold * 100) + d + 13 * halo + 29 * overlap) % 1000
    procs = [ctx.Process(target=_worker, args=(r, world, port, threshold, d, q, halo, overlap)) for r in range(world)]
    for p in procs: p.start()
    res = sorted([q.get(timeout=180) for _ in procs])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert [r[0] for r in res] == list(range(world))
    assert all(r[1] == float(k_ref) for r in res)
    state = np.concatenate([r[2] for r in res])
    out = np.concatenate([r[3] for r in res])
    assert state.shape == st_ref.shape and out.shape == out_ref.shape
    assert rel_err(state, st_ref) < 1e-6 and rel_err(out, out_ref) < 1e-6


@pytest.mark.parametrize('world,chunks,threshold', [(2, 2, 0.0), (3, 4, 0.02), (2, 'tuned', 0.02), (3, 3, 0.0)])
def test_pipelined_exchange_matches_single_process_oracle(world, chunks, threshold):
    """The exchange of an iteration in several point-to-point rounds, each started as soon as its chunk of rows is written
    (`ShardedLoop.set_pipeline`): real collectives over gloo, the chunk launches replaced by a NumPy stand-in that
    computes exactly rows [lo, hi); ragged last rank (203 nodes over 2 / 3 ranks), early exit through the flag row that travels with
    the LAST chunk, and the measured choice of the chunk count."""
    d = 6
    g, model, s0 = _problem(threshold, d)
    seq = MultiGraphSequencer([g], 'n', 'average', 1, shuffle=False, device='cpu')
    k_ref, st_ref, out_ref = oracle_loop(model, seq[0][0], s0, np.float64)
    if threshold > 0: assert 1 < k_ref < model.max_iteration
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 29500 + (os.getpid() + 601 + world * 17 + (7 if chunks == 'tuned' else chunks) * 3) % 1000
    procs = [ctx.Process(target=_worker, args=(r, world, port, threshold, d, q, False, True, None, False, chunks)) for r in range(world)]
    for p in procs: p.start()
    res = sorted([q.get(timeout=180) for _ in procs])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert all(r[1] == float(k_ref) for r in res)
    state = np.concatenate([r[2] for r in res]); out = np.concatenate([r[3] for r in res])
    assert rel_err(state, st_ref) < 1e-6 and rel_err(out, out_ref) < 1e-6


@pytest.mark.parametrize('overlap', [False, True])
def test_sharded_loop_from_per_rank_graph_slices(overlap):
    """Every rank builds its shard from its own `GraphSlice` (arcs of its destination range + all labels) instead of the
    replicated `GraphObject`: same result."""
    world, threshold, d = 2, 0.02, 6
    g, model, s0 = _problem(threshold, d)
    seq = MultiGraphSequencer([g], 'n', 'average', 1, shuffle=False, device='cpu')
    k_ref, st_ref, out_ref = oracle_loop(model, seq[0][0], s0, np.float64)
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 29500 + (os.getpid() + 977 + 31 * overlap) % 1000
    procs = [ctx.Process(target=_worker, args=(r, world, port, threshold, d, q, False, overlap, None, True)) for r in range(world)]
    for p in procs: p.start()
    res = sorted([q.get(timeout=180) for _ in procs])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert all(r[1] == float(k_ref) for r in res)
    assert rel_err(np.concatenate([r[2] for r in res]), st_ref) < 1e-6 and rel_err(np.concatenate([r[3] for r in res]), out_ref) < 1e-6


def test_shard_plan_from_a_slice_equals_the_plan_from_the_whole_graph():
    """`GraphSlice.from_graph` and the generator-side `synth.er_graph_slice` (which never builds the whole graph's matrices)
    against `ShardPlan(GraphObject)`: bit-identical operators, labels and index lists, every aggregation mode, ragged ranges."""
    from gnnkeras_amd.distributed import GraphSlice
    from gnnkeras_amd.synth import er_graph_slice
    for mode in ('average', 'sum', 'normalized'):
        g = er_graph(1003, 9000, aggregation_mode=mode, seed=5)
        for R in (1, 3, 8):
            for r, (lo, hi) in enumerate(partition(1003, R)[1]):
                a = ShardPlan(g, r, R)
                for b in (ShardPlan(GraphSlice.from_graph(g, lo, hi), r, R),
                          ShardPlan(er_graph_slice(1003, 9000, lo, hi, aggregation_mode=mode, seed=5), r, R)):
                    for op in ('adjacency', 'arcnode'):
                        for name in ('rowptr', 'src', 'w', 'row_scale'):
                            x, y = getattr(getattr(a, op), name), getattr(getattr(b, op), name)
                            assert (x is None and y is None) or np.array_equal(x, y), (mode, R, r, op, name)
                    for name in ('arc_labels', 'nodes_full', 'nodes_local', 'out_index', 'arc_index'):
                        assert np.array_equal(getattr(a, name), getattr(b, name)), name
                    assert (a.e_local, a.n_local, a.L, a.A, a.per_arc_weights) == (b.e_local, b.n_local, b.L, b.A, b.per_arc_weights)
    with pytest.raises(ValueError):
        ShardPlan(GraphSlice.from_graph(g, 0, 100), 1, 8)              # a slice of another rank's range


def test_composite_shard_plan_from_a_generated_slice_equals_the_plan_from_the_whole_graph():
    """BASELINE C5 on shards: `synth.er_composite_graph_slice` (a rank generates only its own destination range of the heterogeneous
    graph) against `ShardPlan(CompositeGraphObject)` and `GraphSlice.from_graph`: bit-identical operators, per-type node lists and
    per-source-type adjacencies, every aggregation mode incl. 'composite_average' (reference composite_graph_class.py:57-103)."""
    from gnnkeras_amd.distributed import GraphSlice
    from gnnkeras_amd.synth import er_composite_graph, er_composite_graph_slice
    dims = (5, 3, 2)
    for mode in ('average', 'composite_average', 'sum', 'normalized'):
        g = er_composite_graph(1003, 9000, dim_node_label=dims, aggregation_mode=mode, seed=5)
        for R in (1, 4):
            for r, (lo, hi) in enumerate(partition(1003, R)[1]):
                a = ShardPlan(g, r, R)
                assert a.composite
                for b in (ShardPlan(GraphSlice.from_graph(g, lo, hi), r, R),
                          ShardPlan(er_composite_graph_slice(1003, 9000, lo, hi, dim_node_label=dims, aggregation_mode=mode, seed=5), r, R)):
                    ops = [(a.adjacency, b.adjacency), (a.arcnode, b.arcnode)] + list(zip(a.composite_adjacency, b.composite_adjacency))
                    assert len(a.composite_adjacency) == len(b.composite_adjacency) == len(dims)
                    for x_op, y_op in ops:
                        for name in ('rowptr', 'src', 'w', 'row_scale'):
                            x, y = getattr(x_op, name), getattr(y_op, name)
                            assert (x is None and y is None) or np.array_equal(x, y), (mode, R, r, name)
                    for name in ('arc_labels', 'nodes_full', 'nodes_local', 'out_index', 'arc_index', 'type_nodes', 'type_offsets'):
                        assert np.array_equal(getattr(a, name), getattr(b, name)), name
                    assert a.dim_node_label == b.dim_node_label == list(dims)


def test_hub_rows_are_split_on_shards():
    """A shard whose adjacency has a row above 512 in-arcs uploads the light operator + segment lists like the single-GPU path
    (host-side part: the split reconstructs the row; the device part is tests/test_gpu_round3.py::test_hub_rows_on_shards)."""
    from gnnkeras_amd.sparse import split_heavy, HEAVY_THRESHOLD
    rng = np.random.default_rng(0)
    N = 3000
    ids = np.unique(np.concatenate([er_graph(N, 9000, seed=2).arc_ids, np.stack([rng.choice(N, 1500, replace=False), np.full(1500, 77)], 1)]), axis=0)
    ids = ids[ids[:, 0] != ids[:, 1]]
    g = GraphObject(np.ones((N, 2)), np.concatenate([ids, np.ones((len(ids), 1))], 1), np.ones((N, 1)), focus='n', aggregation_mode='average')
    p = ShardPlan(g, 0, 2)
    assert p.adjacency.max_degree > HEAVY_THRESHOLD
    light, heavy = split_heavy(p.adjacency)
    assert light.n_src == p.n_rows_full + heavy['n_seg'] and light.n_dst == p.n_local
    j = 77
    virt = light.src[light.rowptr[j]:light.rowptr[j + 1]]
    assert np.array_equal(virt, p.n_rows_full + np.arange(heavy['n_seg']))
    seg = np.concatenate([p.adjacency.src[b:e] for b, e in zip(heavy['seg_beg'], heavy['seg_end'])])
    assert np.array_equal(seg, p.adjacency.src[p.adjacency.rowptr[j]:p.adjacency.rowptr[j + 1]])


def test_partition_and_padded_rows():
    chunk, ranges = partition(10, 4)
    assert chunk == 3 and ranges == [(0, 3), (3, 6), (6, 9), (9, 10)]
    assert list(padded_row(np.arange(10), 3)) == [0, 1, 2, 4, 5, 6, 8, 9, 10, 12]
    chunk, ranges = partition(5, 8)                                       # more ranks than chunks: empty tails
    assert chunk == 1 and ranges[5:] == [(5, 5)] * 3


def test_shard_plan_covers_every_arc_once():
    g = er_graph(300, 2500, dim_node_label=4, dim_arc_label=2, seed=3)
    plans = [ShardPlan(g, r, 4) for r in range(4)]
    assert sum(p.e_local for p in plans) == 2500 and sum(p.n_local for p in plans) == 300
    assert np.array_equal(np.sort(np.concatenate([p.arc_index for p in plans])), np.arange(2500))
    for p in plans:
        # local rows keep ascending-source order and the whole-graph aggregation weights ('average' = 1/in-degree)
        for j in range(p.n_local):
            s = p.adjacency.src[p.adjacency.rowptr[j]:p.adjacency.rowptr[j + 1]]
            assert np.all(np.diff(s) > 0)
        indeg = np.bincount(g.arc_ids[:, 1], minlength=300)[p.lo:p.hi]
        assert np.array_equal(np.diff(p.adjacency.rowptr), indeg)
        assert p.adjacency.w is None and np.allclose(p.adjacency.row_scale[indeg > 0], 1 / indeg[indeg > 0])
        full = p.pad_state(np.arange(300 * 2, dtype=np.float32).reshape(300, 2), 16)
        assert full.shape == (4 * (p.chunk + 1), 16)
        assert np.array_equal(full[padded_row(np.arange(300), p.chunk), :2].reshape(-1), np.arange(600))
        assert np.all(full[p.chunk::p.chunk + 1] == 0)                    # flag rows start clear


def test_halo_plan_lists_are_consistent():
    """What rank p expects from rank r (halo[r] on p) is exactly what r packs for p (send_rows[p] on r), in the same
    order; block-diagonal batches sharded on graph boundaries need no halo at all."""
    g = er_graph(300, 2500, dim_node_label=4, dim_arc_label=2, seed=3)
    R = 4
    plans = [HaloShardPlan(g, r, R) for r in range(R)]
    for p_ in plans:
        for r_ in plans:
            if p_.rank == r_.rank: continue
            assert np.array_equal(p_.halo[r_.rank], r_.send_rows[p_.rank] + r_.lo)
            assert p_.recv_rows[r_.rank] == r_.send_counts[p_.rank]
        assert len(p_.pack_index) == sum(p_.send_counts)
        # every source id of the local operator points at the right global node
        srcs = p_.adjacency.src
        assert np.all(p_.view_global[srcs] >= 0)
        dst = np.repeat(np.arange(p_.n_local), np.diff(p_.adjacency.rowptr)) + p_.lo
        pairs = set(zip(p_.view_global[srcs].tolist(), dst.tolist()))
        want = set(map(tuple, g.arc_ids[(g.arc_ids[:, 1] >= p_.lo) & (g.arc_ids[:, 1] < p_.hi)].tolist()))
        assert pairs == want
    # two disjoint components cut exactly at the boundary: nothing but flag rows travels
    ring = lambda n, off: np.array([[off + i, off + (i + 1) % n, 1.0] for i in range(n)])
    gg = GraphObject(np.ones((20, 2)), np.concatenate([ring(10, 0), ring(10, 10)]), np.ones((20, 1)), focus='n')
    for r in range(2):
        hp = HaloShardPlan(gg, r, 2)
        assert sum(len(h) for h in hp.halo) == 0 and hp.send_counts[1 - r] == 1 and hp.n_rows_view == 10 + 1 + 1


def test_exchange_choice_is_a_property_of_the_graph_not_of_the_rank():
    """One rank's halo is almost everything, the other's is empty: both must still pick the same collective."""
    rng = np.random.default_rng(0)
    n = 400
    # arcs into the first half come from everywhere (rank 0 reads most of rank 1's rows); the second half only reads itself
    a = np.stack([rng.integers(0, n, 4000), rng.integers(0, n // 2, 4000)], 1)
    b = np.stack([rng.integers(n // 2, n, 4000), rng.integers(n // 2, n, 4000)], 1)
    ab = np.concatenate([a, b])
    ids = np.unique(ab[ab[:, 0] != ab[:, 1]], axis=0).astype(float)
    g = GraphObject(rng.normal(size=(n, 3)), np.concatenate([ids, np.ones((len(ids), 1))], 1), rng.normal(size=(n, 2)),
                    focus='n', aggregation_mode='sum')
    assert choose_exchange(g, 2) == 'allgather'                # decided by the WORST rank
    block = np.array([[i, (i + 1) % (n // 2)] for i in range(n // 2)] + [[n // 2 + i, n // 2 + (i + 1) % (n // 2)] for i in range(n // 2)], float)
    g2 = GraphObject(rng.normal(size=(n, 3)), np.concatenate([block, np.ones((len(block), 1))], 1), rng.normal(size=(n, 2)),
                     focus='n', aggregation_mode='sum')
    assert choose_exchange(g2, 2) == 'halo'                    # block-diagonal: nothing but flags to exchange


def test_split_csr_partitions_every_row_in_order():
    g = er_graph(300, 2500, dim_node_label=4, dim_arc_label=2, seed=3)
    for plan in (ShardPlan(g, 1, 4), HaloShardPlan(g, 2, 4)):
        c = plan.adjacency
        own, halo = split_csr(c, *plan.own_rows)
        assert own.nnz + halo.nnz == c.nnz and own.row_scale is None and halo.row_scale is c.row_scale
        lo, hi = plan.own_rows
        assert np.all((own.src >= lo) & (own.src < hi)) and not np.any((halo.src >= lo) & (halo.src < hi))
        for j in range(c.n_dst):
            row = c.src[c.rowptr[j]:c.rowptr[j + 1]]
            a, b = own.src[own.rowptr[j]:own.rowptr[j + 1]], halo.src[halo.rowptr[j]:halo.rowptr[j + 1]]
            assert np.array_equal(a, row[(row >= lo) & (row < hi)]) and np.array_equal(b, row[(row < lo) | (row >= hi)])


@pytest.mark.parametrize('transport,overlap', [('direct', False), ('direct', True), ('measured', True)])
def test_whole_slice_transports_agree(transport, overlap):
    """The one-hop all-gather (R - 1 concurrent send / receive pairs) moves the same slices as the all-gather collective, and
    the measured choice between the two is the same on every rank."""
    world, threshold, d = 3, 0.02, 6
    g, model, s0 = _problem(threshold, d)
    seq = MultiGraphSequencer([g], 'n', 'average', 1, shuffle=False, device='cpu')
    k_ref, st_ref, out_ref = oracle_loop(model, seq[0][0], s0, np.float64)
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 29500 + (os.getpid() + 977 + 31 * overlap + 7 * len(transport)) % 1000
    procs = [ctx.Process(target=_worker, args=(r, world, port, threshold, d, q, False, overlap, transport)) for r in range(world)]
    for p in procs: p.start()
    res = sorted([q.get(timeout=180) for _ in procs])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert all(r[1] == float(k_ref) for r in res)
    assert rel_err(np.concatenate([r[2] for r in res]), st_ref) < 1e-6 and rel_err(np.concatenate([r[3] for r in res]), out_ref) < 1e-6


# ----------------------------------------------------------------------------------------------------------------------
# arc- and graph-focused models on shards (reference GNN.py:317-330, :341-346)
# ----------------------------------------------------------------------------------------------------------------------
def _problem_focus(focus, d):
    rng = np.random.default_rng(3)
    if focus == 'a':
        g0 = er_graph(203, 1500, dim_node_label=5, dim_arc_label=2, seed=7)
        E = g0.arcs.shape[0]
        om = rng.random(E) < 0.6
        g = GraphObject(g0.nodes, g0.arcs, rng.normal(size=(int(om.sum()), 2)), focus='a', set_mask=rng.random(E) < 0.9, output_mask=om,
                        aggregation_mode='average')
        n_t = int((g.set_mask & g.output_mask).sum())
        g = GraphObject(g0.nodes, g0.arcs, rng.normal(size=(int(om.sum()), 2)), focus='a', set_mask=g.set_mask, output_mask=om,
                        aggregation_mode='average')
        cls = GNNarcBased
    else:
        parts = [er_graph(n, 6 * n, dim_node_label=5, dim_arc_label=2, seed=11 + n) for n in (40, 71, 23, 69)]      # 203 nodes, 4 graphs
        parts = [GraphObject(q.nodes, q.arcs, rng.normal(size=(1, 2)), focus='g', aggregation_mode='average') for q in parts]
        g = GraphObject.merge(parts, focus='g', aggregation_mode='average')
        cls = GNNgraphBased
    inp, lay = get_inout_dims('state', 5, 2, 2, focus, d)
    ns = MLP(inp[0], lay, 'tanh', 'lecun_normal', 'lecun_normal', rng=0, device='cpu')
    ns.set_weights([w * 0.4 if w.ndim == 2 else w for w in ns.get_weights()])
    inp, lay = get_inout_dims('output', 5, 2, 2, focus, d)
    no = MLP(inp[0], lay, 'softmax', 'glorot_normal', 'glorot_normal', rng=1, device='cpu')
    model = cls(ns, no, d, 7, 0.01)
    N = g.nodes.shape[0]
    s0 = rng.normal(0, 0.1, (N, d)).astype(np.float32) if d else None
    return g, model, s0


def _worker_focus(rank, world, port, focus, d, overlap, from_slice, out_q):
    os.environ['MASTER_ADDR'], os.environ['MASTER_PORT'] = '127.0.0.1', str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        g, model, s0 = _problem_focus(focus, d)
        if from_slice:
            from gnnkeras_amd.distributed import GraphSlice
            g = GraphSlice.from_graph(g, *partition(g.nodes.shape[0], world)[1][rank], focus=focus)
        sl = OracleShardedLoop(model, g, rank, world, 'cpu', overlap=overlap)
        k, state, out = sl.forward(s0)
        ids = sl.plan.arc_out_index if focus == 'a' else None
        out_q.put((rank, float(k), state.numpy(), out.numpy(), ids))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('focus,d,world,overlap,from_slice', [('a', 6, 2, False, False), ('a', 0, 3, True, True), ('g', 6, 3, False, True),
                                                              ('g', 0, 2, True, False)])
def test_arc_and_graph_focused_models_on_shards(focus, d, world, overlap, from_slice):
    """Arc focus: every rank applies the output network to the masked arcs whose DESTINATION it owns (their sources' final states
    come out of the exchanged buffer); the ranks' rows, placed by their global arc ids, are the single-process output.  Graph focus:
    per-graph partial sums over the own nodes, one all-reduce; every rank ends with the complete pooled output.  Graph boundaries do
    not coincide with the shard boundaries."""
    g, model, s0 = _problem_focus(focus, d)
    seq = MultiGraphSequencer([g], focus, 'average', 1, shuffle=False, device='cpu')
    k_ref, st_ref, out_ref = oracle_loop(model, seq[0][0], s0, np.float64)
    assert 1 < k_ref <= model.max_iteration
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 29500 + (os.getpid() + 311 * (focus == 'a') + 17 * world + d) % 1000
    procs = [ctx.Process(target=_worker_focus, args=(r, world, port, focus, d, overlap, from_slice, q)) for r in range(world)]
    for p in procs: p.start()
    res = sorted([q.get(timeout=180) for _ in procs], key=lambda r: r[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert all(r[1] == float(k_ref) for r in res)
    assert rel_err(np.concatenate([r[2] for r in res]), st_ref) < 1e-6
    if focus == 'g':
        for r in res: assert r[3].shape == out_ref.shape and rel_err(r[3], out_ref) < 1e-6          # complete on every rank
    else:
        mask = np.flatnonzero(g.set_mask & g.output_mask)                       # global arc ids of the output rows, ascending
        out = np.full(out_ref.shape, np.nan, dtype=np.float32)
        for r in res: out[np.searchsorted(mask, r[4])] = r[3]
        assert sum(len(r[4]) for r in res) == len(mask) and rel_err(out, out_ref) < 1e-6
