// gnn_train_step for HETEROGENEOUS models (reference GNN/Models/CompositeGNN.py:275-304: `train_step` of CompositeGNNnodeBased /
// graphBased): one state network per node type, each applied to the rows of its type
//     state_new[type t rows] = net_state[t]([labels[:, :d_t] | state | Adj^T state | aggregated_component][type t rows], training=True)
// (CompositeGNN.py:215-234), BatchNormalization of network t on the batch statistics of ITS rows, moving averages updated once per
// executed iteration and network, the output network on the converged state alone (CompositeGNN.py:237-239), Keras loss, and
// back-propagation through the k executed iterations.
//
// Round 3 trained such models from Python on the device building blocks (Models/training.py): ~25 launches per node type and
// iteration pair with the interpreter between them - 15.6 ms per step on batches of 32 small typed graphs (d = 32, 20 iterations)
// against 1.1 ms for a homogeneous batch of that size.  This is the same arithmetic with the orchestration inside the library
// (the general kernels of train_loop.hpp with per-type row lists: k_segdense / k_dense_grad_* / k_colstats_* take a row index per
// segment), one host synchronisation per step (to learn k).  Node, graph and arc focus (CompositeGNN.py:315-327: the output network over
// [state_src | state_dst | arc label] of the masked arcs); LGNN label gradients keep the building-block path.
#pragma once
// (included by gnnloop.hip behind train_loop.hpp: shares its anonymous-namespace helpers)
#include "train_composite_big.hpp"      // large graphs: the row-streaming kernels on per-type position ranges

namespace {

struct CType {                       // one node type
    const gnn_mlp_t *m;
    gnn_mlp_grads_t g;
    int count, d_t, in_dim, off_state, off_agg, off_comp;
    const int *rows;                 // device: node ids of this type, ascending
    NetCtx nc;
    float *stats, *stats_tpl;        // [K][2 in_dim], [2 in_dim]: mean | var per input column
    float *Wf, *bf;                  // [K][in_dim x H1], [K][H1]: folded first layers (thin first layers only)
    float *Gc, *dx;                  // [count, S] gathered gradient rows, [count, 2 S] d loss / d [state | agg] rows
};

struct CPlan {
    int N, E, S, L, A, M, R, T, K, G, n_types, sum_d, W_comp;
    bool pooled, agg_taped;
    int *grads_ok;               // first word of the tape (gnn_train_args_t::grads_ok_dev)
    int *flags; float *k_dev;
    float *states, *agg, *agg_comp;
    float *stats_o, *Wf_o, *bf_o, *dx_o_all, *dx_full, *G_state, *G_out, *dpred, *loss_rows, *loss_part, *part;
    size_t part_floats;
    NetCtx co;
    CType ty[GNN_MAX_TYPES];
    // small graphs: the K forward / k backward iterations as ONE persistent launch each (kernels_train_small.hpp), the nodes walked in type
    // order in tiles of <= 64 nodes of one type: the tape's rows are POSITIONS (position i = node type_nodes[i]), `inv` maps back
    bool small; int SPs, ldS, n_wg, wg_begin[GNN_MAX_TYPES + 1];
    int *inv; float *sm_cc, *sm_part, *sm_dxa, *sm_partW[GNN_MAX_TYPES], *sm_partBN[GNN_MAX_TYPES]; unsigned long long *sm_bar;
    int *isrc, *idst;            // arc focus: the end nodes of the masked arcs
    // large graphs (train_composite_big.hpp): the tape's rows are POSITIONS too, every type a contiguous range of them
    bool big; CBig B;
    bool head_fast; float *part_h;       // large graphs: a thin output head over EVERY node on the row-streaming head kernels (k_head_wgrad / k_head_dx)
    size_t bytes;
};

// inv[perm[i]] = i
__global__ void __launch_bounds__(256) k_invert_perm(const int *__restrict__ perm, int n, int *__restrict__ inv) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) inv[perm[i]] = i;
}

// rows[m] of a [*, ld_dst] matrix <- row m of a compact [M, width] one (every node has exactly one type: the types' rows partition it)
__global__ void __launch_bounds__(256) k_scatter_rows(const float *__restrict__ src, int ld_src, const int *__restrict__ idx, int M, int width,
                                                      float *__restrict__ dst, int ld_dst) {
    const size_t total = (size_t)M * width;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t m = i / width;
        const int j = (int)(i % width);
        dst[(size_t)idx[m] * ld_dst + j] = src[m * ld_src + j];
    }
}

int scatter_rows(const float *src, int ld_src, const int *idx, int M, int width, float *dst, int ld_dst, hipStream_t st) {
    if (M == 0 || width == 0) return 0;
    k_scatter_rows<<<std::min(cdiv((long)M * width, 256), 256 * 16), 256, 0, st>>>(src, ld_src, idx, M, width, dst, ld_dst);
    LAUNCH_OK();
    return 0;
}

int gather_rows(const float *src, int ld_src, const int *idx, int M, int width, float *dst, int ld_dst, hipStream_t st) {
    if (M == 0 || width == 0) return 0;
    const long total = (long)M * ((width & 3) == 0 ? width / 4 : width);
    gnn::k_gather_rows<<<std::min(cdiv(total, 256), 256 * 16), 256, 0, st>>>(src, ld_src, idx, M, width, dst, ld_dst);
    LAUNCH_OK();
    return 0;
}

int make_cplan(const gnn_train_args_t &ta, void *ws, CPlan &p) {
    const gnn_loop_args_t &a = ta.loop;
    memset(&p, 0, sizeof(p));
    if (a.abi_version != GNN_ABI_VERSION) return fail("abi_version %d != %d", a.abi_version, GNN_ABI_VERSION);
    if (a.n_nodes < 1) return fail("gnn_train_step: empty graph");
    if (a.n_types < 1 || a.n_types > GNN_MAX_TYPES) return fail("n_types %d out of [1,%d]", a.n_types, GNN_MAX_TYPES);
    if (a.max_iteration < 1) return fail("composite GNN requires max_iteration > 0");
    p.N = a.n_nodes; p.E = a.n_arcs; p.L = a.dim_node_label; p.A = a.dim_arc_label; p.n_types = a.n_types;
    p.S = a.state_dim > 0 ? a.state_dim : a.dim_node_label;
    p.K = a.max_iteration; p.M = a.n_out;
    for (int t = 0; t < p.n_types; ++t) {
        if (a.type_dim_label[t] < 0 || a.type_dim_label[t] > p.L) return fail("type_dim_label[%d]=%d out of [0,%d]", t, a.type_dim_label[t], p.L);
        p.sum_d += a.type_dim_label[t];
    }
    p.W_comp = p.sum_d + p.A;
    if (a.type_offsets[0] != 0 || a.type_offsets[p.n_types] != p.N) return fail("type_offsets must span [0, n_nodes]");
    const gnn_mlp_t &no = a.net_output;
    TRY(check_mlp(no, "net_output", ws != nullptr));
    TRY(check_dropout(ta.drop_output, no, "net_output"));
    bool drop_s = false;                               // (a state network with Dropout layers: the general kernels)
    for (int t = 0; t < p.n_types; ++t) {
        TRY(check_mlp(a.net_state[t], "net_state", ws != nullptr));
        TRY(check_dropout(ta.drop_state[t], a.net_state[t], "net_state"));
        drop_s |= ta.drop_state[t].n > 0;
    }
    const bool arc = a.focus == GNN_FOCUS_ARC;      // CompositeGNN.py:315-327: [state_src | state_dst | arc label] of the masked arcs
    const int expect_o = arc ? 2 * p.S + p.A : p.S;
    if (no.in_dim != expect_o) return fail("net_output.in_dim %d != %d expected for this focus (composite models filter on the state alone)", no.in_dim, expect_o);
    p.T = no.units[no.n_layers - 1];
    p.pooled = a.focus == GNN_FOCUS_GRAPH;
    p.G = p.pooled ? a.nodegraph.n_dst : 0;
    p.R = p.pooled ? p.G : p.M;

    // ---- which path: the persistent small-graph kernels (one-layer state networks of the state's width, every non-empty type with or every
    // one without BatchNormalization - the tiles meet at the same grid barriers -, all tiles resident at once) or one launch per layer, type
    // and iteration
    p.SPs = p.S <= 16 ? 16 : p.S <= 32 ? 32 : 64;
    p.n_wg = 0; p.wg_begin[0] = 0;
    bool uniform = p.S <= 64 && train_small_enabled() && p.N < train_big_min_nodes() && !drop_s;
    int bn_seen = -1;
    for (int t = 0; t < p.n_types; ++t) {
        const gnn_mlp_t &m = a.net_state[t];
        const int cnt = a.type_offsets[t + 1] - a.type_offsets[t];
        p.n_wg += cdiv(std::max(cnt, 0), 64);
        p.wg_begin[t + 1] = p.n_wg;
        if (m.n_layers != 1 || m.units[0] != p.S || m.activation[0] == GNN_ACT_SOFTMAX) uniform = false;
        if (cnt > 0) { if (bn_seen < 0) bn_seen = m.has_bn ? 1 : 0; else if (bn_seen != (m.has_bn ? 1 : 0)) uniform = false; }
    }
    p.small = uniform && p.n_wg >= 1 && p.n_wg <= std::min(device_cus(), 256) && (size_t)p.K * p.N * p.SPs * sizeof(float) <= agg_tape_budget();
    p.ldS = p.small ? p.SPs : p.S;
    p.big = !p.small && !drop_s && composite_big_applies(ta, p.N, p.S, p.W_comp, &p.B.XT);
    p.B.XW = 32 * p.B.XT;

    Carver c(ws);
    p.grads_ok = c.take<int>(4);
    p.flags = c.take<int>(p.K + 8);
    p.k_dev = c.take<float>(4);
    p.states = c.take<float>((size_t)(p.K + 1) * p.N * p.ldS);
    p.agg_taped = (size_t)p.K * p.N * p.ldS * sizeof(float) <= agg_tape_budget();
    p.agg = c.take<float>((size_t)(p.agg_taped ? p.K : 1) * p.N * p.ldS);
    p.agg_comp = c.take<float>((size_t)p.N * std::max(p.W_comp, 1));
    p.stats_o = c.take<float>(2 * (size_t)no.in_dim);
    p.Wf_o = c.take<float>((size_t)no.in_dim * no.units[0]); p.bf_o = c.take<float>(no.units[0]);
    p.dx_o_all = c.take<float>((size_t)std::max(p.M, 1) * no.in_dim);
    p.dx_full = c.take<float>((size_t)p.N * 2 * p.S);                // (large graphs: d loss / d [state | agg] by position)
    p.G_state = c.take<float>((size_t)p.N * p.S);
    p.G_out = c.take<float>((size_t)std::max(p.M, 1) * p.T);
    p.dpred = c.take<float>((size_t)std::max(p.R, 1) * p.T);
    p.loss_rows = c.take<float>(std::max(p.R, 1));
    p.loss_part = c.take<float>(256);
    p.isrc = c.take<int>(arc ? std::max(p.M, 1) : 0); p.idst = c.take<int>(arc ? std::max(p.M, 1) : 0);
    p.part_floats = 0;
    for (int t = 0; t < p.n_types; ++t) {
        CType &y = p.ty[t];
        y.m = &a.net_state[t];
        TRY(check_mlp(*y.m, "net_state", ws != nullptr));
        y.g = ta.grad_state_types[t];
        y.count = a.type_offsets[t + 1] - a.type_offsets[t];
        if (y.count < 0) return fail("type_offsets must ascend");
        y.rows = a.type_nodes ? a.type_nodes + a.type_offsets[t] : nullptr;
        y.d_t = a.type_dim_label[t];
        y.in_dim = y.d_t + 2 * p.S + p.W_comp;
        if (y.m->in_dim != y.in_dim) return fail("net_state[%d].in_dim %d != %d expected from the graph dims", t, y.m->in_dim, y.in_dim);
        if (y.m->units[y.m->n_layers - 1] != p.S) return fail("net_state[%d] output width %d != state width %d", t, y.m->units[y.m->n_layers - 1], p.S);
        y.off_state = y.d_t; y.off_agg = y.d_t + p.S; y.off_comp = y.d_t + 2 * p.S;
        carve_net(c, y.nc, *y.m, p.big ? 0 : y.count, p.part_floats, &ta.drop_state[t]);      // (large graphs: no per-layer row buffers - the kernels stream the tape)
        y.nc.m = y.m; y.nc.g = &y.g;
        y.stats = c.take<float>((size_t)p.K * 2 * y.in_dim);
        y.stats_tpl = c.take<float>(2 * (size_t)y.in_dim);
        y.Wf = c.take<float>((size_t)p.K * y.in_dim * y.m->units[0]);
        y.bf = c.take<float>((size_t)p.K * y.m->units[0]);
        y.Gc = c.take<float>(p.big ? 0 : (size_t)std::max(y.count, 1) * p.S);
        y.dx = c.take<float>(p.big ? 0 : (size_t)std::max(y.count, 1) * 2 * p.S);
        int nc_; rows_per_chunk_for(std::max(y.count, 1), &nc_);
        p.part_floats = std::max(p.part_floats, (size_t)(nc_ + 1) * y.in_dim);
        const int tiles_t = p.wg_begin[t + 1] - p.wg_begin[t];
        p.sm_partW[t] = c.take<float>(p.small ? (size_t)tiles_t * ((size_t)y.in_dim * p.S + p.S) : 0);
        p.sm_partBN[t] = c.take<float>(p.small ? (size_t)tiles_t * 2 * y.in_dim : 0);
    }
    // every node is an output row (out_index the identity: checked on the device, read with k), one thin Dense over the state alone
    p.head_fast = p.big && !arc && no.n_layers == 1 && no.units[0] <= 4 && p.M == p.N && p.S % 4 == 0 && p.S / 4 <= 32 && ta.drop_output.n == 0;
    p.part_h = c.take<float>(p.head_fast ? (size_t)BIG_HEAD_BLOCKS * ((size_t)no.in_dim * no.units[0] + no.units[0]) : 0);
    p.inv = c.take<int>((p.small || p.big) ? p.N : 0);
    if (p.big) {
        CBig &B = p.B;
        const gnn_csr_t &ad = a.adjacency, &as = ta.adjacency_by_source;
        B.scan_tmp = c.take<int>((size_t)cdiv(p.N, SCAN_CHUNK) + 2);
        B.d.rp = c.take<int>((size_t)p.N + 1); B.d.src = c.take<int>((size_t)std::max(ad.nnz, 1));
        B.d.w = c.take<float>(ad.w ? (size_t)std::max(ad.nnz, 1) : 0); B.d.row_scale = c.take<float>(ad.row_scale ? (size_t)p.N : 0);
        B.s.rp = c.take<int>((size_t)p.N + 1); B.s.src = c.take<int>((size_t)std::max(as.nnz, 1));
        B.s.w = c.take<float>(as.w ? (size_t)std::max(as.nnz, 1) : 0); B.s.row_scale = c.take<float>(as.row_scale ? (size_t)p.N : 0);
        if (!ad.w) B.d.w = nullptr;
        if (!ad.row_scale) B.d.row_scale = nullptr;
        if (!as.w) B.s.w = nullptr;
        if (!as.row_scale) B.s.row_scale = nullptr;
        B.xc = c.take<float>((size_t)p.N * B.XW);
        B.Cc = c.take<float>((size_t)p.N * p.S);
        B.Gpos = c.take<float>((size_t)p.N * p.S);
        int counts[GNN_MAX_TYPES];
        for (int t = 0; t < p.n_types; ++t) counts[t] = p.ty[t].count;
        split_blocks(std::min(device_cus(), BIG_FWD_BLOCKS), counts, p.n_types, B.fwd_blocks);          // one 8-wave workgroup per CU
        split_blocks(2 * device_cus(), counts, p.n_types, B.bwd_blocks);                                // 256-thread workgroups, two per CU
        split_blocks(std::min(device_cus(), BIG_WGRAD_BLOCKS), counts, p.n_types, B.wgrad_blocks);      // one workgroup per CU (LDS ring)
        B.one_act = true; B.act = -1;
        for (int t = 0; t < p.n_types; ++t) {
            const CType &y = p.ty[t];
            if (y.count > 0) { if (B.act < 0) B.act = y.m->activation[0]; else if (B.act != y.m->activation[0]) B.one_act = false; }
            B.fwd_blocks[t] = std::min(B.fwd_blocks[t], std::max(1, cdiv(cdiv(std::max(y.count, 1), 16), gnn::TB_WAVES)));
            B.bwd_blocks[t] = std::min(B.bwd_blocks[t], std::max(1, cdiv(cdiv(std::max(y.count, 1), 16), 4)));
            B.wgrad_blocks[t] = std::min(B.wgrad_blocks[t], std::max(1, cdiv(std::max(y.count, 1), 64)));
            if (y.count == 0) B.fwd_blocks[t] = B.bwd_blocks[t] = B.wgrad_blocks[t] = 0;
            B.part_a[t] = c.take<float>((size_t)BIG_AGG_BLOCKS * 2 * p.S);
            B.part_y[t] = c.take<float>((size_t)std::max(BIG_FWD_BLOCKS, 1) * 2 * std::max(p.S, B.XW));
            B.part_w[t] = c.take<float>((size_t)std::max(B.wgrad_blocks[t], 1) * ((size_t)y.in_dim * p.S + p.S));
        }
        for (int t = 0; t < p.n_types; ++t) {         // the constants line of type t: [labels[:, :d_t] | aggregated_component | 1 | 0 ..]
            const CType &y = p.ty[t];
            gnn::ConstCols &cc = B.cc[t];
            memset(&cc, 0, sizeof(cc));
            if (y.d_t > 0) { cc.width[cc.n] = y.d_t; cc.wrow[cc.n] = 0; ++cc.n; }
            if (p.W_comp > 0) { cc.width[cc.n] = p.W_comp; cc.wrow[cc.n] = y.off_comp; ++cc.n; }
            B.Kc[t] = y.d_t + p.W_comp;
        }
    }
    p.sm_cc = c.take<float>(p.small ? (size_t)p.N * p.SPs : 0);
    p.sm_part = c.take<float>(p.small ? (size_t)2 * p.n_wg * 8 * p.SPs : 0);
    p.sm_dxa = c.take<float>(p.small ? (size_t)p.N * p.SPs : 0);
    p.sm_bar = c.take<unsigned long long>(p.small ? 4 : 0);
    carve_net(c, p.co, no, p.M, p.part_floats, &ta.drop_output);
    p.co.m = &no; p.co.g = &ta.grad_output;
    {
        int nc_; rows_per_chunk_for(std::max(p.M, 1), &nc_);
        p.part_floats = std::max(p.part_floats, (size_t)(nc_ + 1) * no.in_dim);
    }
    p.part = c.take<float>(p.part_floats);
    p.bytes = (c.off + 255) & ~(size_t)255;
    return 0;
}

// segments of network `y`'s input at iteration t (CompositeGNN.py:222-223): [labels[:, :d_t] | state | agg | aggregated_component], all
// read through the type's row list; returns the count and the positions of the state / agg segments
int ctype_segs(const gnn_loop_args_t &a, const CPlan &p, const CType &y, int t, gnn::Seg *segs, int *i_state, int *i_agg) {
    int n = 0;
    const float *st_t = p.states + (size_t)t * p.N * p.S;
    const float *agg_t = p.agg + (p.agg_taped ? (size_t)t * p.N * p.S : 0);
    if (y.d_t > 0) segs[n++] = gnn::Seg{a.nodes, y.rows, a.ld_nodes, y.d_t, 0};
    *i_state = n; segs[n++] = gnn::Seg{st_t, y.rows, p.S, p.S, y.off_state};
    *i_agg = n;   segs[n++] = gnn::Seg{agg_t, y.rows, p.S, p.S, y.off_agg};
    if (p.W_comp > 0) segs[n++] = gnn::Seg{p.agg_comp, y.rows, p.W_comp, p.W_comp, y.off_comp};
    return n;
}

// ---- large graphs: the step in position space (train_composite_big.hpp) --------------------------------------------------------------------
// rows [off, off + count) of the by-destination / by-source adjacency in positions
gnn_csr_t big_sub_csr(const PosCsr &c, int N, int off, int count, bool with_w) {
    gnn_csr_t r;
    r.n_dst = count; r.n_src = N; r.nnz = c.nnz; r.rowptr = c.rp + off; r.src = c.src;
    r.w = with_w ? c.w : nullptr; r.row_scale = (with_w && c.row_scale) ? c.row_scale + off : nullptr;
    return r;
}

// positions, the re-labelled adjacency, state_0 by position, every type's constants line and the constant part of its first layer
int composite_big_setup(const gnn_train_args_t &ta, CPlan &p, hipStream_t st) {
    const gnn_loop_args_t &a = ta.loop;
    CBig &B = p.B;
    k_invert_perm<<<std::min(cdiv(p.N, 256), 1024), 256, 0, st>>>(a.type_nodes, p.N, p.inv);
    LAUNCH_OK();
    TRY(build_pos_csr(a.adjacency, a.type_nodes, p.inv, p.N, B.d, B.scan_tmp, st));
    TRY(build_pos_csr(ta.adjacency_by_source, a.type_nodes, p.inv, p.N, B.s, B.scan_tmp, st));
    if (a.state_dim > 0) TRY(gather_rows(a.state0, p.S, a.type_nodes, p.N, p.S, p.states, p.S, st));
    else TRY(gather_rows(a.nodes, a.ld_nodes, a.type_nodes, p.N, p.S, p.states, p.S, st));
    for (int q = 0; q < p.n_types; ++q) {
        const CType &y = p.ty[q];
        if (y.count == 0) continue;
        const int off = a.type_offsets[q];
        gnn::PackSegs ps;
        memset(&ps, 0, sizeof(ps));
        gnn::ConstSegs cs;
        memset(&cs, 0, sizeof(cs));
        if (y.d_t > 0) {
            ps.ptr[ps.n] = a.nodes; ps.ld[ps.n] = a.ld_nodes; ps.width[ps.n] = y.d_t; ps.wrow[ps.n] = 0; ++ps.n;
            cs.ptr[cs.n] = a.nodes; cs.ld[cs.n] = a.ld_nodes; cs.width[cs.n] = y.d_t; cs.wrow[cs.n] = 0; ++cs.n;
        }
        if (p.W_comp > 0) {
            ps.ptr[ps.n] = p.agg_comp; ps.ld[ps.n] = p.W_comp; ps.width[ps.n] = p.W_comp; ps.wrow[ps.n] = y.off_comp; ++ps.n;
            cs.ptr[cs.n] = p.agg_comp; cs.ld[cs.n] = p.W_comp; cs.width[cs.n] = p.W_comp; cs.wrow[cs.n] = y.off_comp; ++cs.n;
        }
        float *line = B.xc + (size_t)off * B.XW;
        const int grid = (int)std::min<long>(cdiv((long)y.count * B.XW, 256), 256 * 16);
        if (B.XT == 2) k_pack_xc_pos<64><<<grid, 256, 0, st>>>(y.count, y.rows, ps, line);
        else k_pack_xc_pos<32><<<grid, 256, 0, st>>>(y.count, y.rows, ps, line);
        LAUNCH_OK();
        if (y.m->has_bn) {
            // the batch statistics of the constant columns (the same in every iteration): one pass over the packed line, moments around its
            // row 0; each segment's columns land at its BatchNorm columns of the template, which every iteration's slot starts from
            HIP_OK(hipMemsetAsync(y.stats_tpl, 0, sizeof(float) * 2 * y.in_dim, st));
            int sgrid = 0;
            TRY(rows_stats(nullptr, line, B.XW, B.XW, y.count, B.part_y[q], st, &sgrid));
            int c0 = 0;
            for (int sg = 0; sg < ps.n; ++sg) {
                gnn::k_stats_finish<<<ps.width[sg], 256, 0, st>>>(nullptr, B.part_y[q] + c0, sgrid, B.XW, 1.0f / (float)y.count, y.stats_tpl + ps.wrow[sg],
                                                                 y.stats_tpl + y.in_dim + ps.wrow[sg], line + c0);
                LAUNCH_OK();
                c0 += ps.width[sg];
            }
            gnn::k_replicate<<<std::min(cdiv((long)2 * y.in_dim * p.K, 256), 1024), 256, 0, st>>>(y.stats_tpl, 2 * y.in_dim, p.K, y.stats);
            LAUNCH_OK();
        }
        // Cc = b + sum over the constant columns of (a (x - mean) + beta) W: the first layer over the line's segments alone, BatchNormalization
        // applied as the columns are staged (the mean leaves the value first)
        gnn::SegDenseArgs sa;
        memset(&sa, 0, sizeof(sa));
        sa.M = y.count; sa.H = p.S;
        int c0 = 0;
        for (int sg = 0; sg < ps.n; ++sg) { sa.seg[sa.nseg++] = gnn::Seg{line + c0, nullptr, B.XW, ps.width[sg], ps.wrow[sg]}; c0 += ps.width[sg]; }
        sa.W = y.m->kernel[0]; sa.ldw = p.S; sa.bias = y.m->bias[0]; sa.act = GNN_ACT_LINEAR;
        sa.Y = B.Cc + (size_t)off * p.S; sa.ldy = p.S;
        if (y.m->has_bn) { sa.in_gamma = y.m->bn_gamma; sa.in_beta = y.m->bn_beta; sa.in_mean = y.stats_tpl; sa.in_var = y.stats_tpl + y.in_dim; sa.in_eps = y.m->bn_eps; }
        if (sa.nseg > 0) TRY(launch_segdense(sa, st));
        else TRY(launch_copy2d(nullptr, y.m->bias[0], 0, sa.Y, p.S, y.count, p.S, p.S, st));      // (no constant inputs at all: the bias row)
    }
    return 0;
}

// the K gated training-mode iterations (CompositeGNN.py:215-234), every type's rows by its own network.  Per iteration: one neighbour sum
// (+ column statistics) per type, then ONE launch each for the statistics' finish, the folds, the first layers of all types (when they share
// the activation: the kernel is compiled per activation) and the new rows' statistics.
int composite_big_forward(const gnn_train_args_t &ta, CPlan &p, hipStream_t st) {
    const gnn_loop_args_t &a = ta.loop;
    CBig &B = p.B;
    const size_t NS = (size_t)p.N * p.S;
    for (int t = 0; t < p.K; ++t) {
        const int *gate = p.flags + t;
        const float *s_t = p.states + (size_t)t * NS;
        float *s_n = p.states + (size_t)(t + 1) * NS;
        float *agg_t = p.agg + (p.agg_taped ? (size_t)t * NS : 0);
        gnn::StatsFinishJobs fin;
        memset(&fin, 0, sizeof(fin));
        int n_fin = 0;
        for (int q = 0; q < p.n_types; ++q) {
            CType &y = p.ty[q];
            if (y.count == 0) continue;
            const int off = a.type_offsets[q];
            const size_t ro = (size_t)off * p.S;
            const gnn_csr_t cd = big_sub_csr(B.d, p.N, off, y.count, true);
            float *stats = y.stats + (size_t)t * 2 * y.in_dim;
            if (y.m->has_bn) {
                // (moments around the previous iteration's column means, as the homogeneous step takes them)
                const float *prev = t > 0 ? y.stats + (size_t)(t - 1) * 2 * y.in_dim : nullptr;
                int grid = 0;
                TRY(launch_aggregate_stats(gate, cd, s_t, p.S, agg_t + ro, B.part_a[q], nullptr, nullptr, prev ? prev + y.off_agg : nullptr, st, &grid));
                fin.t[n_fin++] = gnn::StatsFinishT{B.part_a[q], grid, 1.0f / (float)y.count, stats + y.off_agg, stats + y.in_dim + y.off_agg, prev ? prev + y.off_agg : nullptr};
                if (t == 0) {       // (later iterations: the launch that wrote the rows left their statistics)
                    TRY(rows_stats(gate, s_t + ro, p.S, p.S, y.count, B.part_y[q], st, &grid));
                    fin.t[n_fin++] = gnn::StatsFinishT{B.part_y[q], grid, 1.0f / (float)y.count, stats + y.off_state, stats + y.in_dim + y.off_state, s_t + ro};
                }
            } else TRY(launch_aggregate(gate, cd, s_t, p.S, p.S, agg_t + ro, p.S, st));
        }
        if (n_fin > 0) { gnn::k_stats_finish_jobs<<<dim3(p.S, n_fin), 256, 0, st>>>(gate, fin, p.S); LAUNCH_OK(); }
        // the state / agg rows of every type's first layer with this iteration's statistics folded in; the bias the kernel adds is the shift of
        // THOSE columns alone (sum beta W over them) - b and the constant columns' share sit in Cc
        FoldList fl;
        gnn::TypeLaunch<gnn::TrainFwdArgs> fw;
        memset(&fw, 0, sizeof(fw));
        gnn::StatsFinishJobs nxt;
        memset(&nxt, 0, sizeof(nxt));
        int n_nxt = 0;
        fw.n = p.n_types;
        for (int q = 0; q < p.n_types; ++q) {
            CType &y = p.ty[q];
            fw.blk_begin[q + 1] = fw.blk_begin[q] + (y.count > 0 ? B.fwd_blocks[q] : 0);
            if (y.count == 0) continue;
            const gnn_mlp_t &ns = *y.m;
            const bool bn = ns.has_bn != 0;
            const size_t ro = (size_t)a.type_offsets[q] * p.S;
            float *stats = y.stats + (size_t)t * 2 * y.in_dim;
            float *Wf = y.Wf + (size_t)t * y.in_dim * p.S, *bf = y.bf + (size_t)t * p.S;
            gnn::FoldJob &j = fl.fa.job[fl.fa.n_jobs++];
            j.centred = 1;
            j.W = ns.kernel[0]; j.b = nullptr; j.K = y.in_dim; j.H = p.S;
            j.gamma = bn ? ns.bn_gamma : nullptr; j.beta = ns.bn_beta; j.mean = stats; j.var = stats + y.in_dim; j.eps = ns.bn_eps;
            j.Wf = Wf; j.bf = bf; j.blk_begin = fl.blocks;
            j.dyn0 = y.off_state; j.dyn1 = y.off_agg; j.dyn_w = p.S;
            fl.blocks += j.H;
            gnn::TrainFwdArgs &fa = fw.t[q];
            fa.in_mean = bn ? stats : nullptr;
            fa.gate = gate; fa.M = y.count;
            fa.state = s_t + ro; fa.ld_state = p.S; fa.agg = agg_t + ro; fa.ld_agg = p.S;
            fa.addend = B.Cc + ro; fa.ld_add = p.S;
            fa.Wf = Wf; fa.bf = bf; fa.H = p.S; fa.wrow_state = y.off_state; fa.wrow_agg = y.off_agg;
            fa.act = ns.activation[0];
            fa.Y = s_n + ro; fa.ldy = p.S;
            fa.thr = a.state_threshold; fa.pred_flag = p.flags + t + 1; fa.pred_k = p.k_dev; fa.pred_kval = (float)(t + 1);
            const bool next_stats = bn && t + 1 < p.K;
            fa.stat_part = next_stats ? B.part_y[q] : nullptr;
            fa.stat_shift = next_stats ? stats + y.off_state : nullptr;       // (the new rows' moments around the input rows' column means)
            if (next_stats) {
                float *nx = y.stats + (size_t)(t + 1) * 2 * y.in_dim;
                nxt.t[n_nxt++] = gnn::StatsFinishT{B.part_y[q], B.one_act ? B.fwd_blocks[q] : 0 /* (set below) */, 1.0f / (float)y.count, nx + y.off_state,
                                                   nx + y.in_dim + y.off_state, stats + y.off_state};
            }
        }
        if (fl.fa.n_jobs > 0) TRY(launch_fold_list(fl, st));
        if (B.one_act) TRY(launch_train_fwd_types(fw, B.act, p.S, st));
        else {
            int jn = 0;
            for (int q = 0; q < p.n_types; ++q) {
                if (p.ty[q].count == 0) continue;
                int grid = 0;
                TRY(launch_train_fwd_add(fw.t[q], p.S, st, &grid));
                if (fw.t[q].stat_part) nxt.t[jn++].n_part = grid;
            }
        }
        if (n_nxt > 0) { gnn::k_stats_finish_jobs<<<dim3(p.S, n_nxt), 256, 0, st>>>(gate, nxt, p.S); LAUNCH_OK(); }
    }
    return 0;
}

// back-propagation through the k executed iterations; p.G_state = d loss / d state_k in the caller's node order.  Per iteration ONE launch each
// for the weight-gradient products, their reduction, the parameter gradients and the input gradients of all types, then one transposed
// neighbour sum per type (its epilogue applies the type's activation and BatchNormalization coefficients).
int composite_big_backward(const gnn_train_args_t &ta, CPlan &p, int k, hipStream_t st) {
    const gnn_loop_args_t &a = ta.loop;
    CBig &B = p.B;
    const size_t NS = (size_t)p.N * p.S;
    if (k == 0) return 0;
    {   // the first dZ: G by position, times act'(state_k) of the row's own type
        ActRanges ar;
        memset(&ar, 0, sizeof(ar));
        ar.n = p.n_types;
        for (int q = 0; q < p.n_types; ++q) { ar.begin[q] = a.type_offsets[q]; ar.act[q] = p.ty[q].m->activation[0]; }
        ar.begin[p.n_types] = p.N;
        k_gather_rows_dz<<<(int)std::min<long>(cdiv((long)p.N * (p.S / 4), 256), 256 * 16), 256, 0, st>>>(p.G_state, a.type_nodes, p.states + (size_t)k * NS, p.N, p.S, ar, B.Gpos);
        LAUNCH_OK();
    }
    const bool unit_w = !a.adjacency.w;       // entries depend on the destination only: the agg-half of a row's gradient is scaled once, the transposed walk is unit-weight
    float *dx = p.dx_full;
    int max_in = 1;
    for (int q = 0; q < p.n_types; ++q) max_in = std::max(max_in, p.ty[q].in_dim);
    for (int t = k - 1; t >= 0; --t) {
        const float *s_t = p.states + (size_t)t * NS;
        const float *agg_t = p.agg + (p.agg_taped ? (size_t)t * NS : 0);
        if (!p.agg_taped) {
            const gnn_csr_t all = big_sub_csr(B.d, p.N, 0, p.N, true);
            TRY(launch_aggregate(nullptr, all, s_t, p.S, p.S, p.agg, p.S, st));
        }
        gnn::TypeLaunch<gnn::TrainWgradArgs> wg;
        gnn::TypeLaunch<gnn::TrainBwdArgs> bw;
        gnn::ReduceTypes rd;
        gnn::ParamGradsTypes pg;
        memset(&wg, 0, sizeof(wg)); memset(&bw, 0, sizeof(bw)); memset(&rd, 0, sizeof(rd)); memset(&pg, 0, sizeof(pg));
        wg.n = bw.n = p.n_types;
        for (int q = 0; q < p.n_types; ++q) {
            CType &y = p.ty[q];
            wg.blk_begin[q + 1] = wg.blk_begin[q]; bw.blk_begin[q + 1] = bw.blk_begin[q];
            if (y.count == 0) continue;
            const gnn_mlp_t &ns = *y.m;
            const bool bn = ns.has_bn != 0;
            const int off = a.type_offsets[q];
            const size_t ro = (size_t)off * p.S;
            const float *stats = bn ? y.stats + (size_t)t * 2 * y.in_dim : nullptr;
            // P = [state | agg | constants]^T dZ and q = colsum(dZ) over the type's rows, then its parameter gradients and m1 / m2
            gnn::TrainWgradArgs &wa = wg.t[q];
            wa.M = y.count; wa.rows_per_wg = cdiv(cdiv(y.count, B.wgrad_blocks[q]), 64) * 64;
            wa.G = B.Gpos + ro; wa.Y = nullptr; wa.act = GNN_ACT_LINEAR;
            wa.state = s_t + ro; wa.agg = agg_t + ro; wa.xc = B.xc + (size_t)off * B.XW;
            wa.K = y.in_dim; wa.wrow_state = y.off_state; wa.wrow_agg = y.off_agg; wa.Kc = B.Kc[q]; wa.cs = B.cc[q];
            wa.part = B.part_w[q];
            wa.mean = stats;
            const int grid = cdiv(y.count, wa.rows_per_wg);
            wg.blk_begin[q + 1] = wg.blk_begin[q] + grid;
            const int nP = y.in_dim * p.S + p.S;
            rd.t[q] = gnn::ReduceT{B.part_w[q], grid, nP, y.nc.P, 0, 1.0f, y.in_dim * p.S, y.nc.q};
            pg.t[q] = gnn::ParamGradsT{y.nc.P, y.nc.q, ns.kernel[0], y.in_dim, p.S, bn ? ns.bn_gamma : nullptr, ns.bn_beta, stats, stats ? stats + y.in_dim : nullptr, ns.bn_eps,
                                       1.0f / (float)y.count, y.g.dkernel[0], y.g.dbias[0], y.g.dgamma, y.g.dbeta, bn ? y.nc.m1 : nullptr, bn ? y.nc.m2 : nullptr,
                                       t != k - 1 ? 1 : 0, 1, stats ? 1 : 0};
            gnn::TrainBwdArgs &ba = bw.t[q];
            ba.M = y.count; ba.dZ = B.Gpos + ro; ba.ldz = p.S;
            ba.W = ns.kernel[0]; ba.ldw = p.S; ba.H = p.S; ba.S = p.S; ba.wrow_state = y.off_state; ba.wrow_agg = y.off_agg;
            ba.state = s_t + ro; ba.ld_state = p.S; ba.agg = agg_t + ro; ba.ld_agg = p.S;
            if (bn) { ba.gamma = ns.bn_gamma; ba.mean = stats; ba.var = stats + y.in_dim; ba.m1 = y.nc.m1; ba.m2 = y.nc.m2; ba.eps = ns.bn_eps; }
            ba.defer_state_bn = bn ? 1 : 0;    // (k_aggregate_dz below adds the rest of the state half's BatchNorm gradient: it reads state_t anyway)
            ba.agg_row_scale = (unit_w && B.d.row_scale) ? B.d.row_scale + off : nullptr;
            ba.dx = dx + (size_t)off * 2 * p.S; ba.ld_dx = 2 * p.S;
            bw.blk_begin[q + 1] = bw.blk_begin[q] + B.bwd_blocks[q];
        }
        TRY(launch_train_wgrad_types(wg, p.S, B.XT, st));
        gnn::k_reduce_partials_types<<<dim3(cdiv(max_in * p.S + p.S, 64), p.n_types), 256, 0, st>>>(rd);
        LAUNCH_OK();
        gnn::k_first_layer_param_grads_types<<<dim3(max_in, p.n_types), 64, 0, st>>>(pg);
        LAUNCH_OK();
        if (t == 0) break;                 // nothing consumes d loss / d state_0: no input gradient, no transposed sum
        TRY(launch_train_bwd_types(bw, p.S, st));
        // dZ_{t-1} = (dx_state' + Adj . dx_agg + the deferred BatchNorm term) (.) act'(state_t): arcs by source, every type's rows with ITS network's
        // coefficients and activation (state_t's rows of type q are outputs of network q and inputs of network q)
        for (int q = 0; q < p.n_types; ++q) {
            const CType &y = p.ty[q];
            if (y.count == 0) continue;
            const gnn_mlp_t &ns = *y.m;
            const int off = a.type_offsets[q];
            const size_t ro = (size_t)off * p.S;
            const gnn_csr_t cs = big_sub_csr(B.s, p.N, off, y.count, !unit_w);
            gnn::AggDzArgs z;
            memset(&z, 0, sizeof(z));
            z.Y = s_t + ro; z.ldy = p.S; z.wrow_state = y.off_state;
            if (ns.has_bn) {
                const float *stats = y.stats + (size_t)t * 2 * y.in_dim;
                z.gamma = ns.bn_gamma; z.var = stats + y.in_dim; z.mean = stats; z.m1 = y.nc.m1; z.m2 = y.nc.m2; z.eps = ns.bn_eps;
            }
            if (!launch_aggregate_dz(cs, dx + p.S, 2 * p.S, B.Gpos + ro, p.S, dx + (size_t)off * 2 * p.S, 2 * p.S, z, ns.activation[0], p.S, st))
                return fail("k_aggregate_dz: no instance for state width %d / activation %d", p.S, ns.activation[0]);
            LAUNCH_OK();
        }
    }
    return 0;
}

size_t composite_train_workspace_bytes(const gnn_train_args_t &ta) {
    CPlan p;
    if (make_cplan(ta, nullptr, p)) return 0;
    return p.bytes;
}

int train_step_composite(const gnn_train_args_t &ta) {
    const gnn_loop_args_t &a = ta.loop;
    CPlan p;
    TRY(make_cplan(ta, ta.tape, p));
    if (!ta.tape || ta.tape_bytes < p.bytes) return fail("tape too small: %zu < %zu bytes", ta.tape_bytes, p.bytes);
    TRY(check_csr(a.adjacency, "adjacency", p.N, p.N));
    TRY(check_csr(a.arcnode, "arcnode", p.N, p.E));
    TRY(check_csr(ta.adjacency_by_source, "adjacency_by_source", p.N, p.N));
    for (int t = 0; t < p.n_types; ++t) {
        TRY(check_csr(a.composite_adjacency[t], "composite_adjacency", p.N, p.N));
        TRY(check_grads(*p.ty[t].m, p.ty[t].g, "grad_state_types"));
    }
    if (!a.nodes || !a.type_nodes) return fail("nodes / type_nodes is NULL");
    if (a.state_dim > 0 && !a.state0) return fail("state0 is required when state_dim > 0");
    if (p.M > 0 && !a.out_index) return fail("out_index is NULL");
    if (p.E > 0 && p.A > 0 && !a.arc_labels) return fail("arc_labels is NULL");
    if (a.focus == GNN_FOCUS_ARC && p.E > 0 && (!a.arc_src || !a.arc_dst)) return fail("arc focus needs arc_src / arc_dst");
    if (p.pooled) {
        if (a.nodegraph.n_src != p.M) return fail("graph focus: NodeGraph has %d rows but %d nodes pass the mask", a.nodegraph.n_src, p.M);
        TRY(check_csr(a.nodegraph, "nodegraph", p.G, p.M));
        TRY(check_csr(ta.nodegraph_by_source, "nodegraph_by_source", p.M, p.G));
    }
    if (p.R > 0 && !ta.targets) return fail("targets is NULL");
    if (ta.loss_kind < 0 || ta.loss_kind > 3) return fail("unknown loss kind %d", ta.loss_kind);
    if (!ta.y_pred || !ta.loss || !ta.k_host || !ta.state) return fail("y_pred / loss / k_host / state is NULL");
    TRY(check_grads(a.net_output, ta.grad_output, "grad_output"));
    const gnn_mlp_t &no = a.net_output;
    const bool bn_o = no.has_bn != 0;
    hipStream_t st = (hipStream_t)a.stream;
    const size_t NS = (size_t)p.N * p.ldS;             // one state matrix of the tape

    // ---- setup: transposes, aggregated_component (CompositeGNN.py:251-253), state_0, the constant columns' statistics ------------
    if (ta.prev_grads_ok_host) HIP_OK(hipMemcpyAsync(ta.prev_grads_ok_host, p.grads_ok, sizeof(int), hipMemcpyDeviceToHost, st));
    HIP_OK(hipMemsetAsync(p.grads_ok, 0, sizeof(int) * 4, st));
    if (ta.grads_ok_dev) *ta.grads_ok_dev = p.grads_ok;
    for (int t = 0; t < p.n_types; ++t) TRY(transposes(p.ty[t].nc, st));
    TRY(transposes(p.co, st));
    HIP_OK(hipMemsetAsync(p.flags, 0, sizeof(int) * (p.K + 8), st));
    HIP_OK(hipMemsetAsync(p.k_dev, 0, sizeof(float) * 4, st));
    {
        int col = 0;
        for (int t = 0; t < p.n_types; ++t) {
            const int dt = a.type_dim_label[t];
            if (dt > 0) TRY(launch_aggregate(nullptr, a.composite_adjacency[t], a.nodes, a.ld_nodes, dt, p.agg_comp + col, p.W_comp, st));
            col += dt;
        }
        if (p.A > 0) TRY(launch_aggregate(nullptr, a.arcnode, a.arc_labels, a.ld_arcs, p.A, p.agg_comp + col, p.W_comp, st));
    }
    if (p.small) {          // the tape's rows are positions in type order (position i = node type_nodes[i]), ldS floats each, pad columns zero
        if (p.ldS != p.S) HIP_OK(hipMemsetAsync(p.states, 0, sizeof(float) * NS, st));
        if (a.state_dim > 0) TRY(gather_rows(a.state0, p.S, a.type_nodes, p.N, p.S, p.states, p.ldS, st));
        else TRY(gather_rows(a.nodes, a.ld_nodes, a.type_nodes, p.N, p.S, p.states, p.ldS, st));
        k_invert_perm<<<std::min(cdiv(p.N, 256), 1024), 256, 0, st>>>(a.type_nodes, p.N, p.inv);
        LAUNCH_OK();
    } else if (p.big) {     // (composite_big_setup below: state_0 by position)
    } else if (a.state_dim > 0) HIP_OK(hipMemcpyAsync(p.states, a.state0, sizeof(float) * NS, hipMemcpyDeviceToDevice, st));
    else TRY(launch_copy2d(nullptr, a.nodes, a.ld_nodes, p.states, p.S, p.N, p.S, p.S, st));
    gnn::Seg segs[GNN_MAX_SEGS];
    for (int t = 0; t < p.n_types && !p.big; ++t) {      // (large graphs: composite_big_setup takes them from the packed constants line, one pass)
        CType &y = p.ty[t];
        if (!y.m->has_bn || y.count == 0) continue;
        int is_, ia_;
        const int n = ctype_segs(a, p, y, 0, segs, &is_, &ia_);
        gnn::Seg cst[GNN_MAX_SEGS]; int nc = 0;
        for (int s = 0; s < n; ++s) if (s != is_ && s != ia_) cst[nc++] = segs[s];
        HIP_OK(hipMemsetAsync(y.stats_tpl, 0, sizeof(float) * 2 * y.in_dim, st));
        TRY(colstats_segs(nullptr, cst, nc, y.count, y.stats_tpl, y.stats_tpl + y.in_dim, p.part, st));
        gnn::k_replicate<<<std::min(cdiv((long)2 * y.in_dim * p.K, 256), 1024), 256, 0, st>>>(y.stats_tpl, 2 * y.in_dim, p.K, y.stats);
        LAUNCH_OK();
    }

    // (gnn_last_kernel_name(): which orchestration took the step - tests and bench.py read it)
    if (p.big) GNN_SET_KERNEL_NAME("train_step composite: row-streaming kernels on type ranges (fwd_b6<ADD>, wgrad_b6<XT=%d>)", p.B.XT);
    else if (p.small) GNN_SET_KERNEL_NAME("train_step composite: persistent small-graph kernels");
    else GNN_SET_KERNEL_NAME("train_step composite: general kernels (one launch per layer, type and iteration)");
    if (p.big) TRY(composite_big_setup(ta, p, st));
    if (p.head_fast) {     // the thin-head kernels treat output row m as node m: out_index must BE the identity (checked on the device, read with k)
        gnn::k_not_identity<<<std::min(cdiv(p.M, 256), 1024), 256, 0, st>>>(a.out_index, p.M, p.k_dev + 2);
        LAUNCH_OK();
    }

    // ---- training-mode forward: gated iterations; tape = states, neighbour sums, per-type statistics ---------------------------------
    TRY(launch_converge(nullptr, p.states, nullptr, p.N, p.S, p.ldS, 0, a.state_threshold, p.flags, nullptr, 0.f, st));
    if (p.big) TRY(composite_big_forward(ta, p, st));
    gnn::TileTab tiles;
    gnn::TypeTab yt;
    memset(&tiles, 0, sizeof(tiles)); memset(&yt, 0, sizeof(yt));
    if (p.small) {
        // tiles of <= 64 consecutive positions of ONE type; per type the network, its statistics tape and where its shares go
        int b = 0;
        for (int q = 0; q < p.n_types; ++q) {
            const CType &y = p.ty[q];
            const int pos0 = a.type_offsets[q];
            for (int r0 = 0; r0 < y.count; r0 += 64) tiles.begin[b++] = pos0 + r0;
            yt.wg_begin[q] = p.wg_begin[q]; yt.count[q] = y.count;
            yt.W[q] = y.m->kernel[0]; yt.gamma[q] = y.m->has_bn ? y.m->bn_gamma : nullptr; yt.beta[q] = y.m->bn_beta; yt.stats[q] = y.stats;
            yt.in_s[q] = y.in_dim; yt.off_state[q] = y.off_state; yt.off_agg[q] = y.off_agg; yt.act[q] = y.m->activation[0];
            yt.partW[q] = p.sm_partW[q]; yt.partBN[q] = p.sm_partBN[q];
        }
        tiles.begin[b] = p.N; tiles.n = b;
        yt.n = p.n_types; yt.wg_begin[p.n_types] = p.n_wg; yt.perm = a.type_nodes; yt.inv = p.inv;
        if (b != p.n_wg) return fail("composite tiles: %d != %d", b, p.n_wg);
        // the constant part of every node's first layer (its type's network over its labels and the aggregated component), in position order
        for (int q = 0; q < p.n_types; ++q) {
            const CType &y = p.ty[q];
            if (y.count == 0) continue;
            gnn::ConstSegs cs;
            memset(&cs, 0, sizeof(cs));
            if (y.d_t > 0) { cs.ptr[cs.n] = a.nodes; cs.ld[cs.n] = a.ld_nodes; cs.width[cs.n] = y.d_t; cs.wrow[cs.n] = 0; ++cs.n; }
            if (p.W_comp > 0) { cs.ptr[cs.n] = p.agg_comp; cs.ld[cs.n] = p.W_comp; cs.width[cs.n] = p.W_comp; cs.wrow[cs.n] = y.off_comp; ++cs.n; }
            gnn::k_train_small_const<<<cdiv(y.count * p.SPs, 256), 256, 0, st>>>(y.count, p.SPs, p.S, cs, y.m->kernel[0], y.m->bias[0], y.m->has_bn ? y.m->bn_gamma : nullptr,
                                                                             y.m->bn_beta, y.stats_tpl, y.stats_tpl + y.in_dim, y.m->bn_eps,
                                                                             p.sm_cc + (size_t)a.type_offsets[q] * p.SPs, y.rows);
            LAUNCH_OK();
        }
        HIP_OK(hipMemsetAsync(p.sm_bar, 0, sizeof(unsigned long long) * 4, st));
        gnn::TrainSmallFwd fa;
        memset(&fa, 0, sizeof(fa));
        fa.N = p.N; fa.S = p.SPs; fa.Sw = p.S; fa.K = p.K;
        fa.rowptr = a.adjacency.rowptr; fa.src = a.adjacency.src; fa.w = a.adjacency.w; fa.row_scale = a.adjacency.row_scale;
        fa.states = p.states; fa.agg = p.agg; fa.eps = p.ty[0].m->bn_eps;
        fa.Cc = p.sm_cc; fa.thr = a.state_threshold; fa.flag0 = p.flags;
        fa.bar = p.sm_bar; fa.part = p.sm_part; fa.k_out = p.k_dev; fa.wait_ticks = gnn::wait_ticks();
        if (const char *e = getenv("GNN_DEBUG_FAIL_FWD")) { if (e[0] == '1') fa.wait_ticks = 0; }      // (test hook: every barrier wait of THIS forward launch expires at once)
        switch (p.SPs) {
            case 16: TRY(launch_train_small_fwd_sq<1>(fa, tiles, p.n_wg, a.adjacency.w != nullptr, st, &yt)); break;
            case 32: TRY(launch_train_small_fwd_sq<2>(fa, tiles, p.n_wg, a.adjacency.w != nullptr, st, &yt)); break;
            default: TRY(launch_train_small_fwd_sq<4>(fa, tiles, p.n_wg, a.adjacency.w != nullptr, st, &yt)); break;
        }
    }
    for (int t = 0; t < p.K && !p.small && !p.big; ++t) {
        const int *gate = p.flags + t;
        const float *s_t = p.states + (size_t)t * NS;
        float *s_n = p.states + (size_t)(t + 1) * NS;
        float *agg_t = p.agg + (p.agg_taped ? (size_t)t * NS : 0);
        TRY(launch_aggregate(gate, a.adjacency, s_t, p.S, p.S, agg_t, p.S, st));
        for (int q = 0; q < p.n_types; ++q) {
            CType &y = p.ty[q];
            if (y.count == 0) continue;
            const gnn_mlp_t &ns = *y.m;
            int is_, ia_;
            const int n = ctype_segs(a, p, y, t, segs, &is_, &ia_);
            const float *W0 = ns.kernel[0], *b0 = ns.bias[0], *bn_on_load = nullptr, *centre = nullptr;
            if (ns.has_bn) {
                float *stats = y.stats + (size_t)t * 2 * y.in_dim;
                gnn::Seg dyn[2] = {segs[is_], segs[ia_]};
                TRY(colstats_segs(gate, dyn, 2, y.count, stats, stats + y.in_dim, p.part, st));
                if (ns.units[0] <= 4) {               // thin first layer: the thin-dense kernel wants folded weights (centred: it subtracts the means on load)
                    float *Wf = y.Wf + (size_t)t * y.in_dim * ns.units[0], *bf = y.bf + (size_t)t * ns.units[0];
                    TRY(fold_with_stats(ns, stats, Wf, bf, st, true));
                    W0 = Wf; b0 = bf; centre = stats;
                } else bn_on_load = stats;
            }
            float *hs[GNN_MAX_LAYERS];
            for (int l = 0; l < ns.n_layers; ++l) hs[l] = y.nc.hid[l];
            DropRun drs{&ta.drop_state[q], ta.drop_seed, t, {}};      // (ABI 8: the type's Dropout layers, fresh masks every iteration)
            for (int qq = 0; qq <= ns.n_layers; ++qq) drs.buf[qq] = y.nc.dropbuf[qq];
            TRY(forward_layers(ns, segs, n, y.count, W0, b0, hs, gate, st, nullptr, bn_on_load, centre, &drs));
            // the type's rows of the new state (CompositeGNN.py:229-231: scatter_nd + reduce_sum over the one-hot types).  A closed gate
            // leaves hs stale and the scatter harmless: state t + 1 is never read then.
            TRY(scatter_rows(drop_at(&drs, ns.n_layers) ? drs.buf[ns.n_layers] : hs[ns.n_layers - 1], p.S, y.rows, y.count, p.S, s_n, p.S, st));
        }
        TRY(launch_converge(gate, s_n, s_t, p.N, p.S, p.S, p.S, a.state_threshold, p.flags + t + 1, p.k_dev, (float)(t + 1), st));
    }
    float k_f2[3] = {0.0f, 0.0f, 0.0f};
    HIP_OK(hipMemcpyAsync(k_f2, p.k_dev, 3 * sizeof(float), hipMemcpyDeviceToHost, st));
    HIP_OK(hipStreamSynchronize(st));                                  // the one host synchronisation of the step
    const int k = (int)k_f2[0];
    *ta.k_host = k;
    if (k_f2[2] != 0.0f) p.head_fast = false;      // a permuted / repeated out_index of length n_nodes: the general head (gathers, scatter-add)
    if (k_f2[1] != 0.0f) return fail("a workgroup of the persistent training kernel never arrived at a grid barrier (not resident?)");
    if (k < 0 || k > p.K) return fail("iteration count %d out of range", k);
    if (p.small || p.big) TRY(scatter_rows(p.states + (size_t)k * NS, p.ldS, a.type_nodes, p.N, p.S, ta.state, p.S, st));      // positions -> the caller's node order
    else HIP_OK(hipMemcpyAsync(ta.state, p.states + (size_t)k * NS, sizeof(float) * NS, hipMemcpyDeviceToDevice, st));
    const float *state_k = ta.state;                                   // [N, S] in the caller's node order
    // The moving averages of BatchNormalization (one update per executed call).  On the persistent path they wait until the backward
    // launch has passed its grid barriers and are gated by the step's validity word (train_loop.hpp: a failed backward changes nothing).
    auto moving_state = [&](const int *gate) -> int {
        for (int q = 0; q < p.n_types && k > 0; ++q) {
            const CType &y = p.ty[q];
            if (!y.m->has_bn || y.count == 0) continue;
            gnn::k_bn_moving_multi<<<cdiv(y.in_dim, 256), 256, 0, st>>>(y.stats, 2 * y.in_dim, k, y.in_dim, const_cast<float *>(y.m->bn_mean),
                                                                       const_cast<float *>(y.m->bn_var), ta.bn_momentum, gate);
            LAUNCH_OK();
        }
        return 0;
    };
    if (!p.small) TRY(moving_state(nullptr));

    // ---- output network on the converged state of the masked nodes (CompositeGNN.py:237-239, :270), training mode --------------------
    gnn::Seg osegs[3];
    int nos = 0, n_state_segs = 0;
    int bn_req_off[2] = {0, 0};                 // column offsets of the state segments inside the output net's input
    const int *bn_req_idx[2] = {nullptr, nullptr};
    if (a.focus == GNN_FOCUS_ARC && p.M > 0) {
        k_arc_endpoints<<<cdiv(p.M, 256), 256, 0, st>>>(a.out_index, a.arc_src, a.arc_dst, p.M, p.isrc, p.idst);
        LAUNCH_OK();
        const int *ends[2] = {p.isrc, p.idst};
        for (int e = 0; e < 2; ++e) {
            bn_req_off[n_state_segs] = e * p.S; bn_req_idx[n_state_segs++] = ends[e];
            osegs[nos++] = gnn::Seg{state_k, ends[e], p.S, p.S, e * p.S};
        }
        if (p.A > 0) osegs[nos++] = gnn::Seg{a.arc_labels, a.out_index, a.ld_arcs, p.A, 2 * p.S};
    } else {
        bn_req_off[0] = 0; bn_req_idx[0] = a.out_index; n_state_segs = 1;
        osegs[nos++] = gnn::Seg{state_k, a.out_index, p.S, p.S, 0};
    }
    DropRun dro{&ta.drop_output, ta.drop_seed, 0, {}};
    for (int qq = 0; qq <= no.n_layers; ++qq) dro.buf[qq] = p.co.dropbuf[qq];
    const bool drop_o_last = drop_at(&dro, no.n_layers);          // (a Dropout layer behind the last Dense: the network's output is its dropped-out copy)
    if (drop_o_last && !p.pooled) dro.buf[no.n_layers] = ta.y_pred;
    float *ohs[GNN_MAX_LAYERS];
    for (int l = 0; l < no.n_layers; ++l) ohs[l] = (l == no.n_layers - 1 && !p.pooled && !drop_o_last) ? ta.y_pred : p.co.hid[l];
    float *out_nodes = drop_o_last ? dro.buf[no.n_layers] : ohs[no.n_layers - 1];
    if (p.M > 0) {
        const float *W0 = no.kernel[0], *b0 = no.bias[0];
        if (bn_o) {
            if (p.head_fast) {          // every node a row, in order: one pass over the state (moments around its row 0)
                int grid = 0;
                TRY(rows_stats(nullptr, state_k, p.S, p.S, p.N, p.B.part_y[0], st, &grid));
                gnn::k_stats_finish<<<p.S, 256, 0, st>>>(nullptr, p.B.part_y[0], grid, p.S, 1.0f / (float)p.N, p.stats_o, p.stats_o + no.in_dim, state_k);
                LAUNCH_OK();
            } else
            TRY(colstats_segs(nullptr, osegs, nos, p.M, p.stats_o, p.stats_o + no.in_dim, p.part, st));
            TRY(fold_with_stats(no, p.stats_o, p.Wf_o, p.bf_o, st, true));
            if (!p.small) {
                gnn::k_bn_moving_multi<<<cdiv(no.in_dim, 256), 256, 0, st>>>(p.stats_o, 2 * no.in_dim, 1, no.in_dim, const_cast<float *>(no.bn_mean),
                                                                            const_cast<float *>(no.bn_var), ta.bn_momentum);
                LAUNCH_OK();
            }
            W0 = p.Wf_o; b0 = p.bf_o;
        }
        TRY(forward_layers(no, osegs, nos, p.M, W0, b0, ohs, nullptr, st, nullptr, nullptr, bn_o ? p.stats_o : nullptr, &dro));
    }
    if (p.pooled) TRY(launch_aggregate(nullptr, a.nodegraph, out_nodes, p.T, p.T, ta.y_pred, p.T, st));
    gnn::k_loss_grad<<<cdiv(std::max(p.R, 1), 256), 256, 0, st>>>(ta.loss_kind, ta.targets, ta.y_pred, ta.sample_weight, p.R, p.T, p.dpred, p.loss_rows);
    LAUNCH_OK();
    if (p.R > 65536) {
        gnn::k_sum_partials<<<256, 256, 0, st>>>(p.loss_rows, p.R, p.loss_part);
        LAUNCH_OK();
        gnn::k_sum_scale<<<1, 256, 0, st>>>(p.loss_part, 256, 1.0f / (float)std::max(p.R, 1), ta.loss);
    } else gnn::k_sum_scale<<<1, 256, 0, st>>>(p.loss_rows, p.R, 1.0f / (float)std::max(p.R, 1), ta.loss);
    LAUNCH_OK();
    float *G_out = p.dpred;
    if (p.pooled) { TRY(launch_aggregate(nullptr, ta.nodegraph_by_source, p.dpred, p.T, p.T, p.G_out, p.T, st)); G_out = p.G_out; }

    // ---- backward: output network, then the k iterations ------------------------------------------------------------------------------
    if (p.head_fast) {          // the head's parameter gradients, d loss / d state_k written straight into the state gradient (every row is an output row)
        TrainPlan hp;
        memset(&hp, 0, sizeof(hp));
        hp.M = p.M; hp.N = p.N; hp.S = p.S; hp.ldS = p.S; hp.T = p.T; hp.with_labels = false; hp.L = 0;
        hp.part_h = p.part_h; hp.co = p.co; hp.G_state = p.G_state;
        TRY(head_backward(hp, a, state_k, out_nodes, G_out, bn_o ? p.stats_o : nullptr, st, -1));
    } else {
    HIP_OK(hipMemsetAsync(p.G_state, 0, sizeof(float) * (size_t)p.N * p.S, st));
    if (p.M > 0) {
        TRY(net_backward(p.co, osegs, nos, ohs, G_out, p.T, p.M, bn_o ? p.stats_o : nullptr, false, p.dx_o_all, no.in_dim, p.part, st, -1, 0, &dro));
        gnn::BnGradReq rq[2];
        for (int i = 0; i < n_state_segs; ++i) rq[i] = gnn::BnGradReq{p.dx_o_all + bn_req_off[i], no.in_dim, state_k, p.S, bn_req_idx[i], p.S, bn_req_off[i]};
        TRY(bn_input_grads(no, p.co, p.stats_o, rq, n_state_segs, p.M, st));
        for (int i = 0; i < n_state_segs; ++i) {     // (an arc's two end nodes, one after the other: the additions to a node's row stay ordered)
            gnn::k_scatter_add_rows<<<std::min(cdiv((long)p.M * p.S, 256), 256 * 16), 256, 0, st>>>(p.dx_o_all + bn_req_off[i], no.in_dim, bn_req_idx[i], p.M, p.S, p.G_state, p.S);
            LAUNCH_OK();
        }
    } else TRY(zero_grads(no, ta.grad_output, st));
    }
    for (int q = 0; q < p.n_types; ++q)
        if (k == 0 || p.ty[q].count == 0) TRY(zero_grads(*p.ty[q].m, p.ty[q].g, st));
    if (p.small && k > 0) {
        // the k iterations of back-propagation in one persistent launch; every tile leaves its share of ITS type's gradients
        gnn::TypeConsts yc;
        memset(&yc, 0, sizeof(yc));
        for (int q = 0; q < p.n_types; ++q) {
            const CType &y = p.ty[q];
            gnn::ConstSegs &cs = yc.cs[q];
            if (y.d_t > 0) { cs.ptr[cs.n] = a.nodes; cs.ld[cs.n] = a.ld_nodes; cs.width[cs.n] = y.d_t; cs.wrow[cs.n] = 0; ++cs.n; }
            if (p.W_comp > 0) { cs.ptr[cs.n] = p.agg_comp; cs.ld[cs.n] = p.W_comp; cs.width[cs.n] = p.W_comp; cs.wrow[cs.n] = y.off_comp; ++cs.n; }
        }
        gnn::TrainSmallBwd ba;
        memset(&ba, 0, sizeof(ba));
        const gnn_csr_t &cs_ = ta.adjacency_by_source;
        const bool unit_w = !a.adjacency.w;       // entries depend on the destination only: scale the agg-half once per row, walk unit weights
        ba.N = p.N; ba.S = p.SPs; ba.Sw = p.S; ba.k = k;
        ba.rowptr_s = cs_.rowptr; ba.src_s = cs_.src; ba.w_s = unit_w ? nullptr : cs_.w; ba.row_scale_s = unit_w ? nullptr : cs_.row_scale;
        ba.row_scale = unit_w ? a.adjacency.row_scale : nullptr;
        ba.states = p.states; ba.agg = p.agg; ba.eps = p.ty[0].m->bn_eps;
        ba.G0 = p.G_state; ba.dxa = p.sm_dxa; ba.bar = p.sm_bar + 2; ba.part = p.sm_part;
        ba.inv_n = 1.0f / (float)p.N; ba.err = p.k_dev; ba.wait_ticks = gnn::wait_ticks();
        if (const char *e = getenv("GNN_DEBUG_FAIL_BWD")) { if (e[0] == '1') ba.wait_ticks = 0; }
        switch (p.SPs) {
            case 16: TRY(launch_train_small_bwd_sq<1>(ba, tiles, p.n_wg, ba.w_s != nullptr, st, &yt, &yc)); break;
            case 32: TRY(launch_train_small_bwd_sq<2>(ba, tiles, p.n_wg, ba.w_s != nullptr, st, &yt, &yc)); break;
            default: TRY(launch_train_small_bwd_sq<4>(ba, tiles, p.n_wg, ba.w_s != nullptr, st, &yt, &yc)); break;
        }
        for (int q = 0; q < p.n_types; ++q) {          // every tile's [kernel | bias] (and [d gamma | d beta]) share, summed in tile order
            const CType &y = p.ty[q];
            const int tiles_q = p.wg_begin[q + 1] - p.wg_begin[q];
            if (tiles_q == 0) continue;
            const int n = (y.in_dim + 1) * p.S;
            gnn::k_reduce_partials<<<cdiv(n, 64), 256, 0, st>>>(p.sm_partW[q], tiles_q, n, y.g.dkernel[0], 0, 1.0f, y.in_dim * p.S, y.g.dbias[0]);
            LAUNCH_OK();
            if (y.m->has_bn) {
                gnn::k_reduce_partials<<<cdiv(2 * y.in_dim, 64), 256, 0, st>>>(p.sm_partBN[q], tiles_q, 2 * y.in_dim, y.g.dgamma, 0, 1.0f, y.in_dim, y.g.dbeta);
                LAUNCH_OK();
            }
        }
    }
    if (p.big) TRY(composite_big_backward(ta, p, k, st));
    for (int t = k - 1; t >= 0 && !p.small && !p.big; --t) {
        const float *s_t = p.states + (size_t)t * NS;
        const float *s_n = p.states + (size_t)(t + 1) * NS;
        float *agg_t = p.agg + (p.agg_taped ? (size_t)t * NS : 0);
        if (!p.agg_taped) TRY(launch_aggregate(nullptr, a.adjacency, s_t, p.S, p.S, p.agg, p.S, st));
        for (int q = 0; q < p.n_types; ++q) {
            CType &y = p.ty[q];
            if (y.count == 0) continue;
            const gnn_mlp_t &ns = *y.m;
            int is_, ia_;
            const int n = ctype_segs(a, p, y, t, segs, &is_, &ia_);
            const float *stats = ns.has_bn ? y.stats + (size_t)t * 2 * y.in_dim : nullptr;
            float *hs[GNN_MAX_LAYERS];
            for (int l = 0; l < ns.n_layers; ++l) hs[l] = y.nc.hid[l];
            DropRun drs{&ta.drop_state[q], ta.drop_seed, t, {}};      // (iteration t's masks again: the same keys)
            for (int qq = 0; qq <= ns.n_layers; ++qq) drs.buf[qq] = y.nc.dropbuf[qq];
            const bool drop_any = ta.drop_state[q].n > 0;
            if (ns.n_layers > 1 || drop_any) {     // hidden activations are not on the tape: recompute them (with Dropout layers: the last layer's too)
                const bool folded = ns.has_bn && ns.units[0] <= 4;
                const float *W0 = folded ? y.Wf + (size_t)t * y.in_dim * ns.units[0] : ns.kernel[0], *b0 = folded ? y.bf + (size_t)t * ns.units[0] : ns.bias[0];
                gnn_mlp_t head = ns; head.n_layers = drop_any ? ns.n_layers : ns.n_layers - 1;
                TRY(forward_layers(head, segs, n, y.count, W0, b0, hs, nullptr, st, nullptr, (ns.has_bn && !folded) ? stats : nullptr, folded ? stats : nullptr, &drs, true));
            }
            if (!drop_any) TRY(gather_rows(s_n, p.S, y.rows, y.count, p.S, hs[ns.n_layers - 1], p.S, st));      // the last layer's output: the type's rows of state t + 1
            TRY(gather_rows(p.G_state, p.S, y.rows, y.count, p.S, y.Gc, p.S, st));
            TRY(net_backward(y.nc, segs, n, hs, y.Gc, p.S, y.count, stats, t != k - 1, y.dx, 2 * p.S, p.part, st, -1, y.off_state, &drs));
            gnn::BnGradReq rq[2] = {gnn::BnGradReq{y.dx, 2 * p.S, s_t, p.S, y.rows, p.S, y.off_state},
                                    gnn::BnGradReq{y.dx + p.S, 2 * p.S, agg_t, p.S, y.rows, p.S, y.off_agg}};
            TRY(bn_input_grads(ns, y.nc, stats, rq, 2, y.count, st));
            TRY(scatter_rows(y.dx, 2 * p.S, y.rows, y.count, 2 * p.S, p.dx_full, 2 * p.S, st));
        }
        {   // G_state = d state (own) + Adj . d agg   (arcs walked by source)
            const gnn_csr_t &c = ta.adjacency_by_source;
            int G = 4;
            while (G < p.S && G < 64) G <<= 1;
            const int groups = 256 / G, grid = std::min(cdiv(p.N, groups), 256 * 16);
#define AGGA(GG) gnn::k_aggregate_add<GG><<<grid, 256, 0, st>>>(c.n_dst, c.rowptr, c.src, c.w, c.row_scale, p.dx_full + p.S, 2 * p.S, p.S, \
                                                                 p.dx_full, 2 * p.S, p.G_state, p.S)
            switch (G) { case 4: AGGA(4); break; case 8: AGGA(8); break; case 16: AGGA(16); break; case 32: AGGA(32); break; default: AGGA(64); break; }
#undef AGGA
            LAUNCH_OK();
        }
    }
    if (ta.average_st_grads && k > 0)
        for (int q = 0; q < p.n_types; ++q) TRY(scale_grads(*p.ty[q].m, p.ty[q].g, 1.0f / (float)k, st));
    gnn::k_grads_ok<<<1, 1, 0, st>>>(p.small ? p.k_dev : nullptr, p.grads_ok);      // (only the persistent backward launch can fail)
    LAUNCH_OK();
    if (p.small) {
        TRY(moving_state(p.grads_ok));
        if (bn_o && p.M > 0) {
            gnn::k_bn_moving_multi<<<cdiv(no.in_dim, 256), 256, 0, st>>>(p.stats_o, 2 * no.in_dim, 1, no.in_dim, const_cast<float *>(no.bn_mean),
                                                                        const_cast<float *>(no.bn_var), ta.bn_momentum, p.grads_ok);
            LAUNCH_OK();
        }
    }
    return 0;
}

}  // namespace
