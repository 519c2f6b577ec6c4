// gnn_train_step: one whole training step of a homogeneous model inside the library (reference GNN/Models/GNN.py:277-306:
// `self(x, training=True)`, `compiled_loss`, `tape.gradient` through the unrolled loop, `dwbS / k`).  Included at the end of
// gnnloop.hip, after train_api.hpp (shares the launchers).  Declared in include/gnnloop.h.
//
// Same algebra as gnnkeras_amd/Models/training.py (which stays the general path: composite models, LGNN label gradients):
// the tape is the k + 1 state matrices plus per-iteration BatchNormalization statistics and folded first-layer weights;
// aggregates and hidden activations are recomputed in the backward sweep; the first layer never materialises the
// N x in_dim concatenation (P = X^T dZ per segment, dW = a (.) P + c q^T, BN input-gradient moments from W, P, q).
// What moves into the library is the ORCHESTRATION: ~14 launches per iteration pair instead of ~36, no Python between them.
#pragma once
#include "kernels_train.hpp"
#include "kernels_train_big.hpp"
#include "kernels_train_small.hpp"

namespace {

struct NetCtx {
    const gnn_mlp_t *m;
    const gnn_mlp_grads_t *g;
    float *Wt[GNN_MAX_LAYERS];      // kernel transposes [units[l]][fan_in]
    float *P, *q, *m1, *m2;         // first-layer scratch: [in_dim x H1], [H1], [in_dim], [in_dim]
    float *G[2];                    // gradient ping-pong [rows x max units]
    float *hid[GNN_MAX_LAYERS];     // layer outputs [rows x units[l]] (the last one may alias a caller buffer)
    float *dropbuf[GNN_MAX_LAYERS + 1];   // [q]: what the Dense at position q consumes when Dropout layers sit there ([rows x units[q - 1]]); else NULL
};

// ---- Dropout / AlphaDropout layers of a network call (ABI 8: gnn_dropout_spec_t; reference MLP.py:60-66) -------------------------------------
// The masks are the counter hash of gnn_dropout keyed by (step seed, network, call, layer): nothing is stored, the backward pass of a call
// regenerates them.  Positions 1 .. n_layers (behind a Dense layer's activation); position 0 is refused by the plans.
inline unsigned lowbias32_host(unsigned h) { h ^= h >> 16; h *= 0x7feb352du; h ^= h >> 15; h *= 0x846ca68bu; h ^= h >> 16; return h; }
inline unsigned drop_key(unsigned seed, unsigned net_id, unsigned call, unsigned index) {
    unsigned h = 0x9E3779B9u;
    const unsigned v[4] = {seed, net_id, call, index};
    for (int i = 0; i < 4; ++i) h = lowbias32_host(h ^ v[i]);
    return h;
}
struct DropRun {
    const gnn_dropout_spec_t *d;           // NULL or n == 0: no Dropout layers
    unsigned seed; int call;
    float *buf[GNN_MAX_LAYERS + 1];        // [q]: destination of the dropped-out copy for position q (forward)
};
inline bool drop_at(const DropRun *r, int q) {
    if (!r || !r->d) return false;
    for (int i = 0; i < r->d->n; ++i) if (r->d->pos[i] == q) return true;
    return false;
}
// every Dropout layer at position q: forward x -> y (list order; later layers in place on y), backward in place on y = x (reverse order)
int run_dropout(const DropRun &r, int q, const float *x, int ldx, float *y, int ldy, int M, int H, bool backward, hipStream_t st) {
    const gnn_dropout_spec_t &d = *r.d;
    const float *src = x;
    for (int ii = 0; ii < d.n; ++ii) {
        const int i = backward ? d.n - 1 - ii : ii;
        if (d.pos[i] != q) continue;
        TRY(gnn_dropout(src, src == x ? ldx : ldy, y, ldy, M, H, d.rate[i], drop_key(r.seed, (unsigned)d.net_id, (unsigned)r.call, (unsigned)d.index[i]), d.alpha,
                        backward ? 1 : 0, (void *)st));
        src = y;
    }
    return 0;
}
int check_dropout(const gnn_dropout_spec_t &d, const gnn_mlp_t &m, const char *name) {
    if (d.n < 0 || d.n > GNN_MAX_DROPOUT) return fail("%s: %d dropout layers (0 .. %d)", name, d.n, GNN_MAX_DROPOUT);
    for (int i = 0; i < d.n; ++i) {
        if (d.pos[i] == 0) return fail("%s: Dropout in front of the first Dense layer (position 0): such networks train through the building blocks", name);
        if (d.pos[i] < 0 || d.pos[i] > m.n_layers || (i > 0 && d.pos[i] < d.pos[i - 1])) return fail("%s: dropout positions must ascend within [1, n_layers]", name);
        if (!(d.rate[i] > 0.0f && d.rate[i] < 1.0f)) return fail("%s: dropout rate %g outside (0, 1)", name, (double)d.rate[i]);
    }
    return 0;
}

// Bytes of neighbour sums the tape may keep (one N x S matrix per iteration) instead of recomputing each in the backward sweep.
// Sized for a 288 GB device: 32 GiB by default (C4 with 50 iterations is 12.8 GB), GNN_TRAIN_TAPE_MB overrides.
inline size_t agg_tape_budget() {
    static size_t v = 0;
    if (v == 0) { const char *e = getenv("GNN_TRAIN_TAPE_MB"); v = (e ? (size_t)atol(e) : (size_t)32 * 1024) << 20; v = std::max<size_t>(v, 1); }
    return v;
}

struct TrainPlan {
    int N, E, S, L, A, d, M, R, T, K, G;
    bool pooled, with_labels, agg_taped;      // agg_taped: every iteration's neighbour sum is kept (small graphs) instead of recomputed
    int in_s, in_o, H1s, H1o, kdx_s, off_agg;
    int *grads_ok;               // FIRST word of the tape: validity of the step's gradients (gnn_train_args_t::grads_ok_dev)
    int *flags; float *k_dev;
    float *states, *agg, *agg_arcs, *agg_nodes;
    float *stats_s, *stats_tpl, *Wf_s, *bf_s;
    float *stats_o, *Wf_o, *bf_o;
    float *dx_s_all, *dx_o_all, *G_state, *G_out, *dpred, *loss_rows;
    int *isrc, *idst;
    float *part; size_t part_floats;
    NetCtx cs, co;
    // large graphs (kernels_train_big.hpp): constant inputs packed 32 per node, statistics partials of the two producers
    bool big; int Kc; gnn::ConstCols cc;
    bool head_fast;              // large graphs: thin output head (one Dense of <= 4 units over EVERY node) on the row-streaming head kernels
    float *part_h;
    float *xc, *part_a, *part_y, *loss_part, *part_w;
    // small graphs (kernels_train_small.hpp): the forward / backward iterations as one persistent launch each
    bool small, tiled; int n_wg;
    int SPs, ldS;                // small path: the padded state width the persistent kernels run (16 / 32 / 64) and the tape's row stride
    float *sm_cc, *sm_part, *sm_partW, *sm_partBN; unsigned long long *sm_bar;
    size_t bytes;
};

// From this many nodes on a training iteration is bandwidth work and runs the row-streaming kernels (GNN_TRAIN_BIG_MIN_NODES).
inline int train_big_min_nodes() {
    int v = -1;      // (read at every call: tests switch it inside one process)
    { const char *e = getenv("GNN_TRAIN_BIG_MIN_NODES"); v = e ? atoi(e) : 32768; }
    return v;
}
constexpr int BIG_AGG_BLOCKS = 4096, BIG_FWD_BLOCKS = 1024, BIG_WGRAD_BLOCKS = 512, BIG_HEAD_BLOCKS = 2048;
inline bool train_wgrad_enabled() {       // GNN_TRAIN_WGRAD=0: the round-2 weight-gradient kernels (k_act_grad + k_dense_grad_allk) at large M too
    int v = -1;      // (read at every call: tests switch it inside one process)
    { const char *e = getenv("GNN_TRAIN_WGRAD"); v = (e && e[0] == '0') ? 0 : 1; }
    return v != 0;
}
// The persistent small-graph kernels need every workgroup resident: one 64-node tile per CU (GNN_TRAIN_SMALL=0 switches them off).
inline bool train_small_enabled() {
    int v = -1;      // (read at every call: tests switch it inside one process)
    { const char *e = getenv("GNN_TRAIN_SMALL"); v = (e && e[0] == '0') ? 0 : 1; }
    return v != 0;
}

int max_units_of(const gnn_mlp_t &m) { int h = 1; for (int i = 0; i < m.n_layers; ++i) h = std::max(h, (int)m.units[i]); return h; }

size_t grad_part_floats(int K, int H, int M) { int nc; rows_per_chunk_for(std::max(M, 1), &nc); return (size_t)nc * ((size_t)K * H + H); }

void carve_net(Carver &c, NetCtx &x, const gnn_mlp_t &m, int rows, size_t &part_floats, const gnn_dropout_spec_t *drop = nullptr) {
    int fan_in = m.in_dim;
    DropRun dr{drop, 0u, 0, {}};
    for (int q = 0; q <= GNN_MAX_LAYERS; ++q) x.dropbuf[q] = nullptr;
    for (int q = 1; q <= m.n_layers; ++q)
        if (drop_at(&dr, q)) x.dropbuf[q] = c.take<float>((size_t)std::max(rows, 1) * m.units[q - 1]);
    for (int l = 0; l < m.n_layers; ++l) {
        x.Wt[l] = c.take<float>((size_t)fan_in * m.units[l]);
        x.hid[l] = c.take<float>((size_t)std::max(rows, 1) * m.units[l]);
        part_floats = std::max(part_floats, grad_part_floats(fan_in, m.units[l], rows));
        fan_in = m.units[l];
    }
    x.P = c.take<float>((size_t)m.in_dim * m.units[0]); x.q = c.take<float>(m.units[0]);
    x.m1 = c.take<float>(m.in_dim); x.m2 = c.take<float>(m.in_dim);
    const int mu = std::max(max_units_of(m), 1);
    x.G[0] = c.take<float>((size_t)std::max(rows, 1) * mu); x.G[1] = c.take<float>((size_t)std::max(rows, 1) * mu);
}

int make_train_plan(const gnn_train_args_t &ta, void *ws, TrainPlan &p) {
    const gnn_loop_args_t &a = ta.loop;
    memset(&p, 0, sizeof(p));
    if (a.abi_version != GNN_ABI_VERSION) return fail("abi_version %d != %d", a.abi_version, GNN_ABI_VERSION);
    if (a.composite) return fail("gnn_train_step: homogeneous models only");
    if (a.n_nodes < 1) return fail("gnn_train_step: empty graph");
    if (a.n_heavy_segments > 0) { /* the training kernels walk the plain adjacency */ }
    p.N = a.n_nodes; p.E = a.n_arcs; p.L = a.dim_node_label; p.A = a.dim_arc_label; p.d = a.state_dim;
    p.S = a.state_dim > 0 ? a.state_dim : a.dim_node_label;
    p.K = a.max_iteration; p.M = a.n_out;
    TRY(check_mlp(a.net_state[0], "net_state", ws != nullptr));
    TRY(check_mlp(a.net_output, "net_output", ws != nullptr));
    const gnn_mlp_t &ns = a.net_state[0], &no = a.net_output;
    TRY(check_dropout(ta.drop_state[0], ns, "net_state"));
    TRY(check_dropout(ta.drop_output, no, "net_output"));
    const bool drop_s = ta.drop_state[0].n > 0, drop_o = ta.drop_output.n > 0;      // (a state network with Dropout: the general kernels)
    p.with_labels = a.state_dim > 0;
    p.in_s = ns.in_dim; p.in_o = no.in_dim; p.H1s = ns.units[0]; p.H1o = no.units[0];
    const int expect_s = a.state_dim > 0 ? 2 * p.S + 2 * p.L + p.A : 2 * p.S + p.A;
    if (ns.in_dim != expect_s) return fail("net_state.in_dim %d != %d expected from the graph dims", ns.in_dim, expect_s);
    if (ns.units[ns.n_layers - 1] != p.S) return fail("net_state output width %d != state width %d", ns.units[ns.n_layers - 1], p.S);
    const int node_part = p.with_labels ? p.S + p.L : p.S;
    const int expect_o = a.focus == GNN_FOCUS_ARC ? 2 * node_part + p.A : node_part;
    if (no.in_dim != expect_o) return fail("net_output.in_dim %d != %d expected for this focus", no.in_dim, expect_o);
    p.T = no.units[no.n_layers - 1];
    p.pooled = a.focus == GNN_FOCUS_GRAPH;
    p.G = p.pooled ? a.nodegraph.n_dst : 0;
    p.R = p.pooled ? p.G : p.M;
    p.off_agg = p.with_labels ? p.S + p.L : p.S;          // first-layer row (= BN column) of the aggregated-state segment
    p.kdx_s = 2 * p.S;                                     // d loss / d [state | agg] only, 16-B aligned halves: the label columns between them are never needed

    // ---- which path: the large-graph kernels, the persistent small-graph kernels (state width padded to 16 / 32 / 64 inside the tape:
    // the starter configuration's 14 label columns run the 16-wide kernels), or one launch per layer and iteration
    p.Kc = (p.with_labels ? 2 * p.L : 0) + p.A;
    p.big = p.N >= train_big_min_nodes() && ns.n_layers == 1 && ns.units[0] == p.S && (p.S == 16 || p.S == 32 || p.S == 64) &&
            p.Kc <= 32 && ns.activation[0] != GNN_ACT_SOFTMAX && (size_t)p.N * p.S * 4 < ((size_t)1 << 32) && p.K > 0 && !drop_s;
    // every node is an output row (out_index ascending and n_out == n_nodes: the identity), one thin Dense: kernels_train_big.hpp
    p.head_fast = p.big && no.n_layers == 1 && no.units[0] <= 4 && a.focus != GNN_FOCUS_ARC && p.M == p.N && p.S % 4 == 0 &&
                  p.S / 4 + ((p.with_labels ? p.L : 0) + 3) / 4 <= 32 && !drop_o;
    p.SPs = p.S <= 16 ? 16 : p.S <= 32 ? 32 : 64;
    p.tiled = ta.n_tiles > 0 && ta.tile_node_begin != nullptr;
    p.n_wg = p.tiled ? ta.n_tiles : cdiv(p.N, 64);
    if (p.tiled && p.n_wg > 256) { p.tiled = false; p.n_wg = cdiv(p.N, 64); }
    p.small = !p.big && !drop_s && train_small_enabled() && ns.n_layers == 1 && ns.units[0] == p.S && p.S <= 64 && p.Kc <= 32 &&
              ns.activation[0] != GNN_ACT_SOFTMAX && p.K > 0 && p.n_wg <= device_cus() && p.N < train_big_min_nodes() &&
              (size_t)std::max(p.K, 1) * p.N * p.SPs * sizeof(float) <= agg_tape_budget();
    p.ldS = p.small ? p.SPs : p.S;

    Carver c(ws);
    p.grads_ok = c.take<int>(4);                           // (offset 0 whatever the plan: the next call on this tape reads it back)
    p.flags = c.take<int>(p.K + 8);
    p.k_dev = c.take<float>(4);
    p.states = c.take<float>((size_t)(p.K + 1) * p.N * p.ldS);
    p.agg_taped = (size_t)std::max(p.K, 1) * p.N * p.ldS * sizeof(float) <= agg_tape_budget();
    p.agg = c.take<float>((size_t)(p.agg_taped ? std::max(p.K, 1) : 1) * p.N * p.ldS);
    p.agg_arcs = c.take<float>((size_t)p.N * std::max(p.A, 1));
    p.agg_nodes = c.take<float>((size_t)p.N * std::max(p.L, 1));
    p.stats_s = c.take<float>((size_t)(std::max(p.K, 1) + 1) * 2 * p.in_s);      // (+ 1: the statistics of the LAST state, for the output head's BatchNormalization)
    p.stats_tpl = c.take<float>(2 * (size_t)p.in_s);
    p.Wf_s = c.take<float>((size_t)std::max(p.K, 1) * p.in_s * p.H1s);
    p.bf_s = c.take<float>((size_t)std::max(p.K, 1) * p.H1s);
    p.stats_o = c.take<float>(2 * (size_t)p.in_o);
    p.Wf_o = c.take<float>((size_t)p.in_o * p.H1o); p.bf_o = c.take<float>(p.H1o);
    p.dx_s_all = c.take<float>((size_t)p.N * std::max(p.kdx_s, p.small ? p.SPs : 0));      // (also the persistent backward kernel's [N, SPs] exchange rows)
    p.dx_o_all = c.take<float>((size_t)std::max(p.M, 1) * p.in_o);
    p.G_state = c.take<float>((size_t)p.N * p.S);
    p.G_out = c.take<float>((size_t)std::max(p.M, 1) * p.T);
    p.dpred = c.take<float>((size_t)std::max(p.R, 1) * p.T);
    p.loss_rows = c.take<float>(std::max(p.R, 1));
    p.isrc = c.take<int>(std::max(p.M, 1)); p.idst = c.take<int>(std::max(p.M, 1));
    p.part_floats = 0;
    carve_net(c, p.cs, ns, p.N, p.part_floats, &ta.drop_state[0]);
    carve_net(c, p.co, no, p.M, p.part_floats, &ta.drop_output);
    {   // column-statistics partials of the large-batch path: (chunks + 1) x widest segment
        int nc; rows_per_chunk_for(std::max(p.N, 1), &nc);
        p.part_floats = std::max(p.part_floats, (size_t)(nc + 1) * std::max(p.in_s, p.in_o));
    }
    p.part = c.take<float>(p.part_floats);
    memset(&p.cc, 0, sizeof(p.cc));
    if (p.with_labels) {
        p.cc.n = p.A > 0 ? 3 : 2;
        p.cc.width[0] = p.L; p.cc.wrow[0] = p.S;
        p.cc.width[1] = p.L; p.cc.wrow[1] = 2 * p.S + p.L;
        p.cc.width[2] = p.A; p.cc.wrow[2] = 2 * p.S + 2 * p.L;
    } else {
        p.cc.n = p.A > 0 ? 1 : 0;
        p.cc.width[0] = p.A; p.cc.wrow[0] = 2 * p.S;
    }
    p.xc = c.take<float>(p.big && p.Kc > 0 ? (size_t)p.N * 32 : 0);
    p.part_a = c.take<float>(p.big ? (size_t)BIG_AGG_BLOCKS * 2 * p.S : 0);
    p.part_y = c.take<float>(p.big ? (size_t)BIG_FWD_BLOCKS * 2 * std::max(p.S, 32) : 0);      // (also the one-pass statistics of the 32-wide constants line)
    p.loss_part = c.take<float>(256);
    p.part_w = c.take<float>(p.big ? (size_t)BIG_WGRAD_BLOCKS * ((size_t)p.in_s * p.S + p.S) : 0);      // k_train_wgrad: one partial per workgroup
    p.part_h = c.take<float>(p.head_fast ? (size_t)BIG_HEAD_BLOCKS * ((size_t)p.in_o * p.H1o + p.H1o) : 0);             // k_head_wgrad: one partial per workgroup
    p.sm_cc = c.take<float>(p.small ? (size_t)p.N * p.SPs : 0);
    p.sm_part = c.take<float>(p.small ? (size_t)2 * p.n_wg * 8 * p.SPs : 0);
    p.sm_partW = c.take<float>(p.small ? (size_t)p.n_wg * (p.in_s + 1) * p.S : 0);
    p.sm_partBN = c.take<float>(p.small ? (size_t)p.n_wg * 2 * p.in_s : 0);
    p.sm_bar = c.take<unsigned long long>(p.small ? 4 : 0);
    p.cs.m = &ns; p.cs.g = &ta.grad_state; p.co.m = &no; p.co.g = &ta.grad_output;
    p.bytes = (c.off + 255) & ~(size_t)255;
    return 0;
}

// ---- column statistics of the listed segments into mean / var (positions seg.wrow) --------------------------------------
int colstats_segs(const int *gate, const gnn::Seg *segs, int n, int M, float *mean, float *var, float *part, hipStream_t st) {
    if (n == 0) return 0;
    if (M <= 8192) {
        gnn::StatSegs ss;
        memset(&ss, 0, sizeof(ss));
        ss.n = n; ss.col_begin[0] = 0;
        for (int s = 0; s < n; ++s) { ss.seg[s] = segs[s]; ss.col_begin[s + 1] = ss.col_begin[s] + segs[s].width; }
        gnn::k_colstats_segs_small<<<ss.col_begin[n], 256, 0, st>>>(gate, ss, M, mean, var);
        LAUNCH_OK();
        return 0;
    }
    int n_chunks;
    const int rpc = rows_per_chunk_for(M, &n_chunks);
    for (int s = 0; s < n; ++s) {       // large batches: two passes per segment (launch count does not matter there)
        const gnn::Seg &g = segs[s];
        gnn::k_colstats_partial<<<n_chunks, 256, 0, st>>>(g.ptr, g.ld, g.rowidx, g.width, M, rpc, nullptr, part);
        LAUNCH_OK();
        gnn::k_reduce_partials<<<cdiv(g.width, 64), 256, 0, st>>>(part, n_chunks, g.width, mean + g.wrow, 0, 1.0f / (float)M, g.width, nullptr);
        LAUNCH_OK();
        gnn::k_colstats_partial<<<n_chunks, 256, 0, st>>>(g.ptr, g.ld, g.rowidx, g.width, M, rpc, mean + g.wrow, part);
        LAUNCH_OK();
        gnn::k_reduce_partials<<<cdiv(g.width, 64), 256, 0, st>>>(part, n_chunks, g.width, var + g.wrow, 0, 1.0f / (float)M, g.width, nullptr);
        LAUNCH_OK();
    }
    return 0;
}

int fold_with_stats(const gnn_mlp_t &m, const float *stats, float *Wf, float *bf, hipStream_t st, bool centred = false) {
    FoldList fl;
    gnn::FoldJob &j = fl.fa.job[fl.fa.n_jobs++];
    j.centred = centred ? 1 : 0;
    j.W = m.kernel[0]; j.b = m.bias[0]; j.K = m.in_dim; j.H = m.units[0];
    j.gamma = m.bn_gamma; j.beta = m.bn_beta; j.mean = stats; j.var = stats + m.in_dim; j.eps = m.bn_eps;
    j.Wf = Wf; j.bf = bf; j.blk_begin = 0;
    fl.blocks = j.H;
    return launch_fold_list(fl, st);
}

// layers of `m` over a virtual concatenation; first layer with (W0, b0); outputs into hs[l] (ld = units[l])
struct PredFuse { const float *old; int ld; float thr; int *flag; float *k_out; float k_val; bool fused; };

// `bn_stats` (mean | var of the first layer's input columns): the training-mode BatchNormalization is applied to the inputs as
// the first layer stages them (k_segdense), with (W0, b0) the raw kernel / bias; NULL: (W0, b0) are used as they are.
// `center` (the column means, by weight row): (W0, b0) is a CENTRED fold (fold_with_stats(.., true)) and the first layer subtracts
// the means from its inputs as it stages them (layers whose kernel wants folded weights: the thin-output kernel).
// `drop`: the network's Dropout layers (positions >= 1): hs[l] stay the layers' own outputs, what the next Dense consumes is drop->buf[l + 1];
// the network's output is then drop->buf[n_layers] when a layer sits behind the last Dense (`skip_last_drop`: a backward recompute, which
// does not need it).
int forward_layers(const gnn_mlp_t &m, const gnn::Seg *segs, int nseg, int M, const float *W0, const float *b0, float *const *hs,
                   const int *gate, hipStream_t st, PredFuse *pred = nullptr, const float *bn_stats = nullptr, const float *center = nullptr,
                   const DropRun *drop = nullptr, bool skip_last_drop = false) {
    for (int l = 0; l < m.n_layers; ++l) {
        gnn::SegDenseArgs a;
        memset(&a, 0, sizeof(a));
        a.gate = gate; a.M = M; a.H = m.units[l];
        if (l == 0) {
            a.nseg = nseg;
            for (int s = 0; s < nseg; ++s) a.seg[s] = segs[s];
            a.W = W0; a.bias = b0;
            if (bn_stats) { a.in_gamma = m.bn_gamma; a.in_beta = m.bn_beta; a.in_mean = bn_stats; a.in_var = bn_stats + m.in_dim; a.in_eps = m.bn_eps; }
            else a.in_center = center;
        } else {
            const float *src = hs[l - 1];
            if (drop_at(drop, l)) {
                TRY(run_dropout(*drop, l, hs[l - 1], m.units[l - 1], drop->buf[l], m.units[l - 1], M, m.units[l - 1], false, st));
                src = drop->buf[l];
            }
            a.nseg = 1;
            a.seg[0] = gnn::Seg{src, nullptr, (int)m.units[l - 1], (int)m.units[l - 1], 0};
            a.W = m.kernel[l]; a.bias = m.bias[l];
        }
        a.ldw = a.H;
        const bool thin_softmax = m.activation[l] == GNN_ACT_SOFTMAX && thin_dense_applies(a);
        a.act = (m.activation[l] == GNN_ACT_SOFTMAX && !thin_softmax) ? GNN_ACT_LINEAR : m.activation[l];
        a.Y = hs[l]; a.ldy = m.units[l];
        if (pred && l == m.n_layers - 1 && a.H <= 64 && a.H > 4 && m.activation[l] != GNN_ACT_SOFTMAX && !drop_at(drop, m.n_layers)) {   // predicate in the epilogue
            a.pred_old = pred->old; a.ld_pred = pred->ld; a.pred_thr = pred->thr; a.pred_flag = pred->flag;
            a.pred_k = pred->k_out; a.pred_kval = pred->k_val;
            pred->fused = true;
        }
        TRY(launch_segdense(a, st));
        if (m.activation[l] == GNN_ACT_SOFTMAX && !thin_softmax) TRY(launch_softmax(gate, a.Y, a.M, a.H, a.ldy, nullptr, st));
    }
    if (!skip_last_drop && drop_at(drop, m.n_layers)) {
        const int H = m.units[m.n_layers - 1];
        TRY(run_dropout(*drop, m.n_layers, hs[m.n_layers - 1], H, drop->buf[m.n_layers], H, M, H, false, st));
    }
    return 0;
}

int dense_plain(const float *X, int ldx, int K, const float *W, int ldw, int H, int M, float *Y, int ldy, hipStream_t st) {
    gnn::SegDenseArgs a;
    memset(&a, 0, sizeof(a));
    a.M = M; a.H = H; a.nseg = 1;
    a.seg[0] = gnn::Seg{X, nullptr, ldx, K, 0};
    a.W = W; a.ldw = ldw; a.act = GNN_ACT_LINEAR; a.Y = Y; a.ldy = ldy;
    return launch_segdense(a, st);
}

int act_grad_inplace(float *G, int ldg, const float *Y, int ldy, int M, int H, int act, hipStream_t st) {
    if (M == 0) return 0;
    const long total = act == GNN_ACT_SOFTMAX ? M : (long)M * H;
    gnn::k_act_grad<<<std::min(cdiv(total, 256), 256 * 16), 256, 0, st>>>(G, ldg, Y, ldy, G, ldg, M, H, act);
    LAUNCH_OK();
    return 0;
}

// Back-propagation through one network.  G = d loss / d hs[last] ([M x units_last], leading dimension ldg; overwritten).
// Parameter gradients go to x.g (accumulated when `accumulate`); dx_all (optional) receives d loss / d input columns
// [0, kdx) BEFORE the BatchNormalization input gradient (the caller applies it to the segments it needs).
// `second_row` >= 0: dx_all is [M x kdx] = columns [0, kdx/2) of the input followed by columns [second_row, second_row + kdx/2);
// `first_col`: the block starts at that input column instead of 0 (composite models: the state segment sits behind the type's labels).
// `drop`: the Dropout layers of this call (their forward copies in drop->buf[]; the masks are regenerated from the call's keys).
int net_backward(NetCtx &x, const gnn::Seg *segs, int nseg, float *const *hs, float *G, int ldg, int M, const float *stats, bool accumulate,
                 float *dx_all, int kdx, float *part, hipStream_t st, int second_row = -1, int first_col = 0, const DropRun *drop = nullptr) {
    const gnn_mlp_t &m = *x.m;
    int n_chunks;
    const int rpc = rows_per_chunk_for(std::max(M, 1), &n_chunks);
    if (drop_at(drop, m.n_layers) && M > 0) TRY(run_dropout(*drop, m.n_layers, G, ldg, G, ldg, M, m.units[m.n_layers - 1], true, st));
    for (int l = m.n_layers - 1; l >= 1; --l) {
        const int H = m.units[l], Kp = m.units[l - 1];
        TRY(act_grad_inplace(G, ldg, hs[l], H, M, H, m.activation[l], st));
        dim3 grid(n_chunks, cdiv(Kp, 64), cdiv(H, 64));
        gnn::k_dense_grad_partial<<<grid, 256, 0, st>>>(drop_at(drop, l) ? drop->buf[l] : hs[l - 1], Kp, nullptr, Kp, G, ldg, H, M, rpc, part, 1);
        LAUNCH_OK();
        const int n = Kp * H + H;
        gnn::k_reduce_partials<<<cdiv(n, 64), 256, 0, st>>>(part, n_chunks, n, x.g->dkernel[l], accumulate ? 1 : 0, 1.0f, Kp * H, x.g->dbias[l]);
        LAUNCH_OK();
        float *Gn = (G == x.G[0]) ? x.G[1] : x.G[0];
        TRY(dense_plain(G, ldg, H, x.Wt[l], Kp, Kp, M, Gn, Kp, st));      // dL/d (input of Dense l) = dZ . W[l]^T
        if (drop_at(drop, l) && M > 0) TRY(run_dropout(*drop, l, Gn, Kp, Gn, Kp, M, Kp, true, st));      // ... through the Dropout layers in front of it
        G = Gn; ldg = Kp;
    }
    const int H = m.units[0], K = m.in_dim;
    int Kv = 0;
    for (int s = 0; s < nseg; ++s) Kv += segs[s].width;
    const bool allk = Kv == K && K <= 192 && H <= 64 && n_chunks >= 256;
    TRY(act_grad_inplace(G, ldg, hs[0], H, M, H, m.activation[0], st));
    {
        gnn::GradSegs gs;
        memset(&gs, 0, sizeof(gs));
        gs.n = nseg; gs.blk_begin[0] = 0;
        gs.mean = (m.has_bn && stats) ? stats : nullptr;      // (rows centred as they are staged: see GradSegs::mean)
        for (int s = 0; s < nseg; ++s) { gs.seg[s] = segs[s]; gs.blk_begin[s + 1] = gs.blk_begin[s] + cdiv(segs[s].width, 64); }
        if (allk) {     // every column in one workgroup: dZ is read once (kernels_train.hpp)
            gnn::k_dense_grad_allk<<<n_chunks, 256, 0, st>>>(gs, K, G, ldg, H, M, rpc, part);
        } else {
            dim3 grid(n_chunks, gs.blk_begin[nseg], cdiv(H, 64));
            gnn::k_dense_grad_partial_segs<<<grid, 256, 0, st>>>(gs, K, G, ldg, H, M, rpc, part);
        }
        LAUNCH_OK();
    }
    const bool bn = m.has_bn != 0;
    const float *Pp = x.P, *qp = x.q;
    int fuse_chunks = 1;
    if (n_chunks <= 32) {                      // small batches: the chunk partials are summed inside the parameter-gradient kernel
        Pp = part; qp = part + (size_t)K * H; fuse_chunks = n_chunks;
    } else {
        const int n = K * H + H;
        gnn::k_reduce_partials<<<cdiv(n, 64), 256, 0, st>>>(part, n_chunks, n, x.P, 0, 1.0f, K * H, x.q);
        LAUNCH_OK();
    }
    gnn::k_first_layer_param_grads<<<K, 64, 0, st>>>(
        Pp, qp, m.kernel[0], K, H, bn ? m.bn_gamma : nullptr, m.bn_beta, stats, stats ? stats + K : nullptr, m.bn_eps, 1.0f / (float)M,
        x.g->dkernel[0], x.g->dbias[0], x.g->dgamma, x.g->dbeta, bn ? x.m1 : nullptr, bn ? x.m2 : nullptr, accumulate ? 1 : 0, fuse_chunks,
        (bn && stats) ? 1 : 0);
    LAUNCH_OK();
    if (dx_all && kdx > 0) {
        if (second_row < 0 || second_row == first_col + kdx / 2) {
            TRY(dense_plain(G, ldg, H, x.Wt[0] + first_col, K, kdx, M, dx_all, kdx, st));   // W^T columns [first_col, first_col + kdx) (ldw = K)
        } else {
            TRY(dense_plain(G, ldg, H, x.Wt[0] + first_col, K, kdx / 2, M, dx_all, kdx, st));
            TRY(dense_plain(G, ldg, H, x.Wt[0] + second_row, K, kdx / 2, M, dx_all + kdx / 2, kdx, st));
        }
    }
    return 0;
}

int bn_input_grads(const gnn_mlp_t &m, const NetCtx &x, const float *stats, const gnn::BnGradReq *reqs, int n, int M, hipStream_t st) {
    if (!m.has_bn || n == 0 || M == 0) return 0;
    gnn::BnGradArgs a;
    memset(&a, 0, sizeof(a));
    a.n = n; a.M = M; a.gamma = m.bn_gamma; a.mean = stats; a.var = stats + m.in_dim; a.m1 = x.m1; a.m2 = x.m2; a.eps = m.bn_eps;
    long total = 0;
    for (int i = 0; i < n; ++i) { a.r[i] = reqs[i]; total = std::max(total, (long)M * reqs[i].width); }
    gnn::k_bn_input_grad_segs<<<std::min(cdiv(total, 256), 256 * 16), 256, 0, st>>>(a);
    LAUNCH_OK();
    return 0;
}

int transposes(NetCtx &x, hipStream_t st) {
    const gnn_mlp_t &m = *x.m;
    int fan_in = m.in_dim;
    for (int l = 0; l < m.n_layers; ++l) {
        gnn::k_transpose<<<std::min(cdiv((long)fan_in * m.units[l], 256), 1024), 256, 0, st>>>(m.kernel[l], fan_in, m.units[l], x.Wt[l]);
        LAUNCH_OK();
        fan_in = m.units[l];
    }
    return 0;
}

int zero_grads(const gnn_mlp_t &m, const gnn_mlp_grads_t &g, hipStream_t st) {
    if (m.has_bn) { HIP_OK(hipMemsetAsync(g.dgamma, 0, sizeof(float) * m.in_dim, st)); HIP_OK(hipMemsetAsync(g.dbeta, 0, sizeof(float) * m.in_dim, st)); }
    int fan_in = m.in_dim;
    for (int l = 0; l < m.n_layers; ++l) {
        HIP_OK(hipMemsetAsync(g.dkernel[l], 0, sizeof(float) * (size_t)fan_in * m.units[l], st));
        HIP_OK(hipMemsetAsync(g.dbias[l], 0, sizeof(float) * m.units[l], st));
        fan_in = m.units[l];
    }
    return 0;
}

int scale_grads(const gnn_mlp_t &m, const gnn_mlp_grads_t &g, float s, hipStream_t st) {
    auto sc = [&](float *p, size_t n) -> int {
        gnn::k_axpby<<<std::min(cdiv((long)n, 256), 1024), 256, 0, st>>>(s, p, 0.0f, nullptr, p, n);
        return hipGetLastError() == hipSuccess ? 0 : 1;
    };
    if (m.has_bn) { if (sc(g.dgamma, m.in_dim) || sc(g.dbeta, m.in_dim)) return fail("scale launch failed"); }
    int fan_in = m.in_dim;
    for (int l = 0; l < m.n_layers; ++l) {
        if (sc(g.dkernel[l], (size_t)fan_in * m.units[l]) || sc(g.dbias[l], m.units[l])) return fail("scale launch failed");
        fan_in = m.units[l];
    }
    return 0;
}

int check_grads(const gnn_mlp_t &m, const gnn_mlp_grads_t &g, const char *name) {
    if (m.has_bn && (!g.dgamma || !g.dbeta)) return fail("%s: dgamma / dbeta is NULL", name);
    for (int l = 0; l < m.n_layers; ++l) if (!g.dkernel[l] || !g.dbias[l]) return fail("%s: gradient buffer of layer %d is NULL", name, l);
    return 0;
}

// segments of the state network's input at iteration t (reference GNN.py:222-231): [state | nodes | agg | agg_nodes | agg_arcs]
int state_segs(const gnn_loop_args_t &a, const TrainPlan &p, int t, gnn::Seg *segs) {
    int n = 0, col = 0;
    const float *st_t = p.states + (size_t)t * p.N * p.ldS;
    segs[n++] = gnn::Seg{st_t, nullptr, p.ldS, p.S, col}; col += p.S;
    if (p.with_labels) { segs[n++] = gnn::Seg{a.nodes, nullptr, a.ld_nodes, p.L, col}; col += p.L; }
    segs[n++] = gnn::Seg{p.agg + (p.agg_taped ? (size_t)t * p.N * p.ldS : 0), nullptr, p.ldS, p.S, col}; col += p.S;
    if (p.with_labels) { segs[n++] = gnn::Seg{p.agg_nodes, nullptr, p.L, p.L, col}; col += p.L; }
    if (p.A > 0) { segs[n++] = gnn::Seg{p.agg_arcs, nullptr, p.A, p.A, col}; col += p.A; }
    return n;
}

// ---- launchers of the large-graph kernels -------------------------------------------------------------------------------------------
// the gathers with their eight source ids, then their eight rows requested together through buffer descriptors (kernels_general.hpp
// gather_sum8<.., BUF>): the arrays must fit 4 GiB windows; GNN_GATHER_BUF=0 keeps the pointer form (a dependent pair of round trips per arc)
inline bool gather_buf_enabled() {
    int v = -1;      // (read at every call: tests switch it inside one process)
    { const char *e = getenv("GNN_GATHER_BUF"); v = (e && e[0] == '0') ? 0 : 1; }
    return v != 0;
}
inline bool gather_buf_ok(const gnn_csr_t &c, int ldx) {
    return gather_buf_enabled() && (size_t)c.n_src * (size_t)ldx * 4 < 0xFFFFFFF0ull && (size_t)c.nnz * 4 < 0xFFFFFFF0ull;
}
// `grid_out` != NULL: the partials stay in `part` ([*grid_out][2 S]) for the caller's own k_stats_finish (heterogeneous models: one for all types)
int launch_aggregate_stats(const int *gate, const gnn_csr_t &c, const float *X, int S, float *out, float *part, float *mean, float *var, const float *shift,
                           hipStream_t st, int *grid_out = nullptr) {
    const int lpr = S / 4, groups = 256 / lpr;
    const int grid = std::min(cdiv(c.n_dst, groups), BIG_AGG_BLOCKS);
#define AGGS_(L, W_, B_) gnn::k_aggregate_stats<L, W_, B_><<<grid, 256, 0, st>>>(gate, c.n_dst, c.rowptr, c.src, c.w, c.row_scale, X, S, out, S, part, shift)
#define AGGS(L) (buf ? (c.w ? AGGS_(L, true, true) : AGGS_(L, false, true)) : (c.w ? AGGS_(L, true, false) : AGGS_(L, false, false)))
    const bool buf = gather_buf_ok(c, S);
    switch (lpr) { case 4: AGGS(4); break; case 8: AGGS(8); break; default: AGGS(16); break; }
#undef AGGS
#undef AGGS_
    LAUNCH_OK();
    if (grid_out) { *grid_out = grid; return 0; }
    gnn::k_stats_finish<<<S, 256, 0, st>>>(gate, part, grid, S, 1.0f / (float)c.n_dst, mean, var, shift);
    LAUNCH_OK();
    return 0;
}

// column statistics of a row-major [M][F] matrix (F = 16 / 32 / 64) in one pass; columns [c0, c0 + width) go to mean / var [dst0 ..)
int rows_stats(const int *gate, const float *X, int ld, int F, int M, float *part, hipStream_t st, int *grid_out) {
    const int lpr = F / 4, groups = 256 / lpr;
    const int grid = std::max(1, std::min(cdiv(M, groups * 4), BIG_FWD_BLOCKS));
    switch (lpr) {
        case 4: gnn::k_rows_stats<4><<<grid, 256, 0, st>>>(gate, M, X, ld, part); break;
        case 8: gnn::k_rows_stats<8><<<grid, 256, 0, st>>>(gate, M, X, ld, part); break;
        default: gnn::k_rows_stats<16><<<grid, 256, 0, st>>>(gate, M, X, ld, part); break;
    }
    LAUNCH_OK();
    *grid_out = grid;
    return 0;
}

template <int SQ>
int launch_train_fwd_sq(const gnn::TrainFwdArgs &fa, int grid, hipStream_t st) {
    gnn::k_train_fwd<SQ, SQ><<<grid, 64 * gnn::TB_WAVES, gnn::train_fwd_lds<SQ, SQ>(), st>>>(fa);
    return hipGetLastError() == hipSuccess ? 0 : fail("k_train_fwd launch failed");
}

// GNN_TRAIN_BF16X6=0: the exact-f32 MFMA kernels (k_train_fwd ..) instead of the three-term bf16 split on the bf16 matrix cores
inline bool train_bf16x6_enabled() {
    int v = -1;      // (read at every call: tests switch it inside one process)
    { const char *e = getenv("GNN_TRAIN_BF16X6"); v = (e && e[0] == '0') ? 0 : 1; }
    return v != 0;
}

template <int SQ, int ACT>
int launch_train_fwd_b6_sa(const gnn::TrainFwdArgs &fa, int grid, hipStream_t st) {
    static bool attr = false;
    const size_t lds = gnn::train_fwd_b6_lds<SQ>();
    if (!attr) {
        if (hipFuncSetAttribute((const void *)gnn::k_train_fwd_b6<SQ, ACT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return fail("k_train_fwd_b6: cannot raise the dynamic LDS limit");
        attr = true;
    }
    gnn::k_train_fwd_b6<SQ, ACT><<<grid, 64 * gnn::TB_WAVES, lds, st>>>(fa);
    return hipGetLastError() == hipSuccess ? 0 : fail("k_train_fwd_b6 launch failed");
}

template <int SQ>
int launch_train_fwd_b6_s(const gnn::TrainFwdArgs &fa, int grid, hipStream_t st) {
    switch (fa.act) {
        case GNN_ACT_LINEAR: return launch_train_fwd_b6_sa<SQ, GNN_ACT_LINEAR>(fa, grid, st);
        case GNN_ACT_RELU: return launch_train_fwd_b6_sa<SQ, GNN_ACT_RELU>(fa, grid, st);
        case GNN_ACT_SELU: return launch_train_fwd_b6_sa<SQ, GNN_ACT_SELU>(fa, grid, st);
        case GNN_ACT_TANH: return launch_train_fwd_b6_sa<SQ, GNN_ACT_TANH>(fa, grid, st);
        case GNN_ACT_SIGMOID: return launch_train_fwd_b6_sa<SQ, GNN_ACT_SIGMOID>(fa, grid, st);
        case GNN_ACT_ELU: return launch_train_fwd_b6_sa<SQ, GNN_ACT_ELU>(fa, grid, st);
        case GNN_ACT_SOFTPLUS: return launch_train_fwd_b6_sa<SQ, GNN_ACT_SOFTPLUS>(fa, grid, st);
        default: return -1;                            // (softmax state networks are not on the large-graph path)
    }
}

int launch_train_fwd(const gnn::TrainFwdArgs &fa, int S, hipStream_t st, int *grid_out) {
    if (train_bf16x6_enabled() && fa.H == S) {
        const int n_tiles16 = (fa.M + 15) / 16;
        const int grid = std::max(1, std::min(std::min(device_cus(), BIG_FWD_BLOCKS), cdiv(n_tiles16, gnn::TB_WAVES)));     // one 8-wave workgroup per CU (192 registers)
        int rc = -1;
        switch (S) {
            case 16: rc = launch_train_fwd_b6_s<1>(fa, grid, st); break;
            case 32: rc = launch_train_fwd_b6_s<2>(fa, grid, st); break;
            case 64: rc = launch_train_fwd_b6_s<4>(fa, grid, st); break;
            default: break;
        }
        if (rc >= 0) { *grid_out = grid; return rc; }
    }
    const int n_tiles = (fa.M + 15) / 16;
    const int grid = std::max(1, std::min(std::min(2 * device_cus(), BIG_FWD_BLOCKS), cdiv(n_tiles, gnn::TB_WAVES)));
    *grid_out = grid;
    switch (S) {
        case 16: return launch_train_fwd_sq<1>(fa, grid, st);
        case 32: return launch_train_fwd_sq<2>(fa, grid, st);
        default: return launch_train_fwd_sq<4>(fa, grid, st);
    }
}

template <int HQ, int ACT>
int launch_train_bwd_b6_ha(const gnn::TrainBwdArgs &ba, int grid, hipStream_t st) {
    static bool attr = false;
    const size_t lds = gnn::train_bwd_b6_lds<HQ>();
    if (!attr) {
        if (hipFuncSetAttribute((const void *)gnn::k_train_bwd_dx_b6<HQ, ACT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return fail("k_train_bwd_dx_b6: cannot raise the dynamic LDS limit");
        attr = true;
    }
    gnn::k_train_bwd_dx_b6<HQ, ACT><<<grid, 256, lds, st>>>(ba);
    return hipGetLastError() == hipSuccess ? 0 : fail("k_train_bwd_dx_b6 launch failed");
}

template <int HQ>
int launch_train_bwd_b6_h(const gnn::TrainBwdArgs &ba, int grid, hipStream_t st) {
    switch (ba.Y ? ba.act : GNN_ACT_LINEAR) {          // (dZ already formed: the kernel's Y loads fall out of range, act'(0) = 1)
        case GNN_ACT_LINEAR: return launch_train_bwd_b6_ha<HQ, GNN_ACT_LINEAR>(ba, grid, st);
        case GNN_ACT_RELU: return launch_train_bwd_b6_ha<HQ, GNN_ACT_RELU>(ba, grid, st);
        case GNN_ACT_SELU: return launch_train_bwd_b6_ha<HQ, GNN_ACT_SELU>(ba, grid, st);
        case GNN_ACT_TANH: return launch_train_bwd_b6_ha<HQ, GNN_ACT_TANH>(ba, grid, st);
        case GNN_ACT_SIGMOID: return launch_train_bwd_b6_ha<HQ, GNN_ACT_SIGMOID>(ba, grid, st);
        case GNN_ACT_ELU: return launch_train_bwd_b6_ha<HQ, GNN_ACT_ELU>(ba, grid, st);
        case GNN_ACT_SOFTPLUS: return launch_train_bwd_b6_ha<HQ, GNN_ACT_SOFTPLUS>(ba, grid, st);
        default: return -1;
    }
}

// the weight gradient on v_mfma_f32_32x32x2_f32 (exact f32 like k_train_wgrad; S = 32 / 64): true when launched
template <int NB>
bool launch_train_wgrad32_nb(const gnn::TrainWgradArgs &wa, int grid, hipStream_t st) {
    switch (wa.act) {
        case GNN_ACT_LINEAR: gnn::k_train_wgrad32<NB, GNN_ACT_LINEAR><<<grid, 256, 0, st>>>(wa); return true;
        case GNN_ACT_RELU: gnn::k_train_wgrad32<NB, GNN_ACT_RELU><<<grid, 256, 0, st>>>(wa); return true;
        case GNN_ACT_SELU: gnn::k_train_wgrad32<NB, GNN_ACT_SELU><<<grid, 256, 0, st>>>(wa); return true;
        case GNN_ACT_TANH: gnn::k_train_wgrad32<NB, GNN_ACT_TANH><<<grid, 256, 0, st>>>(wa); return true;
        case GNN_ACT_SIGMOID: gnn::k_train_wgrad32<NB, GNN_ACT_SIGMOID><<<grid, 256, 0, st>>>(wa); return true;
        case GNN_ACT_ELU: gnn::k_train_wgrad32<NB, GNN_ACT_ELU><<<grid, 256, 0, st>>>(wa); return true;
        case GNN_ACT_SOFTPLUS: gnn::k_train_wgrad32<NB, GNN_ACT_SOFTPLUS><<<grid, 256, 0, st>>>(wa); return true;
        default: return false;
    }
}
inline bool train_wgrad32_enabled() {       // GNN_TRAIN_WGRAD32=0: k_train_wgrad (16x16x4) at every width
    int v = -1;      // (read at every call: tests switch it inside one process)
    { const char *e = getenv("GNN_TRAIN_WGRAD32"); v = (e && e[0] == '0') ? 0 : 1; }
    return v != 0;
}
bool launch_train_wgrad32(const gnn::TrainWgradArgs &wa, int S, int grid, hipStream_t st) {
    if (!train_wgrad32_enabled()) return false;
    if (S == 64) return launch_train_wgrad32_nb<2>(wa, grid, st);
    if (S == 32) return launch_train_wgrad32_nb<1>(wa, grid, st);
    return false;
}

// ... and on the bf16 matrix cores (k_train_wgrad_b6: three-term splits, rows through an LDS ring; GNN_TRAIN_WGRAD_B6=0 keeps the f32-input
// kernel): rows_per_wg must be a multiple of 64, one workgroup per CU
template <int NB, int ACT>
bool launch_train_wgrad_b6_na(const gnn::TrainWgradArgs &wa, int grid, hipStream_t st) {
    static bool attr = false;
    const size_t lds = gnn::train_wgrad_b6_lds<NB, ACT>();
    if (!attr) {
        if (hipFuncSetAttribute((const void *)gnn::k_train_wgrad_b6<NB, ACT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return false;
        attr = true;
    }
    gnn::k_train_wgrad_b6<NB, ACT><<<grid, 256, lds, st>>>(wa);
    return true;
}
template <int NB>
bool launch_train_wgrad_b6_nb(const gnn::TrainWgradArgs &wa, int grid, hipStream_t st) {
    switch (wa.act) {
        case GNN_ACT_LINEAR: return launch_train_wgrad_b6_na<NB, GNN_ACT_LINEAR>(wa, grid, st);
        case GNN_ACT_RELU: return launch_train_wgrad_b6_na<NB, GNN_ACT_RELU>(wa, grid, st);
        case GNN_ACT_SELU: return launch_train_wgrad_b6_na<NB, GNN_ACT_SELU>(wa, grid, st);
        case GNN_ACT_TANH: return launch_train_wgrad_b6_na<NB, GNN_ACT_TANH>(wa, grid, st);
        case GNN_ACT_SIGMOID: return launch_train_wgrad_b6_na<NB, GNN_ACT_SIGMOID>(wa, grid, st);
        case GNN_ACT_ELU: return launch_train_wgrad_b6_na<NB, GNN_ACT_ELU>(wa, grid, st);
        case GNN_ACT_SOFTPLUS: return launch_train_wgrad_b6_na<NB, GNN_ACT_SOFTPLUS>(wa, grid, st);
        default: return false;
    }
}
inline bool train_wgrad_b6_enabled() {
    int v = -1;      // (read at every call: tests switch it inside one process)
    { const char *e = getenv("GNN_TRAIN_WGRAD_B6"); v = (e && e[0] == '0') ? 0 : 1; }
    return v != 0 && train_bf16x6_enabled() && train_wgrad32_enabled();
}
bool launch_train_wgrad_b6(const gnn::TrainWgradArgs &wa, int S, int grid, hipStream_t st) {
    if (S == 64) return launch_train_wgrad_b6_nb<2>(wa, grid, st);
    if (S == 32) return launch_train_wgrad_b6_nb<1>(wa, grid, st);
    return false;
}

// weight gradient and input gradient of an iteration in one pass over its rows (k_train_wgrad_dx_b6; the dZ form; GNN_TRAIN_FUSED_BWD=0: the two kernels)
inline bool train_fused_bwd_enabled() {
    int v = -1;      // (read at every call: tests switch it inside one process)
    { const char *e = getenv("GNN_TRAIN_FUSED_BWD"); v = (e && e[0] == '0') ? 0 : 1; }
    return v != 0;
}
template <int NB>
bool launch_train_wgrad_dx_b6_nb(const gnn::TrainWgradArgs &wa, const gnn::TrainBwdArgs &ba, int grid, hipStream_t st) {
    static bool attr = false;
    const size_t lds = gnn::train_wgrad_dx_b6_lds<NB>();
    if (!attr) {
        if (hipFuncSetAttribute((const void *)gnn::k_train_wgrad_dx_b6<NB>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return false;
        attr = true;
    }
    gnn::k_train_wgrad_dx_b6<NB><<<grid, 256, lds, st>>>(wa, ba);
    return true;
}
bool launch_train_wgrad_dx_b6(const gnn::TrainWgradArgs &wa, const gnn::TrainBwdArgs &ba, int S, int grid, hipStream_t st) {
    // With BatchNormalization the input gradient needs m1 / m2 - column moments of dZ W^T that k_first_layer_param_grads derives from THIS
    // iteration's finished P and q - so the two products cannot share a pass there (the kernel itself takes them as given: scripts/micro/
    // rowgemm_check.hip checks it bit for bit against the two kernels on arbitrary m1 / m2).  Without it nothing global stands between them.
    if (ba.gamma) return false;
    if (wa.Y || wa.act != GNN_ACT_LINEAR || ba.Y) return false;           // the dZ form only
    if (ba.H != S || ba.S != S || ba.ldz != S || ba.ld_state != S || ba.ld_agg != S || ba.M != wa.M || ba.dZ != wa.G || ba.agg != wa.agg) return false;
    if ((size_t)wa.rows_per_wg * (size_t)ba.ld_dx * 4 >= 0xFFFFFFF0ull) return false;
    if (S == 64) return launch_train_wgrad_dx_b6_nb<2>(wa, ba, grid, st);
    if (S == 32) return launch_train_wgrad_dx_b6_nb<1>(wa, ba, grid, st);
    return false;
}

int launch_train_bwd_dx(const gnn::TrainBwdArgs &ba, int S, hipStream_t st) {
    if (train_bf16x6_enabled() && ba.H == S && ba.S == S && ba.ldz == S) {
        const int grid = std::max(1, std::min(2 * device_cus(), cdiv((ba.M + 15) / 16, 4)));       // 256-thread workgroups, two per CU
        int rc = -1;
        switch (S) {
            case 16: rc = launch_train_bwd_b6_h<1>(ba, grid, st); break;
            case 32: rc = launch_train_bwd_b6_h<2>(ba, grid, st); break;
            case 64: rc = launch_train_bwd_b6_h<4>(ba, grid, st); break;
            default: break;
        }
        if (rc >= 0) return rc;
    }
    const int n_tiles = (ba.M + 15) / 16;
    const int grid = std::max(1, std::min(2 * device_cus(), cdiv(n_tiles, gnn::TB_WAVES)));
    switch (S) {
        case 16: gnn::k_train_bwd_dx<1, 2><<<grid, 64 * gnn::TB_WAVES, gnn::train_bwd_lds<1, 2>(), st>>>(ba); break;
        case 32: gnn::k_train_bwd_dx<2, 4><<<grid, 64 * gnn::TB_WAVES, gnn::train_bwd_lds<2, 4>(), st>>>(ba); break;
        default: gnn::k_train_bwd_dx<4, 8><<<grid, 64 * gnn::TB_WAVES, gnn::train_bwd_lds<4, 8>(), st>>>(ba); break;
    }
    LAUNCH_OK();
    return 0;
}

// G_{t-1} -> dZ_{t-1} in the transposed aggregate's epilogue (kernels_train_big.hpp: k_aggregate_dz); false: no instance for this shape
inline bool train_dz_enabled() {          // GNN_TRAIN_DZ=0: the round-4 flow (every dense kernel forms dZ itself, the last iteration's dx is still computed)
    int v = -1;      // (read at every call: tests switch it inside one process)
    { const char *e = getenv("GNN_TRAIN_DZ"); v = (e && e[0] == '0') ? 0 : 1; }
    return v != 0;
}
template <int LPR, bool HAS_W>
bool launch_aggregate_dz_lw(const gnn_csr_t &c, const float *Xa, int ldx, float *out, int ldo, const float *addend, int ld_add, const gnn::AggDzArgs &z, int act,
                            int grid, hipStream_t st) {
    const bool buf = gather_buf_ok(c, ldx);
#define AGGDZ_(A_, B_) gnn::k_aggregate_dz<LPR, HAS_W, A_, B_><<<grid, 256, 0, st>>>(c.n_dst, c.rowptr, c.src, c.w, c.row_scale, Xa, ldx, out, ldo, addend, ld_add, z)
#define AGGDZ(A_) do { if (buf) AGGDZ_(A_, true); else AGGDZ_(A_, false); } while (0)
    switch (act) {
        case GNN_ACT_LINEAR: AGGDZ(GNN_ACT_LINEAR); return true;
        case GNN_ACT_RELU: AGGDZ(GNN_ACT_RELU); return true;
        case GNN_ACT_SELU: AGGDZ(GNN_ACT_SELU); return true;
        case GNN_ACT_TANH: AGGDZ(GNN_ACT_TANH); return true;
        case GNN_ACT_SIGMOID: AGGDZ(GNN_ACT_SIGMOID); return true;
        case GNN_ACT_ELU: AGGDZ(GNN_ACT_ELU); return true;
        case GNN_ACT_SOFTPLUS: AGGDZ(GNN_ACT_SOFTPLUS); return true;
        default: return false;
    }
#undef AGGDZ
#undef AGGDZ_
}
bool launch_aggregate_dz(const gnn_csr_t &c, const float *Xa, int ldx, float *out, int ldo, const float *addend, int ld_add, const gnn::AggDzArgs &z, int act, int S,
                         hipStream_t st) {
    const int lpr = S / 4, groups = 256 / lpr, grid = std::min(cdiv(c.n_dst, groups), 256 * 16);
    switch (lpr) {
        case 4: return c.w ? launch_aggregate_dz_lw<4, true>(c, Xa, ldx, out, ldo, addend, ld_add, z, act, grid, st) : launch_aggregate_dz_lw<4, false>(c, Xa, ldx, out, ldo, addend, ld_add, z, act, grid, st);
        case 8: return c.w ? launch_aggregate_dz_lw<8, true>(c, Xa, ldx, out, ldo, addend, ld_add, z, act, grid, st) : launch_aggregate_dz_lw<8, false>(c, Xa, ldx, out, ldo, addend, ld_add, z, act, grid, st);
        case 16: return c.w ? launch_aggregate_dz_lw<16, true>(c, Xa, ldx, out, ldo, addend, ld_add, z, act, grid, st) : launch_aggregate_dz_lw<16, false>(c, Xa, ldx, out, ldo, addend, ld_add, z, act, grid, st);
        default: return false;
    }
}

gnn::ConstSegs const_segs_of(const gnn_loop_args_t &a, const TrainPlan &p) {
    gnn::ConstSegs cs;
    memset(&cs, 0, sizeof(cs));
    gnn::Seg segs[GNN_MAX_SEGS];
    const int n0 = state_segs(a, p, 0, segs);
    for (int s = 0; s < n0; ++s)
        if (segs[s].ptr != p.states && segs[s].ptr != p.agg) { cs.ptr[cs.n] = segs[s].ptr; cs.ld[cs.n] = segs[s].ld; cs.width[cs.n] = segs[s].width; cs.wrow[cs.n] = segs[s].wrow; ++cs.n; }
    return cs;
}

template <typename Kern, typename Args, typename... Extra>
int launch_persistent(Kern kern, const Args &args, const gnn::TileTab &tt, int n_wg, size_t lds, hipStream_t st, const Extra &... extra) {
    lds = std::max(lds, gnn::TS_LDS);
    static std::vector<const void *> allowed;          // kernels whose dynamic-LDS limit has been raised
    static std::mutex allowed_mutex;                   // (callers may run steps of different models on different host threads)
    const void *fn = reinterpret_cast<const void *>(kern);
    {
        std::lock_guard<std::mutex> lock(allowed_mutex);
        if (std::find(allowed.begin(), allowed.end(), fn) == allowed.end()) {
            HIP_OK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));      // (the same for every launch of one kernel)
            allowed.push_back(fn);
        }
    }
    // every workgroup waits for the others at the grid barriers: the grid must fit the device (n_wg <= CUs by the plan; this asks the
    // runtime about THIS kernel's registers / LDS)
    if (!gnn::persistent_fits(fn, gnn::TS_NT, lds, n_wg, device_cus())) return fail("persistent training kernel: %d workgroups cannot be resident at once", n_wg);
    kern<<<n_wg, gnn::TS_NT, lds, st>>>(args, tt, extra...);
    LAUNCH_OK();
    return 0;
}

// `yt` (heterogeneous models: tiles cut at TYPE boundaries, arcs cross them: the general form with a tile table); yt == NULL: homogeneous -
// a tile table then means tiles cut at GRAPH boundaries that no arc leaves (the LOCAL form)
template <int SQ>
int launch_train_small_fwd_sq(const gnn::TrainSmallFwd &fa, const gnn::TileTab &tt, int n_wg, bool has_w, hipStream_t st, const gnn::TypeTab *yt = nullptr) {
    const size_t lds = gnn::train_small_fwd_lds<SQ>();
    gnn::TypeTab none;
    memset(&none, 0, sizeof(none));
    const gnn::TypeTab &y = yt ? *yt : none;
    if (tt.n > 0 && !yt) return has_w ? launch_persistent(&gnn::k_train_small_fwd<SQ, true, true>, fa, tt, n_wg, lds, st, y)
                                      : launch_persistent(&gnn::k_train_small_fwd<SQ, false, true>, fa, tt, n_wg, lds, st, y);
    return has_w ? launch_persistent(&gnn::k_train_small_fwd<SQ, true, false>, fa, tt, n_wg, lds, st, y)
                 : launch_persistent(&gnn::k_train_small_fwd<SQ, false, false>, fa, tt, n_wg, lds, st, y);
}

template <int SQ>
int launch_train_small_bwd_sq(const gnn::TrainSmallBwd &ba, const gnn::TileTab &tt, int n_wg, bool has_w, hipStream_t st, const gnn::TypeTab *yt = nullptr,
                              const gnn::TypeConsts *yc = nullptr) {
    const bool local = tt.n > 0 && !yt;
    const size_t lds = gnn::train_small_bwd_lds<SQ>(local);
    gnn::TypeTab none;
    memset(&none, 0, sizeof(none));
    gnn::TypeConsts nonec;
    memset(&nonec, 0, sizeof(nonec));
    const gnn::TypeTab &y = yt ? *yt : none;
    const gnn::TypeConsts &c = yc ? *yc : nonec;
    if (local) return has_w ? launch_persistent(&gnn::k_train_small_bwd<SQ, true, true>, ba, tt, n_wg, lds, st, y, c)
                            : launch_persistent(&gnn::k_train_small_bwd<SQ, false, true>, ba, tt, n_wg, lds, st, y, c);
    return has_w ? launch_persistent(&gnn::k_train_small_bwd<SQ, true, false>, ba, tt, n_wg, lds, st, y, c)
                 : launch_persistent(&gnn::k_train_small_bwd<SQ, false, false>, ba, tt, n_wg, lds, st, y, c);
}

// Back-propagation through a thin output head over every node of a large graph (kernels_train_big.hpp: k_head_wgrad / k_head_dx): the
// head's parameter gradients, and d loss / d state_k written straight into the state gradient.  G = d loss / d head output [M x T].
template <int T>
int head_backward_t(const TrainPlan &p, const gnn_loop_args_t &a, const float *state_k, const float *out_nodes, float *G, const float *stats, hipStream_t st, int dz_act) {
    const gnn_mlp_t &m = a.net_output;
    TRY(act_grad_inplace(G, p.T, out_nodes, p.T, p.M, p.T, m.activation[0], st));
    gnn::HeadArgs h;
    memset(&h, 0, sizeof(h));
    h.M = p.M; h.S = p.S; h.L = p.with_labels ? p.L : 0; h.T = T;
    h.state = state_k; h.ld_state = p.ldS; h.labels = p.with_labels ? a.nodes : nullptr; h.ld_labels = a.ld_nodes;
    h.dZ = G; h.ldz = p.T;
    const int grid = std::min(BIG_HEAD_BLOCKS, cdiv(p.M, 256));
    h.rows_per_wg = cdiv(p.M, grid);
    h.part = p.part_h;
    gnn::k_head_wgrad<T><<<grid, 256, 0, st>>>(h);
    LAUNCH_OK();
    const int K = m.in_dim, n = K * T + T;
    gnn::k_reduce_partials<<<cdiv(n, 64), 256, 0, st>>>(p.part_h, grid, n, p.co.P, 0, 1.0f, K * T, p.co.q);
    LAUNCH_OK();
    const bool bn = m.has_bn != 0;
    gnn::k_first_layer_param_grads<<<K, 64, 0, st>>>(p.co.P, p.co.q, m.kernel[0], K, T, bn ? m.bn_gamma : nullptr, m.bn_beta, stats, stats ? stats + K : nullptr,
                                                     m.bn_eps, 1.0f / (float)p.M, p.co.g->dkernel[0], p.co.g->dbias[0], p.co.g->dgamma, p.co.g->dbeta,
                                                     bn ? p.co.m1 : nullptr, bn ? p.co.m2 : nullptr, 0, 1);
    LAUNCH_OK();
    h.W = m.kernel[0];
    if (bn) { h.gamma = m.bn_gamma; h.mean = stats; h.var = stats + K; h.m1 = p.co.m1; h.m2 = p.co.m2; h.eps = m.bn_eps; }
    h.dx = p.G_state; h.ld_dx = p.S; h.dz_act = dz_act;
    const int lpr = p.S / 4 <= 4 ? 4 : p.S / 4 <= 8 ? 8 : 16;
    gnn::k_head_dx<T><<<std::min(cdiv(p.M, 256 / lpr), 256 * 16), 256, 0, st>>>(h);
    LAUNCH_OK();
    return 0;
}

// `dz_act` >= 0: the state gradient leaves as the loop's last dZ (HeadArgs::dz_act)
int head_backward(const TrainPlan &p, const gnn_loop_args_t &a, const float *state_k, const float *out_nodes, float *G, const float *stats, hipStream_t st, int dz_act = -1) {
    switch (p.T) {
        case 1: return head_backward_t<1>(p, a, state_k, out_nodes, G, stats, st, dz_act);
        case 2: return head_backward_t<2>(p, a, state_k, out_nodes, G, stats, st, dz_act);
        case 3: return head_backward_t<3>(p, a, state_k, out_nodes, G, stats, st, dz_act);
        default: return head_backward_t<4>(p, a, state_k, out_nodes, G, stats, st, dz_act);
    }
}

// the caller's tiles (HOST array), checked: ascending, <= 64 nodes each, covering [0, N)
int tile_table(const gnn_train_args_t &ta, const TrainPlan &p, gnn::TileTab &tt) {
    memset(&tt, 0, sizeof(tt));
    if (!p.tiled) return 0;
    tt.n = ta.n_tiles;
    for (int b = 0; b <= tt.n; ++b) tt.begin[b] = ta.tile_node_begin[b];
    if (tt.begin[0] != 0 || tt.begin[tt.n] != p.N) return fail("tile_node_begin must run from 0 to n_nodes");
    for (int b = 0; b < tt.n; ++b)
        if (tt.begin[b + 1] <= tt.begin[b] || tt.begin[b + 1] - tt.begin[b] > 64) return fail("tile %d has %d nodes (1 .. 64 allowed)", b, tt.begin[b + 1] - tt.begin[b]);
    return 0;
}

// heterogeneous models: train_composite.hpp (behind this file in the same translation unit)
int train_step_composite(const gnn_train_args_t &ta);
size_t composite_train_workspace_bytes(const gnn_train_args_t &ta);

}  // namespace

extern "C" {

size_t gnn_train_workspace_bytes(const gnn_train_args_t *args) {
    if (!args) { fail("args is NULL"); return 0; }
    if (args->loop.composite) return composite_train_workspace_bytes(*args);
    TrainPlan p;
    if (make_train_plan(*args, nullptr, p)) return 0;
    return p.bytes;
}

static int train_step_impl(const gnn_train_args_t &ta);

int gnn_train_step(const gnn_train_args_t *args) {
    if (!args) return fail("args is NULL");
    const int rc = train_step_impl(*args);
    // A call that fails behind its first launches has already fetched the previous step's validity word into *prev_grads_ok_host and reset
    // the word on the tape: the copy must have LANDED when the caller looks at it (the caller's own view of the word is stale by then).  A call
    // that fails in front of them leaves *prev_grads_ok_host as the caller initialised it (a sentinel) and the word on the tape untouched.
    if (rc != 0 && args->prev_grads_ok_host && args->tape) {
        const std::string msg = gnn_last_error();                 // (the synchronisation must not replace the message)
        (void)hipStreamSynchronize((hipStream_t)args->loop.stream);
        fail("%s", msg.c_str());
    }
    return rc;
}

static int train_step_impl(const gnn_train_args_t &ta) {
    const gnn_loop_args_t &a = ta.loop;
    TrainPlan p;
    if (!ta.tape || ((uintptr_t)ta.tape & 255) != 0) return fail("tape must be a 256-byte aligned device buffer");
    const bool fwd_only = ta.forward_only != 0;                   // ABI 9: the training-mode forward alone (no loss, no gradients)
    if (a.composite) {
        if (fwd_only) return fail("gnn_train_step(forward_only): homogeneous models only");
        return train_step_composite(ta);                           // one state network per node type (train_composite.hpp)
    }
    if (fwd_only && ta.prev_grads_ok_host) return fail("gnn_train_step(forward_only): prev_grads_ok_host must be NULL");
    TRY(make_train_plan(ta, ta.tape, p));
    if (ta.tape_bytes < p.bytes) return fail("tape too small: %zu < %zu bytes", ta.tape_bytes, p.bytes);
    TRY(check_csr(a.adjacency, "adjacency", p.N, p.N));
    TRY(check_csr(a.arcnode, "arcnode", p.N, p.E));
    if (!fwd_only) TRY(check_csr(ta.adjacency_by_source, "adjacency_by_source", p.N, p.N));
    if (!a.nodes || (p.E > 0 && p.A > 0 && !a.arc_labels)) return fail("nodes / arc_labels is NULL");
    if (a.state_dim > 0 && !a.state0) return fail("state0 is required when state_dim > 0");
    if (p.M > 0 && !a.out_index) return fail("out_index is NULL");
    if (a.focus == GNN_FOCUS_ARC && p.E > 0 && (!a.arc_src || !a.arc_dst)) return fail("arc focus needs arc_src / arc_dst");
    if (p.pooled) {
        if (a.nodegraph.n_src != p.M) return fail("graph focus: every node must pass the mask");
        TRY(check_csr(a.nodegraph, "nodegraph", p.G, p.M));
        if (!fwd_only) TRY(check_csr(ta.nodegraph_by_source, "nodegraph_by_source", p.M, p.G));
    }
    if (p.R < 1 || (!fwd_only && !ta.targets)) return fail("gnn_train_step needs at least one target row");
    if (!fwd_only && (ta.loss_kind < 0 || ta.loss_kind > 3)) return fail("unknown loss kind %d", ta.loss_kind);
    if (!ta.y_pred || (!fwd_only && !ta.loss) || !ta.k_host || !ta.state) return fail("y_pred / loss / k_host / state is NULL");
    if (!fwd_only) {
        TRY(check_grads(a.net_state[0], ta.grad_state, "grad_state"));
        TRY(check_grads(a.net_output, ta.grad_output, "grad_output"));
    }
    const gnn_mlp_t &ns = a.net_state[0], &no = a.net_output;
    const bool bn_s = ns.has_bn != 0, bn_o = no.has_bn != 0;
    const bool fold_s = ns.units[0] <= 4;       // see the forward loop
    hipStream_t st = (hipStream_t)a.stream;
    GNN_SET_KERNEL_NAME(p.big ? "train_step: row-streaming kernels (kernels_train_big.hpp)" : p.small ? "train_step: persistent small-graph kernels" : "train_step: general kernels");
    const size_t NS = (size_t)p.N * p.ldS;             // one state matrix of the tape (rows of ldS floats: padded on the persistent small-graph path)

    // ---- setup: transposes, aggregates of the constants, state_0, iteration-invariant statistics -----------------------------------
    // the validity word the previous call on this tape left (stream-ordered behind that call's launches; on the host at this call's one
    // synchronisation), then this call's: 0 until the last launch of the step has been issued
    if (ta.prev_grads_ok_host) HIP_OK(hipMemcpyAsync(ta.prev_grads_ok_host, p.grads_ok, sizeof(int), hipMemcpyDeviceToHost, st));
    if (!fwd_only) {               // (a forward leaves the word of the last STEP on this tape as it is)
        HIP_OK(hipMemsetAsync(p.grads_ok, 0, sizeof(int) * 4, st));
        if (ta.grads_ok_dev) *ta.grads_ok_dev = p.grads_ok;
        TRY(transposes(p.cs, st));
        TRY(transposes(p.co, st));
    }
    HIP_OK(hipMemsetAsync(p.flags, 0, sizeof(int) * (p.K + 8), st));
    HIP_OK(hipMemsetAsync(p.k_dev, 0, sizeof(float) * 4, st));
    if (p.A > 0) TRY(launch_aggregate(nullptr, a.arcnode, a.arc_labels, a.ld_arcs, p.A, p.agg_arcs, p.A, st));
    if (p.with_labels) TRY(launch_aggregate(nullptr, a.adjacency, a.nodes, a.ld_nodes, p.L, p.agg_nodes, p.L, st));
    if (a.state_dim > 0 && p.ldS == p.S) HIP_OK(hipMemcpyAsync(p.states, a.state0, sizeof(float) * NS, hipMemcpyDeviceToDevice, st));
    else if (a.state_dim > 0) TRY(launch_copy2d(nullptr, a.state0, p.S, p.states, p.ldS, p.N, p.S, p.ldS, st));
    else TRY(launch_copy2d(nullptr, a.nodes, a.ld_nodes, p.states, p.ldS, p.N, p.S, p.ldS, st));
    gnn::Seg segs[GNN_MAX_SEGS];
    if (p.big && p.Kc > 0) {      // the constant inputs of a node as one 128-byte line: [labels | aggregated labels | aggregated arc labels | 1 | 0 ..]
        gnn::PackSegs ps;
        memset(&ps, 0, sizeof(ps));
        const int n0 = state_segs(a, p, 0, segs);
        for (int s = 0; s < n0; ++s)
            if (segs[s].ptr != p.states && segs[s].ptr != p.agg) { ps.ptr[ps.n] = segs[s].ptr; ps.ld[ps.n] = segs[s].ld; ps.width[ps.n] = segs[s].width; ps.wrow[ps.n] = segs[s].wrow; ++ps.n; }
        gnn::k_pack_xc<<<(int)std::min<long>(cdiv((long)p.N * 32, 256), 256 * 16), 256, 0, st>>>(p.N, nullptr, ps, p.xc);
        LAUNCH_OK();
    }
    if (bn_s && p.K > 0) {
        const int n = state_segs(a, p, 0, segs);
        gnn::Seg cst[GNN_MAX_SEGS]; int nc = 0;
        for (int s = 0; s < n; ++s) if (segs[s].ptr != p.states && segs[s].ptr != p.agg) cst[nc++] = segs[s];
        HIP_OK(hipMemsetAsync(p.stats_tpl, 0, sizeof(float) * 2 * p.in_s, st));
        if (p.big && p.Kc > 0) {  // one pass over the packed line; each segment's columns land at its BatchNorm columns
            int grid = 0;
            TRY(rows_stats(nullptr, p.xc, 32, 32, p.N, p.part_y, st, &grid));
            int c0 = 0;
            for (int s = 0; s < p.cc.n; ++s) {
                gnn::k_stats_finish<<<p.cc.width[s], 256, 0, st>>>(nullptr, p.part_y + c0, grid, 32, 1.0f / (float)p.N, p.stats_tpl + p.cc.wrow[s],
                                                                  p.stats_tpl + p.in_s + p.cc.wrow[s], p.xc + c0);      // (k_rows_stats: moments around row 0)
                LAUNCH_OK();
                c0 += p.cc.width[s];
            }
        } else TRY(colstats_segs(nullptr, cst, nc, p.N, p.stats_tpl, p.stats_tpl + p.in_s, p.part, st));
        gnn::k_replicate<<<std::min(cdiv((long)2 * p.in_s * p.K, 256), 1024), 256, 0, st>>>(p.stats_tpl, 2 * p.in_s, p.K, p.stats_s);
        LAUNCH_OK();
    }
    if (p.head_fast) {     // the thin-head kernels treat output row m as node m: out_index must BE the identity (checked on the device, read with k)
        gnn::k_not_identity<<<std::min(cdiv(p.M, 256), 1024), 256, 0, st>>>(a.out_index, p.M, p.k_dev + 2);
        LAUNCH_OK();
    }
    // ---- training-mode forward: gated iterations, tape = states + statistics + folded first layers ------------------------------
    TRY(launch_converge(nullptr, p.states, nullptr, p.N, p.S, p.ldS, 0, a.state_threshold, p.flags, nullptr, 0.f, st));
    for (int t = 0; t < p.K && p.big; ++t) {
        // Large graphs (kernels_train_big.hpp): the neighbour sum leaves its own column statistics, the dense kernel those of the
        // state it writes (the next iteration's input); the statistics are folded into the weights (tiny launch) and the layer
        // streams its rows straight into the matrix cores.
        const int *gate = p.flags + t;
        const float *s_t = p.states + (size_t)t * NS;
        float *s_n = p.states + (size_t)(t + 1) * NS;
        float *agg_t = p.agg + (p.agg_taped ? (size_t)t * NS : 0);
        float *stats = p.stats_s + (size_t)t * 2 * p.in_s;
        if (bn_s) {
            // (moments around the previous iteration's column means: one-pass sums, but nothing of the mean's size left to cancel)
            const float *prev = t > 0 ? p.stats_s + (size_t)(t - 1) * 2 * p.in_s : nullptr;
            TRY(launch_aggregate_stats(gate, a.adjacency, s_t, p.S, agg_t, p.part_a, stats + p.off_agg, stats + p.in_s + p.off_agg, prev ? prev + p.off_agg : nullptr, st));
            if (t == 0) {          // (later iterations: k_train_fwd leaves the statistics of the state it writes)
                int grid = 0;
                TRY(rows_stats(gate, s_t, p.S, p.S, p.N, p.part_y, st, &grid));
                gnn::k_stats_finish<<<p.S, 256, 0, st>>>(gate, p.part_y, grid, p.S, 1.0f / (float)p.N, stats, stats + p.in_s, s_t);
                LAUNCH_OK();
            }
        } else TRY(launch_aggregate(gate, a.adjacency, s_t, p.S, p.S, agg_t, p.S, st));
        const float *W0 = ns.kernel[0], *b0 = ns.bias[0];
        if (bn_s) {
            float *Wf = p.Wf_s + (size_t)t * p.in_s * p.H1s, *bf = p.bf_s + (size_t)t * p.H1s;
            // (the rows are centred by the kernel as they arrive - Sum a W (x - mean) + (b + Sum beta W): a folded bias b + Sum (beta - mean a) W
            //  would cancel Sum a W mean against itself, and inputs that sit far from zero - raw labels - lose that many digits of z)
            TRY(fold_with_stats(ns, stats, Wf, bf, st, true));
            W0 = Wf; b0 = bf;
        }
        gnn::TrainFwdArgs fa;
        memset(&fa, 0, sizeof(fa));
        fa.in_mean = bn_s ? stats : nullptr;
        fa.gate = gate; fa.M = p.N;
        fa.state = s_t; fa.ld_state = p.S; fa.agg = agg_t; fa.ld_agg = p.S; fa.xc = p.Kc > 0 ? p.xc : nullptr;
        fa.Wf = W0; fa.bf = b0; fa.H = p.S; fa.wrow_state = 0; fa.wrow_agg = p.off_agg; fa.cs = p.cc;
        fa.act = ns.activation[0];
        fa.Y = s_n; fa.ldy = p.S;
        fa.thr = a.state_threshold; fa.pred_flag = p.flags + t + 1; fa.pred_k = p.k_dev; fa.pred_kval = (float)(t + 1);
        const bool next_stats = bn_s && (t + 1 < p.K || (p.head_fast && bn_o));      // (the last state's: the output head's BatchNorm input)
        fa.stat_part = next_stats ? p.part_y : nullptr;
        fa.stat_shift = next_stats ? stats : nullptr;         // (the new state's moments around the input state's column means)
        int grid = 0;
        TRY(launch_train_fwd(fa, p.S, st, &grid));
        if (next_stats) {
            float *nxt = p.stats_s + (size_t)(t + 1) * 2 * p.in_s;
            gnn::k_stats_finish<<<p.S, 256, 0, st>>>(gate, p.part_y, grid, p.S, 1.0f / (float)p.N, nxt, nxt + p.in_s, stats);
            LAUNCH_OK();
        }
    }
    gnn::TileTab tiles;
    TRY(tile_table(ta, p, tiles));
    if (p.small) {
        // Small graphs (kernels_train_small.hpp): all K gated iterations in one persistent launch, one workgroup per 64-node tile
        const gnn::ConstSegs cs = const_segs_of(a, p);
        gnn::k_train_small_const<<<cdiv(p.N * p.SPs, 256), 256, 0, st>>>(p.N, p.SPs, p.S, cs, ns.kernel[0], ns.bias[0], bn_s ? ns.bn_gamma : nullptr, ns.bn_beta,
                                                                       p.stats_tpl, p.stats_tpl + p.in_s, ns.bn_eps, p.sm_cc);
        LAUNCH_OK();
        HIP_OK(hipMemsetAsync(p.sm_bar, 0, sizeof(unsigned long long) * 4, st));
        gnn::TrainSmallFwd fa;
        memset(&fa, 0, sizeof(fa));
        fa.N = p.N; fa.S = p.SPs; fa.Sw = p.S; fa.K = p.K;
        fa.rowptr = a.adjacency.rowptr; fa.src = a.adjacency.src; fa.w = a.adjacency.w; fa.row_scale = a.adjacency.row_scale;
        fa.states = p.states; fa.agg = p.agg; fa.stats = p.stats_s; fa.in_s = p.in_s; fa.off_agg = p.off_agg;
        fa.W = ns.kernel[0]; fa.gamma = bn_s ? ns.bn_gamma : nullptr; fa.beta = ns.bn_beta; fa.eps = ns.bn_eps; fa.act = ns.activation[0];
        fa.Cc = p.sm_cc; fa.thr = a.state_threshold; fa.flag0 = p.flags;
        fa.bar = p.sm_bar; fa.part = p.sm_part; fa.k_out = p.k_dev; fa.wait_ticks = gnn::wait_ticks();
        if (const char *e = getenv("GNN_DEBUG_FAIL_FWD")) { if (e[0] == '1') fa.wait_ticks = 0; }      // (test hook: every barrier wait of THIS forward launch expires at once)
        switch (p.SPs) {
            case 16: TRY(launch_train_small_fwd_sq<1>(fa, tiles, p.n_wg, a.adjacency.w != nullptr, st)); break;
            case 32: TRY(launch_train_small_fwd_sq<2>(fa, tiles, p.n_wg, a.adjacency.w != nullptr, st)); break;
            default: TRY(launch_train_small_fwd_sq<4>(fa, tiles, p.n_wg, a.adjacency.w != nullptr, st)); break;
        }
    }
    for (int t = 0; t < p.K && !p.big && !p.small; ++t) {
        const int *gate = p.flags + t;
        const float *s_t = p.states + (size_t)t * NS;
        float *s_n = p.states + (size_t)(t + 1) * NS;
        float *agg_t = p.agg + (p.agg_taped ? (size_t)t * NS : 0);
        TRY(launch_aggregate(gate, a.adjacency, s_t, p.S, p.S, agg_t, p.S, st));
        const int n = state_segs(a, p, t, segs);
        const float *W0 = ns.kernel[0], *b0 = ns.bias[0], *bn_on_load = nullptr, *centre = nullptr;
        if (bn_s) {
            float *stats = p.stats_s + (size_t)t * 2 * p.in_s;
            gnn::Seg dyn[2] = {segs[0], segs[p.with_labels ? 2 : 1]};
            TRY(colstats_segs(gate, dyn, 2, p.N, stats, stats + p.in_s, p.part, st));
            if (fold_s) {                        // thin first layer (<= 4 units): the thin-dense kernel wants folded weights
                float *Wf = p.Wf_s + (size_t)t * p.in_s * p.H1s, *bf = p.bf_s + (size_t)t * p.H1s;
                TRY(fold_with_stats(ns, stats, Wf, bf, st, true));
                W0 = Wf; b0 = bf; centre = stats;
            } else bn_on_load = stats;           // BatchNormalization applied as the first layer stages its inputs: no fold launch
        }
        // Dropout layers (ABI 8): fresh masks every iteration (call = t); a layer behind the last Dense makes the NEW STATE the dropped-out
        // output (the tape holds it; the network's own output stays in its layer buffer for the backward recompute)
        DropRun drs{&ta.drop_state[0], ta.drop_seed, t, {}};
        for (int q = 0; q <= ns.n_layers; ++q) drs.buf[q] = p.cs.dropbuf[q];
        const bool drop_last = drop_at(&drs, ns.n_layers);
        if (drop_last) drs.buf[ns.n_layers] = s_n;
        float *hs[GNN_MAX_LAYERS];
        for (int l = 0; l < ns.n_layers; ++l) hs[l] = (l == ns.n_layers - 1 && !drop_last) ? s_n : p.cs.hid[l];
        PredFuse pf{s_t, p.S, a.state_threshold, p.flags + t + 1, p.k_dev, (float)(t + 1), false};
        TRY(forward_layers(ns, segs, n, p.N, W0, b0, hs, gate, st, &pf, bn_on_load, centre, &drs));
        if (!pf.fused) TRY(launch_converge(gate, s_n, s_t, p.N, p.S, p.S, p.S, a.state_threshold, p.flags + t + 1, p.k_dev, (float)(t + 1), st));
    }
    float k_f2[3] = {0.0f, 0.0f, 0.0f};
    HIP_OK(hipMemcpyAsync(k_f2, p.k_dev, 3 * sizeof(float), hipMemcpyDeviceToHost, st));
    HIP_OK(hipStreamSynchronize(st));                                  // the one host synchronisation of the step
    const int k = (int)k_f2[0];
    *ta.k_host = k;
    if (k_f2[1] == 1.0f) return fail("tile_node_begin: an arc of `adjacency` leaves its tile");
    if (k_f2[1] != 0.0f) return fail("a workgroup of the persistent training kernel never arrived at a grid barrier (not resident?)");
    if (k < 0 || k > p.K) return fail("iteration count %d out of range", k);
    if (k_f2[2] != 0.0f) p.head_fast = false;      // a permuted / repeated out_index of length n_nodes: the general head (gathers, scatter-add)
    const float *state_k = p.states + (size_t)k * NS;
    if (p.ldS == p.S) HIP_OK(hipMemcpyAsync(ta.state, state_k, sizeof(float) * NS, hipMemcpyDeviceToDevice, st));
    else TRY(launch_copy2d(nullptr, state_k, p.ldS, ta.state, p.S, p.N, p.S, p.S, st));
    // The moving averages of BatchNormalization (one update per executed call).  On the persistent small-graph path they wait until the
    // backward launch has passed its grid barriers and are gated by the step's validity word: a step whose backward failed changes nothing.
    auto moving_state = [&](const int *gate) -> int {
        if (bn_s && k > 0) {
            gnn::k_bn_moving_multi<<<cdiv(p.in_s, 256), 256, 0, st>>>(p.stats_s, 2 * p.in_s, k, p.in_s, const_cast<float *>(ns.bn_mean),
                                                                    const_cast<float *>(ns.bn_var), ta.bn_momentum, gate);
            LAUNCH_OK();
        }
        return 0;
    };
    bool moving_output_pending = false;
    if (!p.small) TRY(moving_state(nullptr));

    // ---- output network, training mode ---------------------------------------------------------------------------------------------
    gnn::Seg osegs[GNN_MAX_SEGS];
    int nos = 0, ocol = 0;
    int bn_req_off[2] = {0, 0};                 // column offsets of the state segments inside the output net's input
    const int *bn_req_idx[2] = {nullptr, nullptr};
    int n_state_segs = 0;
    if (p.M > 0) {
        if (a.focus == GNN_FOCUS_ARC) {
            k_arc_endpoints<<<cdiv(p.M, 256), 256, 0, st>>>(a.out_index, a.arc_src, a.arc_dst, p.M, p.isrc, p.idst);
            LAUNCH_OK();
            const int *ends[2] = {p.isrc, p.idst};
            for (int e = 0; e < 2; ++e) {
                bn_req_off[n_state_segs] = ocol; bn_req_idx[n_state_segs++] = ends[e];
                osegs[nos++] = gnn::Seg{state_k, ends[e], p.ldS, p.S, ocol}; ocol += p.S;
                if (p.with_labels) { osegs[nos++] = gnn::Seg{a.nodes, ends[e], a.ld_nodes, p.L, ocol}; ocol += p.L; }
            }
            if (p.A > 0) { osegs[nos++] = gnn::Seg{a.arc_labels, a.out_index, a.ld_arcs, p.A, ocol}; ocol += p.A; }
        } else {
            bn_req_off[0] = 0; bn_req_idx[0] = a.out_index; n_state_segs = 1;
            osegs[nos++] = gnn::Seg{state_k, a.out_index, p.ldS, p.S, ocol}; ocol += p.S;
            if (p.with_labels) { osegs[nos++] = gnn::Seg{a.nodes, a.out_index, a.ld_nodes, p.L, ocol}; ocol += p.L; }
        }
    }
    DropRun dro{&ta.drop_output, ta.drop_seed, 0, {}};
    for (int q = 0; q <= no.n_layers; ++q) dro.buf[q] = p.co.dropbuf[q];
    const bool drop_o_last = drop_at(&dro, no.n_layers);          // (a Dropout layer behind the last Dense: the network's output is its dropped-out copy)
    if (drop_o_last && !p.pooled) dro.buf[no.n_layers] = ta.y_pred;
    float *ohs[GNN_MAX_LAYERS];
    for (int l = 0; l < no.n_layers; ++l) ohs[l] = (l == no.n_layers - 1 && !p.pooled && !drop_o_last) ? ta.y_pred : p.co.hid[l];
    float *out_nodes = drop_o_last ? dro.buf[no.n_layers] : ohs[no.n_layers - 1];
    if (p.M > 0) {
        const float *W0 = no.kernel[0], *b0 = no.bias[0];
        if (bn_o) {
            if (p.head_fast && bn_s && k >= 1) {
                // Every node is an output row: the column statistics of [state_k | labels] are already on the tape - the state's were
                // left by the launch that wrote it (k_train_fwd's epilogue, slot k), the labels' are the state network's constant
                // BatchNorm columns S .. S + L (one pass over the packed constants line) - four small copies instead of four passes
                // over a million rows (0.76 ms of a C4-size step).
                const float *slot = p.stats_s + (size_t)k * 2 * p.in_s;
                HIP_OK(hipMemcpyAsync(p.stats_o, slot, sizeof(float) * p.S, hipMemcpyDeviceToDevice, st));
                HIP_OK(hipMemcpyAsync(p.stats_o + p.in_o, slot + p.in_s, sizeof(float) * p.S, hipMemcpyDeviceToDevice, st));
                if (p.with_labels) {
                    HIP_OK(hipMemcpyAsync(p.stats_o + p.S, p.stats_tpl + p.S, sizeof(float) * p.L, hipMemcpyDeviceToDevice, st));
                    HIP_OK(hipMemcpyAsync(p.stats_o + p.in_o + p.S, p.stats_tpl + p.in_s + p.S, sizeof(float) * p.L, hipMemcpyDeviceToDevice, st));
                }
            } else
            TRY(colstats_segs(nullptr, osegs, nos, p.M, p.stats_o, p.stats_o + p.in_o, p.part, st));
            TRY(fold_with_stats(no, p.stats_o, p.Wf_o, p.bf_o, st, true));      // (centred: the first layer subtracts the means on load)
            if (p.small) moving_output_pending = true;
            else {
                gnn::k_bn_moving_multi<<<cdiv(p.in_o, 256), 256, 0, st>>>(p.stats_o, 2 * p.in_o, 1, p.in_o, const_cast<float *>(no.bn_mean),
                                                                        const_cast<float *>(no.bn_var), ta.bn_momentum);
                LAUNCH_OK();
            }
            W0 = p.Wf_o; b0 = p.bf_o;
        }
        TRY(forward_layers(no, osegs, nos, p.M, W0, b0, ohs, nullptr, st, nullptr, nullptr, bn_o ? p.stats_o : nullptr, &dro));
    }
    if (p.pooled) TRY(launch_aggregate(nullptr, a.nodegraph, out_nodes, p.T, p.T, ta.y_pred, p.T, st));
    if (fwd_only) {
        // the forward alone: the moving averages the persistent small-graph path keeps back for the step's validity word are due now
        if (p.small) TRY(moving_state(nullptr));
        if (moving_output_pending) {
            gnn::k_bn_moving_multi<<<cdiv(p.in_o, 256), 256, 0, st>>>(p.stats_o, 2 * p.in_o, 1, p.in_o, const_cast<float *>(no.bn_mean),
                                                                    const_cast<float *>(no.bn_var), ta.bn_momentum);
            LAUNCH_OK();
        }
        return 0;
    }
    // ---- loss and its gradient ----------------------------------------------------------------------------------------------------
    gnn::k_loss_grad<<<cdiv(p.R, 256), 256, 0, st>>>(ta.loss_kind, ta.targets, ta.y_pred, ta.sample_weight, p.R, p.T, p.dpred, p.loss_rows);
    LAUNCH_OK();
    if (p.R > 65536) {             // (one block summing a million rows took 0.9 ms)
        gnn::k_sum_partials<<<256, 256, 0, st>>>(p.loss_rows, p.R, p.loss_part);
        LAUNCH_OK();
        gnn::k_sum_scale<<<1, 256, 0, st>>>(p.loss_part, 256, 1.0f / (float)p.R, ta.loss);
    } else gnn::k_sum_scale<<<1, 256, 0, st>>>(p.loss_rows, p.R, 1.0f / (float)p.R, ta.loss);
    LAUNCH_OK();
    float *G_out = p.dpred;
    if (p.pooled) { TRY(launch_aggregate(nullptr, ta.nodegraph_by_source, p.dpred, p.T, p.T, p.G_out, p.T, st)); G_out = p.G_out; }

    // ---- backward: output network, then the k iterations -----------------------------------------------------------------------------------
    // Large graphs (round 5): G_{t-1} leaves the transposed aggregate as dZ_{t-1} = G_{t-1} (.) act'(state_t) (k_aggregate_dz), so that the two
    // dense kernels of an iteration read dZ alone; the first dZ comes from the output head (k_head_dx, or one k_act_grad pass behind the
    // general head).  And iteration 0 needs no input gradient at all: nothing consumes d loss / d state_0.
    const bool dzpath = p.big && p.Kc > 0 && p.Kc < 32 && train_wgrad_enabled() && train_dz_enabled() && (p.S == 16 || p.S == 32 || p.S == 64) &&
                        ns.activation[0] != GNN_ACT_SOFTMAX;
    if (!p.head_fast) HIP_OK(hipMemsetAsync(p.G_state, 0, sizeof(float) * (size_t)p.N * p.S, st));
    if (p.head_fast) {
        TRY(head_backward(p, a, state_k, out_nodes, G_out, bn_o ? p.stats_o : nullptr, st, (dzpath && k > 0) ? (int)ns.activation[0] : -1));
    } else if (p.M > 0) {
        TRY(net_backward(p.co, osegs, nos, ohs, G_out, p.T, p.M, bn_o ? p.stats_o : nullptr, false, p.dx_o_all, p.in_o, p.part, st, -1, 0, &dro));
        gnn::BnGradReq rq[2];
        for (int i = 0; i < n_state_segs; ++i)
            rq[i] = gnn::BnGradReq{p.dx_o_all + bn_req_off[i], p.in_o, state_k, p.ldS, bn_req_idx[i], p.S, bn_req_off[i]};
        TRY(bn_input_grads(no, p.co, p.stats_o, rq, n_state_segs, p.M, st));
        for (int i = 0; i < n_state_segs; ++i) {
            gnn::k_scatter_add_rows<<<std::min(cdiv((long)p.M * p.S, 256), 256 * 16), 256, 0, st>>>(p.dx_o_all + bn_req_off[i], p.in_o, bn_req_idx[i],
                                                                                                   p.M, p.S, p.G_state, p.S);
            LAUNCH_OK();
        }
    } else {
        TRY(zero_grads(no, ta.grad_output, st));
    }
    if (k == 0) TRY(zero_grads(ns, ta.grad_state, st));
    if (p.small && k > 0) {
        // the k iterations of back-propagation in one persistent launch; every workgroup leaves its share of the kernel gradient
        gnn::TrainSmallBwd ba;
        memset(&ba, 0, sizeof(ba));
        const gnn_csr_t &cs_ = ta.adjacency_by_source;
        const bool unit_w = !a.adjacency.w;       // entries depend on the destination only: scale the agg-half once per row, walk unit weights
        ba.N = p.N; ba.S = p.SPs; ba.Sw = p.S; ba.k = k;
        ba.rowptr_s = cs_.rowptr; ba.src_s = cs_.src; ba.w_s = unit_w ? nullptr : cs_.w; ba.row_scale_s = unit_w ? nullptr : cs_.row_scale;
        ba.row_scale = unit_w ? a.adjacency.row_scale : nullptr;
        ba.states = p.states; ba.agg = p.agg; ba.stats = p.stats_s; ba.in_s = p.in_s; ba.off_agg = p.off_agg;
        ba.cs = const_segs_of(a, p);
        ba.W = ns.kernel[0]; ba.gamma = bn_s ? ns.bn_gamma : nullptr; ba.beta = ns.bn_beta; ba.eps = ns.bn_eps; ba.act = ns.activation[0];
        ba.G0 = p.G_state; ba.dxa = p.dx_s_all; ba.bar = p.sm_bar + 2; ba.part = p.sm_part; ba.partW = p.sm_partW;
        ba.partBN = p.sm_partBN;
        ba.inv_n = 1.0f / (float)p.N; ba.err = p.k_dev; ba.wait_ticks = gnn::wait_ticks();
        if (const char *e = getenv("GNN_DEBUG_FAIL_BWD")) { if (e[0] == '1') ba.wait_ticks = 0; }      // (test hook: every barrier wait of THIS launch expires at once)
        switch (p.SPs) {
            case 16: TRY(launch_train_small_bwd_sq<1>(ba, tiles, p.n_wg, ba.w_s != nullptr, st)); break;
            case 32: TRY(launch_train_small_bwd_sq<2>(ba, tiles, p.n_wg, ba.w_s != nullptr, st)); break;
            default: TRY(launch_train_small_bwd_sq<4>(ba, tiles, p.n_wg, ba.w_s != nullptr, st)); break;
        }
        const int n = (p.in_s + 1) * p.S;              // every workgroup's [kernel | bias] share, summed in workgroup order
        gnn::k_reduce_partials<<<cdiv(n, 64), 256, 0, st>>>(p.sm_partW, p.n_wg, n, ta.grad_state.dkernel[0], 0, 1.0f, p.in_s * p.S, ta.grad_state.dbias[0]);
        LAUNCH_OK();
        if (bn_s) {                                    // ... and its [d gamma | d beta] share
            gnn::k_reduce_partials<<<cdiv(2 * p.in_s, 64), 256, 0, st>>>(p.sm_partBN, p.n_wg, 2 * p.in_s, ta.grad_state.dgamma, 0, 1.0f, p.in_s, ta.grad_state.dbeta);
            LAUNCH_OK();
        }
    }
    if (dzpath && k > 0 && !p.head_fast) TRY(act_grad_inplace(p.G_state, p.S, p.states + (size_t)k * NS, p.S, p.N, p.S, ns.activation[0], st));
    for (int t = k - 1; t >= 0 && !p.small; --t) {
        const float *s_t = p.states + (size_t)t * NS;
        float *s_n = p.states + (size_t)(t + 1) * NS;
        const float *agg_t = p.agg + (p.agg_taped ? (size_t)t * NS : 0);
        if (!p.agg_taped) TRY(launch_aggregate(nullptr, a.adjacency, s_t, p.S, p.S, p.agg, p.S, st));
        const int n = state_segs(a, p, t, segs);
        const float *stats = bn_s ? p.stats_s + (size_t)t * 2 * p.in_s : nullptr;
        DropRun drs{&ta.drop_state[0], ta.drop_seed, t, {}};                      // (iteration t's masks again: the same keys)
        for (int q = 0; q <= ns.n_layers; ++q) drs.buf[q] = p.cs.dropbuf[q];
        const bool drop_any = ta.drop_state[0].n > 0;
        float *hs[GNN_MAX_LAYERS];
        for (int l = 0; l < ns.n_layers; ++l) hs[l] = (l == ns.n_layers - 1 && !drop_any) ? s_n : p.cs.hid[l];
        if (ns.n_layers > 1 || drop_any) {         // hidden activations are not on the tape: recompute them (the last layer's are, unless a Dropout layer follows it)
            const bool folded = bn_s && fold_s;
            const float *W0 = folded ? p.Wf_s + (size_t)t * p.in_s * p.H1s : ns.kernel[0], *b0 = folded ? p.bf_s + (size_t)t * p.H1s : ns.bias[0];
            gnn_mlp_t head = ns; head.n_layers = drop_any ? ns.n_layers : ns.n_layers - 1;
            TRY(forward_layers(head, segs, n, p.N, W0, b0, hs, nullptr, st, nullptr, (bn_s && !fold_s) ? stats : nullptr, folded ? stats : nullptr, &drs, true));
        }
        // 'average' / 'sum' / 'normalized' entries depend on the destination only (a.adjacency carries one scale per row): the large-
        // graph kernel scales the agg-half of a row's gradient once, and the transposed aggregate walks UNIT weights (no 4 bytes per arc)
        const bool unit_w = p.big && !a.adjacency.w;
        const bool wgrad = p.big && p.Kc > 0 && p.Kc < 32 && train_wgrad_enabled();
        if (p.big) {
            gnn::TrainBwdArgs ba;                                    // d loss / d [state | agg] of this iteration (not taken at t == 0: nothing consumes it)
            memset(&ba, 0, sizeof(ba));
            ba.M = p.N; ba.dZ = p.G_state; ba.ldz = p.S;            // (net_backward: the activation gradient ran in place; dzpath: dZ arrived as such)
            if (wgrad && !dzpath) { ba.Y = s_n; ba.act = ns.activation[0]; }   // (G is untouched: dZ is formed as the rows arrive)
            ba.W = ns.kernel[0]; ba.ldw = p.H1s; ba.H = p.H1s; ba.S = p.S; ba.wrow_state = 0; ba.wrow_agg = p.off_agg;
            ba.state = s_t; ba.ld_state = p.S; ba.agg = agg_t; ba.ld_agg = p.S;
            if (bn_s) { ba.gamma = ns.bn_gamma; ba.mean = stats; ba.var = stats + p.in_s; ba.m1 = p.cs.m1; ba.m2 = p.cs.m2; ba.eps = ns.bn_eps; }
            ba.defer_state_bn = (dzpath && bn_s) ? 1 : 0;            // (k_aggregate_dz adds the rest of the state half's BatchNorm gradient: it reads state_t anyway)
            ba.agg_row_scale = unit_w ? a.adjacency.row_scale : nullptr;
            ba.dx = p.dx_s_all; ba.ld_dx = p.kdx_s;
            bool dx_done = false;
            if (wgrad) {
                // P = X^T dZ and q on the matrix cores straight from the rows (dZ = G (.) act'(s_n) formed on the way), then the same
                // reduction and parameter-gradient kernels as net_backward
                gnn::TrainWgradArgs wa;
                memset(&wa, 0, sizeof(wa));
                const bool wg_b6 = train_wgrad_b6_enabled() && (p.S == 64 || p.S == 32);
                const int n_wg = std::min(std::min((wg_b6 ? 1 : 2) * device_cus(), BIG_WGRAD_BLOCKS), cdiv(p.N, 64));
                wa.M = p.N; wa.rows_per_wg = wg_b6 ? cdiv(cdiv(p.N, n_wg), 64) * 64 : cdiv(cdiv(p.N, n_wg), 16) * 16;
                wa.G = p.G_state; wa.Y = dzpath ? nullptr : s_n; wa.act = dzpath ? GNN_ACT_LINEAR : ns.activation[0];      // (dzpath: G_state holds dZ)
                wa.state = s_t; wa.agg = agg_t; wa.xc = p.xc;
                wa.K = p.in_s; wa.wrow_state = 0; wa.wrow_agg = p.off_agg; wa.Kc = p.Kc; wa.cs = p.cc;
                wa.part = p.part_w;
                wa.mean = stats;                       // (BatchNormalization: the rows are centred as they arrive, P arrives as P - mean q^T)
                const int grid = cdiv(p.N, wa.rows_per_wg);
                if (wg_b6 && dzpath && t > 0 && train_fused_bwd_enabled() && launch_train_wgrad_dx_b6(wa, ba, p.S, grid, st)) dx_done = true;    // both products from one pass over the rows
                else if (wg_b6 && launch_train_wgrad_b6(wa, p.S, grid, st)) {
                } else if (!launch_train_wgrad32(wa, p.S, grid, st)) {       // (S = 16, or an activation without an instance: the 16x16x4 kernel)
                    switch (p.S) {
                        case 16: gnn::k_train_wgrad<1><<<grid, 256, 0, st>>>(wa); break;
                        case 32: gnn::k_train_wgrad<2><<<grid, 256, 0, st>>>(wa); break;
                        default: gnn::k_train_wgrad<4><<<grid, 256, 0, st>>>(wa); break;
                    }
                }
                LAUNCH_OK();
                const int nP = p.in_s * p.S + p.S;
                gnn::k_reduce_partials<<<cdiv(nP, 64), 256, 0, st>>>(p.part_w, grid, nP, p.cs.P, 0, 1.0f, p.in_s * p.S, p.cs.q);
                LAUNCH_OK();
                gnn::k_first_layer_param_grads<<<p.in_s, 64, 0, st>>>(
                    p.cs.P, p.cs.q, ns.kernel[0], p.in_s, p.S, bn_s ? ns.bn_gamma : nullptr, ns.bn_beta, stats, stats ? stats + p.in_s : nullptr, ns.bn_eps,
                    1.0f / (float)p.N, ta.grad_state.dkernel[0], ta.grad_state.dbias[0], ta.grad_state.dgamma, ta.grad_state.dbeta,
                    bn_s ? p.cs.m1 : nullptr, bn_s ? p.cs.m2 : nullptr, t != k - 1 ? 1 : 0, 1, stats ? 1 : 0);
                LAUNCH_OK();
            } else TRY(net_backward(p.cs, segs, n, hs, p.G_state, p.S, p.N, stats, t != k - 1, nullptr, 0, p.part, st, p.off_agg));
            if (t == 0) break;                                       // nothing consumes d loss / d state_0: no input gradient, no transposed aggregate
            if (!dx_done) TRY(launch_train_bwd_dx(ba, p.S, st));
        } else {
            TRY(net_backward(p.cs, segs, n, hs, p.G_state, p.S, p.N, stats, t != k - 1, t > 0 ? p.dx_s_all : nullptr, t > 0 ? p.kdx_s : 0, p.part, st, p.off_agg, 0, &drs));
            if (t == 0) break;                                       // (as above)
            gnn::BnGradReq rq[2] = {gnn::BnGradReq{p.dx_s_all, p.kdx_s, s_t, p.S, nullptr, p.S, 0},
                                    gnn::BnGradReq{p.dx_s_all + p.S, p.kdx_s, agg_t, p.S, nullptr, p.S, p.off_agg}};
            TRY(bn_input_grads(ns, p.cs, stats, rq, 2, p.N, st));
        }
        if (dzpath) {   // dZ_{t-1} = (dx_state' + Adj . dx_agg + the deferred BatchNorm term) (.) act'(state_t)   (arcs walked by source)
            gnn_csr_t c = ta.adjacency_by_source;
            if (unit_w) { c.w = nullptr; c.row_scale = nullptr; }
            gnn::AggDzArgs z;
            memset(&z, 0, sizeof(z));
            z.Y = s_t; z.ldy = p.S; z.wrow_state = 0;
            if (bn_s) { z.gamma = ns.bn_gamma; z.var = stats + p.in_s; z.mean = stats; z.m1 = p.cs.m1; z.m2 = p.cs.m2; z.eps = ns.bn_eps; }
            if (!launch_aggregate_dz(c, p.dx_s_all + p.S, p.kdx_s, p.G_state, p.S, p.dx_s_all, p.kdx_s, z, ns.activation[0], p.S, st))
                return fail("k_aggregate_dz: no instance for state width %d / activation %d", p.S, ns.activation[0]);
            LAUNCH_OK();
        } else {   // G_state = d state (own) + Adj . d agg   (arcs walked by source)
            gnn_csr_t c = ta.adjacency_by_source;
            if (unit_w) { c.w = nullptr; c.row_scale = nullptr; }
            const float *Xa = p.dx_s_all + p.S;
            const bool vec = (p.S == 16 || p.S == 32 || p.S == 64 || p.S == 128) && p.kdx_s % 4 == 0 &&
                             ((reinterpret_cast<uintptr_t>(Xa) | reinterpret_cast<uintptr_t>(p.dx_s_all) | reinterpret_cast<uintptr_t>(p.G_state)) & 15) == 0;
            if (vec) {      // whole 16-B row pieces, 8 source rows in flight (the scalar walk below is one dependent chain per arc)
                const int lpr = p.S / 4, groups = 256 / lpr, grid = std::min(cdiv(p.N, groups), 256 * 16);
                const bool buf = gather_buf_ok(c, p.kdx_s);
#define AGGV_(L, W_, B_) gnn::k_aggregate_vec<L, W_, B_><<<grid, 256, 0, st>>>(nullptr, c.n_dst, c.rowptr, c.src, c.w, c.row_scale, Xa, p.kdx_s, p.G_state, p.S, p.dx_s_all, p.kdx_s)
#define AGGV(L) (buf ? (c.w ? AGGV_(L, true, true) : AGGV_(L, false, true)) : (c.w ? AGGV_(L, true, false) : AGGV_(L, false, false)))
                switch (lpr) { case 4: AGGV(4); break; case 8: AGGV(8); break; case 16: AGGV(16); break; default: AGGV(32); break; }
#undef AGGV
#undef AGGV_
                LAUNCH_OK();
            } else {
                int G = 4;
                while (G < p.S && G < 64) G <<= 1;
                const int groups = 256 / G, grid = std::min(cdiv(p.N, groups), 256 * 16);
#define AGGA(GG) gnn::k_aggregate_add<GG><<<grid, 256, 0, st>>>(c.n_dst, c.rowptr, c.src, c.w, c.row_scale, Xa, p.kdx_s, p.S, \
                                                                 p.dx_s_all, p.kdx_s, p.G_state, p.S)
                switch (G) { case 4: AGGA(4); break; case 8: AGGA(8); break; case 16: AGGA(16); break; case 32: AGGA(32); break; default: AGGA(64); break; }
#undef AGGA
                LAUNCH_OK();
            }
        }
    }
    if (ta.average_st_grads && k > 0) TRY(scale_grads(ns, ta.grad_state, 1.0f / (float)k, st));
    // ---- the step's validity word, and what waited for it ----------------------------------------------------------------------------------
    gnn::k_grads_ok<<<1, 1, 0, st>>>(p.small ? p.k_dev : nullptr, p.grads_ok);
    LAUNCH_OK();
    if (p.small) {
        TRY(moving_state(p.grads_ok));
        if (moving_output_pending) {
            gnn::k_bn_moving_multi<<<cdiv(p.in_o, 256), 256, 0, st>>>(p.stats_o, 2 * p.in_o, 1, p.in_o, const_cast<float *>(no.bn_mean),
                                                                    const_cast<float *>(no.bn_var), ta.bn_momentum, p.grads_ok);
            LAUNCH_OK();
        }
    }
    return 0;
}

#ifdef GNN_TS_TIMELINE
// phase times of workgroup 0 in the last persistent training launches: out[2][16] (forward, backward), ticks of 10 ns
int gnn_ts_phase_times(unsigned long long *out32) {
    return hipMemcpyFromSymbol(out32, HIP_SYMBOL(gnn::g_ts_phase), 2 * 16 * 8) == hipSuccess ? 0 : 1;
}
#endif

}  // extern "C"
