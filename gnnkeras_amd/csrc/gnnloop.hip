// libgnnloop.so — host side of the C ABI declared in include/gnnloop.h (gfx950 only).
//
// gnn_loop_forward enqueues, on the caller's stream and with no host<->device synchronisation:
//   setup   : BN folding of the first Dense of every network; ArcNode scatter-add (reference GNN.py:254) and neighbour
//             label aggregate (:258) as CSR walks; the iteration-invariant part of the first state layer folded into
//             a per-node constant  C = [labels | agg labels | agg arcs] . W1[const rows] + b1   (SURVEY §7);
//   loop    : `max_iteration` gated iterations (fused kernel, or aggregate + dense + predicate when un-fused);
//             iteration i runs only if flags[i] != 0 and raises flags[i+1] when any node is still moving;
//   output  : apply_filters (:239-242 / :317-330) as row-index segments of the output network's first layer,
//             the output MLP on the f32 matrix cores, optional graph pooling (:341-346).
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stddef.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/gnnloop.h"
#include "kernels_general.hpp"
#include "kernel_state_fused2.hpp"
#include "kernel_state_fused4.hpp"
#include "kernel_state_small.hpp"
#include "kernel_state_mid.hpp"
#include "kernel_state_lds.hpp"
#include "kernel_state_wide.hpp"
#include "kernels_batch.hpp"
#include "kernels_setup.hpp"
#include "kernel_rowdense.hpp"
#include "kernel_state_xwide.hpp"

namespace {

thread_local char g_err[512] = "";

int fail(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return 1;
}

#define HIP_OK(expr)                                                                                    \
    do {                                                                                                \
        hipError_t e_ = (expr);                                                                         \
        if (e_ != hipSuccess) return fail("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)
#define LAUNCH_OK() HIP_OK(hipGetLastError())
#define TRY(expr)                \
    do {                         \
        int rc_ = (expr);        \
        if (rc_) return rc_;     \
    } while (0)

#define FUSED_OK(expr)                                                                                  \
    do {                                                                                                \
        if (expr) return fail("fused state kernel launch failed: %s", hipGetErrorString(hipGetLastError())); \
    } while (0)

inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }
inline int round_up(int a, int b) { return (a + b - 1) / b * b; }

// padded leading dimension of the internal state buffers: rows start 16-byte aligned; power of two in [16, 128] so a
// node's row is covered by 4..32 lanes of one wave loading 16 B each and by whole 16-column MFMA tiles.
int state_ld(int S) {
    if (S <= 128) {
        int p = 16;
        while (p < S) p <<= 1;
        return p;
    }
    return round_up(S, 4);
}

// --- workspace carving: same code computes the size (base == nullptr) and hands out the pointers --------------------
struct Carver {
    char *base;
    size_t off = 0;
    explicit Carver(void *b) : base((char *)b) {}
    template <typename T>
    T *take(size_t count) {
        off = (off + 255) & ~(size_t)255;
        T *p = base ? (T *)(base + off) : nullptr;
        off += count * sizeof(T);
        return p;
    }
};

int max_units(const gnn_mlp_t &m) {
    int h = 1;
    for (int i = 0; i < m.n_layers; ++i) h = std::max(h, (int)m.units[i]);
    return h;
}

int check_mlp(const gnn_mlp_t &m, const char *name, bool need_ptrs) {
    if (m.n_layers < 1 || m.n_layers > GNN_MAX_LAYERS) return fail("%s: n_layers %d out of [1,%d]", name, m.n_layers, GNN_MAX_LAYERS);
    if (m.in_dim < 1) return fail("%s: in_dim %d < 1", name, m.in_dim);
    for (int i = 0; i < m.n_layers; ++i) {
        if (m.units[i] < 1) return fail("%s: layer %d has %d units", name, i, m.units[i]);
        if (m.activation[i] < 0 || m.activation[i] > GNN_ACT_SOFTMAX) return fail("%s: layer %d unknown activation %d", name, i, m.activation[i]);
        if (need_ptrs && (!m.kernel[i] || !m.bias[i])) return fail("%s: layer %d kernel/bias is NULL", name, i);
    }
    if (need_ptrs && m.has_bn && (!m.bn_gamma || !m.bn_beta || !m.bn_mean || !m.bn_var)) return fail("%s: BatchNormalization arrays are NULL", name);
    return 0;
}

int check_csr(const gnn_csr_t &c, const char *name, int n_dst, int n_src) {
    if (c.n_dst != n_dst || c.n_src != n_src) return fail("%s: shape (%d x %d)^T expected, got n_src=%d n_dst=%d", name, n_src, n_dst, c.n_src, c.n_dst);
    if (c.nnz < 0) return fail("%s: nnz < 0", name);
    if (!c.rowptr) return fail("%s: rowptr is NULL", name);
    if (c.nnz > 0 && !c.src) return fail("%s: src is NULL", name);
    return 0;
}

// --- launchers -------------------------------------------------------------------------------------------------------
int launch_aggregate(const int *gate, const gnn_csr_t &c, const float *X, int ldx, int F, float *out, int ldo, hipStream_t st) {
    if (c.n_dst == 0 || F == 0) return 0;
    const bool aligned = ldx % 4 == 0 && ldo % 4 == 0 && ((reinterpret_cast<uintptr_t>(X) | reinterpret_cast<uintptr_t>(out)) & 15) == 0;
    if (aligned && (F == 16 || F == 32 || F == 64 || F == 128 || (F > 128 && F % 4 == 0))) {
        // 16-byte row pieces; rows wider than 128 floats go in column blocks of 128, 64, .. 4 (one launch each: the CSR is
        // walked again, the row bytes are not)
        for (int off = 0; off < F;) {
            int w = 128;
            while (w > F - off) w >>= 1;
            const int lpr = w / 4, groups = 256 / lpr;
            const int grid = std::min(cdiv(c.n_dst, groups), 256 * 16);
            const float *Xb = X + off; float *ob = out + off;
            // (buffer form: the eight source ids, then the eight rows of a trip requested together - kernels_general.hpp gather_sum8; 4 GiB windows)
            static const bool buf_on = !(getenv("GNN_GATHER_BUF") && getenv("GNN_GATHER_BUF")[0] == '0');
            const bool buf = buf_on && (size_t)c.n_src * (size_t)ldx * 4 < 0xFFFFFFF0ull && (size_t)c.nnz * 4 < 0xFFFFFFF0ull;
#define AGGV_(L, W_, B_) gnn::k_aggregate_vec<L, W_, B_><<<grid, 256, 0, st>>>(gate, c.n_dst, c.rowptr, c.src, c.w, c.row_scale, Xb, ldx, ob, ldo)
#define AGGV(L) (buf ? (c.w ? AGGV_(L, true, true) : AGGV_(L, false, true)) : (c.w ? AGGV_(L, true, false) : AGGV_(L, false, false)))
            switch (lpr) {
                case 1: AGGV(1); break;
                case 2: AGGV(2); break;
                case 4: AGGV(4); break;
                case 8: AGGV(8); break;
                case 16: AGGV(16); break;
                default: AGGV(32); break;
            }
#undef AGGV
#undef AGGV_
            LAUNCH_OK();
            off += w;
        }
        return 0;
    }
    int G = 4;
    while (G < F && G < 64) G <<= 1;
    const int groups = 256 / G;
    const int grid = std::min(cdiv(c.n_dst, groups), 256 * 16);
    if (G <= 16 && c.n_dst >= 4096 && (size_t)c.n_src * ldx * 4 < ((size_t)1 << 32) && (size_t)c.nnz * 4 < ((size_t)1 << 32)) {
        // narrow label rows of a large graph: the whole CSR row in flight (k_aggregate_narrow)
#define AGGN(GG) (c.w ? gnn::k_aggregate_narrow<GG, true><<<grid, 256, 0, st>>>(gate, c.n_dst, c.rowptr, c.src, c.w, c.row_scale, X, ldx, F, out, ldo) \
                      : gnn::k_aggregate_narrow<GG, false><<<grid, 256, 0, st>>>(gate, c.n_dst, c.rowptr, c.src, c.w, c.row_scale, X, ldx, F, out, ldo))
        switch (G) { case 4: AGGN(4); break; case 8: AGGN(8); break; default: AGGN(16); break; }
#undef AGGN
        LAUNCH_OK();
        return 0;
    }
#define AGG(GG) gnn::k_aggregate<GG><<<grid, 256, 0, st>>>(gate, c.n_dst, c.rowptr, c.src, c.w, c.row_scale, X, ldx, F, out, ldo)
    switch (G) {
        case 4: AGG(4); break;
        case 8: AGG(8); break;
        case 16: AGG(16); break;
        case 32: AGG(32); break;
        default: AGG(64); break;
    }
#undef AGG
    LAUNCH_OK();
    return 0;
}

// thin outputs (class scores): 16 lanes per row, no MFMA tile to fill; the weights must fit 48 KB of LDS
bool thin_dense_applies(const gnn::SegDenseArgs &a) {
    if (a.H > 4) return false;
    int K = 0;
    for (int s = 0; s < a.nseg; ++s) K = std::max(K, a.seg[s].wrow + a.seg[s].width);
    // (what the kernel asks for: the weights rounded up to whole 16-byte pieces, the column means, a pad - inside the 64 KB a launch gets
    //  without raising the kernel's dynamic-LDS limit: at H = 1 the weights alone may be 48 KB of it)
    const size_t lds = ((((size_t)K * a.H + 3) & ~(size_t)3) + (size_t)K + 4) * sizeof(float);
    return (size_t)K * a.H * sizeof(float) <= 48 * 1024 && lds <= 64 * 1024;
}

int device_cus();

// One contiguous input matrix of 16 .. 64 columns at large M (the layers behind the first): rows straight into the matrix cores
// (kernel_rowdense.hpp).  GNN_ROWDENSE=0 keeps k_segdense.
bool rowdense_applies(const gnn::SegDenseArgs &a) {
    static int off = -1;
    if (off < 0) { const char *e = getenv("GNN_ROWDENSE"); off = (e && e[0] == '0') ? 1 : 0; }
    if (off || a.M < 32768 || a.nseg != 1 || a.addend || a.out_rowidx || a.in_gamma || a.in_center || a.act == GNN_ACT_SOFTMAX) return false;
    const gnn::Seg &s = a.seg[0];
    if (s.rowidx || s.wrow != 0 || s.width < 8 || s.width > 64 || s.width % 4 || a.H < 8 || a.H > 64 || a.H % 4) return false;
    if (s.ld % 4 || a.ldy % 4 || (a.pred_flag && a.ld_pred % 4)) return false;
    if (((reinterpret_cast<uintptr_t>(s.ptr) | reinterpret_cast<uintptr_t>(a.Y) | reinterpret_cast<uintptr_t>(a.pred_old)) & 15) != 0) return false;
    if ((size_t)a.M * std::max(std::max(s.ld, a.ldy), a.ld_pred) * 4 >= ((size_t)1 << 32)) return false;
    return true;
}

template <int KQ>
int launch_rowdense_k(const gnn::RowDenseArgs &r, int nct, int grid, hipStream_t st) {
    switch (nct) {
        case 1: gnn::k_rowdense<KQ, 1><<<grid, 64 * gnn::TB_WAVES, gnn::rowdense_lds<KQ, 1>(), st>>>(r); break;
        case 2: gnn::k_rowdense<KQ, 2><<<grid, 64 * gnn::TB_WAVES, gnn::rowdense_lds<KQ, 2>(), st>>>(r); break;
        case 3: gnn::k_rowdense<KQ, 3><<<grid, 64 * gnn::TB_WAVES, gnn::rowdense_lds<KQ, 3>(), st>>>(r); break;
        default: gnn::k_rowdense<KQ, 4><<<grid, 64 * gnn::TB_WAVES, gnn::rowdense_lds<KQ, 4>(), st>>>(r); break;
    }
    LAUNCH_OK();
    return 0;
}

int launch_rowdense(const gnn::SegDenseArgs &a, hipStream_t st) {
    gnn::RowDenseArgs r;
    memset(&r, 0, sizeof(r));
    const gnn::Seg &s = a.seg[0];
    r.gate = a.gate; r.M = a.M; r.X = s.ptr; r.ldx = s.ld; r.K = s.width; r.W = a.W; r.ldw = a.ldw; r.bias = a.bias; r.H = a.H; r.act = a.act;
    r.Y = a.Y; r.ldy = a.ldy;
    r.pred_old = a.pred_old; r.ld_pred = a.ld_pred; r.thr = a.pred_thr; r.pred_flag = a.pred_flag; r.pred_k = a.pred_k; r.pred_kval = a.pred_kval;
    const int kq = (s.width + 15) / 16, nct = (a.H + 15) / 16;
    const int grid = std::max(1, std::min(2 * device_cus(), cdiv(cdiv(a.M, 16), gnn::TB_WAVES)));
    switch (kq) {
        case 1: return launch_rowdense_k<1>(r, nct, grid, st);
        case 2: return launch_rowdense_k<2>(r, nct, grid, st);
        case 3: return launch_rowdense_k<3>(r, nct, grid, st);
        default: return launch_rowdense_k<4>(r, nct, grid, st);
    }
}

// Wide layers over one or two contiguous matrices + a per-row addend at large M (the un-fused first layer of state widths above 128,
// wide hidden layers): k_rowdense_wide, output columns in passes of 64.
bool rowdense_wide_applies(const gnn::SegDenseArgs &a) {
    static int off = -1;
    if (off < 0) { const char *e = getenv("GNN_ROWDENSE"); off = (e && e[0] == '0') ? 1 : 0; }
    if (off || a.M < 32768 || a.nseg < 1 || a.nseg > 2 || a.out_rowidx || a.add_rowidx || a.in_gamma || a.in_center || a.pred_flag || a.act == GNN_ACT_SOFTMAX) return false;
    if (a.H <= 64 || a.H % 4 || a.ldy % 4 || (a.addend && a.ld_add % 4)) return false;
    int chunks = 0;
    uintptr_t bits = reinterpret_cast<uintptr_t>(a.Y) | reinterpret_cast<uintptr_t>(a.addend);
    size_t span = std::max((size_t)a.ldy, (size_t)a.ld_add);
    for (int s = 0; s < a.nseg; ++s) {
        const gnn::Seg &g = a.seg[s];
        if (g.rowidx || g.width < 4 || g.width % 4 || g.ld % 4) return false;
        chunks += (g.width + 15) / 16;
        bits |= reinterpret_cast<uintptr_t>(g.ptr);
        span = std::max(span, (size_t)g.ld);
    }
    if (chunks > 32 || (bits & 15) != 0 || (size_t)a.M * span * 4 >= ((size_t)1 << 32)) return false;
    return true;
}

int launch_rowdense_wide(const gnn::SegDenseArgs &a, hipStream_t st) {
    gnn::RowDenseWideArgs r;
    memset(&r, 0, sizeof(r));
    r.gate = a.gate; r.M = a.M; r.nseg = a.nseg;
    int chunks = 0;
    for (int s = 0; s < a.nseg; ++s) { r.X[s] = a.seg[s].ptr; r.ldx[s] = a.seg[s].ld; r.width[s] = a.seg[s].width; r.wrow[s] = a.seg[s].wrow; chunks += (a.seg[s].width + 15) / 16; }
    r.W = a.W; r.ldw = a.ldw; r.bias = a.bias; r.addend = a.addend; r.ld_add = a.ld_add;
    r.H = a.H; r.act = a.act; r.Y = a.Y; r.ldy = a.ldy;
    const int passes = cdiv(a.H, 64);
    const size_t lds = (size_t)(16 * chunks * 64 + 64) * sizeof(float);
    dim3 grid(std::max(1, std::min(cdiv(device_cus(), passes), cdiv(cdiv(a.M, 32), gnn::RDW_WAVES))), passes);
    {
        static bool once = false;
        if (!once) { HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void *>(&gnn::k_rowdense_wide<8>), hipFuncAttributeMaxDynamicSharedMemorySize, 16 * 32 * 64 * 4 + 256)); once = true; }
        gnn::k_rowdense_wide<8><<<grid, 64 * gnn::RDW_WAVES, lds, st>>>(r);
    }
    LAUNCH_OK();
    return 0;
}

int launch_segdense(gnn::SegDenseArgs &a, hipStream_t st) {
    if (a.M == 0) return 0;
    if (a.H <= 4) {
        int K = 0;
        for (int s = 0; s < a.nseg; ++s) K = std::max(K, a.seg[s].wrow + a.seg[s].width);
        if ((size_t)K * a.H * sizeof(float) <= 48 * 1024) {
            const int grid = (int)std::max<long>(1, std::min<long>(cdiv(a.M, 16 * gnn::TD_ROWS), 256 * 8));
            const size_t lds = ((size_t)((K * a.H + 3) & ~3) + K + 4) * sizeof(float);
            switch (a.H) {
                case 1: gnn::k_thin_dense<1><<<grid, 256, lds, st>>>(a, K); break;
                case 2: gnn::k_thin_dense<2><<<grid, 256, lds, st>>>(a, K); break;
                case 3: gnn::k_thin_dense<3><<<grid, 256, lds, st>>>(a, K); break;
                default: gnn::k_thin_dense<4><<<grid, 256, lds, st>>>(a, K); break;
            }
            LAUNCH_OK();
            return 0;
        }
    }
    if (rowdense_applies(a)) return launch_rowdense(a, st);
    if (rowdense_wide_applies(a)) return launch_rowdense_wide(a, st);
    if (a.M <= 16384) gnn::k_segdense<4><<<cdiv(a.M, gnn::SD_TM), 256, 0, st>>>(a);      // latency regime
    else              gnn::k_segdense<1><<<cdiv(a.M, gnn::SD_TM), 256, 0, st>>>(a);      // throughput regime
    LAUNCH_OK();
    return 0;
}

int launch_softmax(const int *gate, float *Y, int M, int H, int ldy, const int *rowidx, hipStream_t st) {
    if (M == 0) return 0;
    gnn::k_softmax_rows<<<cdiv(M, 256), 256, 0, st>>>(gate, Y, M, H, ldy, rowidx);
    LAUNCH_OK();
    return 0;
}

int launch_converge(const int *gate, const float *s, const float *so, int N, int S, int ld_s, int ld_so, float thr,
                    int *flag_out, float *k_out, float k_val, hipStream_t st) {
    // The predicate is an OR over nodes: a probe over the first rows usually settles it ("some node still moves"), and
    // the pass over the rest returns at once when it has.  Only a converged state pays for the full read.
    const int probe = N > 65536 ? 4096 : 0;
    if (probe) {
        gnn::k_converge<<<cdiv(probe, 16), 256, 0, st>>>(gate, s, so, probe, S, ld_s, ld_so, thr, flag_out, k_out, k_val, nullptr);
        LAUNCH_OK();
    }
    const int rest = N - probe;
    const int grid = std::max(1, std::min(cdiv(rest, 16), 256 * 8));
    gnn::k_converge<<<grid, 256, 0, st>>>(gate, s + (size_t)probe * ld_s, so ? so + (size_t)probe * ld_so : nullptr, rest, S, ld_s, ld_so,
                                          thr, flag_out, probe ? nullptr : k_out, k_val, probe ? flag_out : nullptr);
    LAUNCH_OK();
    return 0;
}

// BN folding of the first layers of up to GNN_MAX_TYPES + 1 networks in one launch (+ optional zeroing of two small arrays)
struct FoldList {
    gnn::FoldArgs fa;
    int blocks = 0;
    FoldList() { memset(&fa, 0, sizeof(fa)); }
    void add(const gnn_mlp_t &m, float *Wf, float *bf) {
        gnn::FoldJob &j = fa.job[fa.n_jobs++];
        j.W = m.kernel[0]; j.b = m.bias[0]; j.K = m.in_dim; j.H = m.units[0];
        j.gamma = m.has_bn ? m.bn_gamma : nullptr; j.beta = m.bn_beta; j.mean = m.bn_mean; j.var = m.bn_var; j.eps = m.bn_eps;
        j.Wf = Wf; j.bf = bf; j.blk_begin = blocks;
        blocks += j.H;
    }
};

int launch_fold_list(FoldList &fl, hipStream_t st) {
    if (fl.blocks == 0) return 0;
    gnn::k_fold_bn<<<fl.blocks, 128, 0, st>>>(fl.fa);
    LAUNCH_OK();
    return 0;
}

int launch_fold(const gnn_mlp_t &m, float *Wf, float *bf, hipStream_t st) {
    FoldList fl;
    fl.add(m, Wf, bf);
    return launch_fold_list(fl, st);
}

int launch_copy2d(const int *gate, const float *src, int ld_src, float *dst, int ld_dst, int rows, int width, int fill_to, hipStream_t st) {
    if (rows == 0 || fill_to == 0) return 0;
    const long total = (long)rows * fill_to;
    gnn::k_copy2d<<<std::min(cdiv(total, 256), 256 * 16), 256, 0, st>>>(gate, src, ld_src, dst, ld_dst, rows, width, fill_to);
    LAUNCH_OK();
    return 0;
}

// Run a reference MLP whose first layer has already been BN-folded into (Wf, bf).  The first layer reads the virtual
// concatenation described by `segs`; `addend` (optional) is a per-row constant already containing the bias.
struct MlpRun {
    const gnn_mlp_t *mlp;
    const float *Wf, *bf;
    gnn::Seg segs[GNN_MAX_SEGS];
    int nseg = 0;
    const float *addend = nullptr; int ld_add = 0; const int *add_rowidx = nullptr;
    int M = 0;
    float *hid[2] = {nullptr, nullptr}; int ld_hid = 0;
    float *Y = nullptr; int ldy = 0; const int *out_rowidx = nullptr;
    const int *gate = nullptr;
    // optional: the convergence predicate in the last layer's epilogue (homogeneous models, 4 < width <= 64, no softmax)
    const float *pred_old = nullptr; int ld_pred = 0; float pred_thr = 0.f; int *pred_flag = nullptr; float *pred_k = nullptr; float pred_kval = 0.f;
    mutable bool pred_fused = false;
};

int run_mlp(const MlpRun &r, hipStream_t st) {
    const gnn_mlp_t &m = *r.mlp;
    const float *cur = nullptr;
    int cur_ld = 0;
    for (int l = 0; l < m.n_layers; ++l) {
        const bool last = (l == m.n_layers - 1);
        gnn::SegDenseArgs a;
        memset(&a, 0, sizeof(a));
        a.gate = r.gate;
        a.M = r.M;
        a.H = m.units[l];
        if (l == 0) {
            a.nseg = r.nseg;
            for (int s = 0; s < r.nseg; ++s) a.seg[s] = r.segs[s];
            a.W = r.Wf;
            a.bias = r.addend ? nullptr : r.bf;
            a.addend = r.addend; a.ld_add = r.ld_add; a.add_rowidx = r.add_rowidx;
        } else {
            a.nseg = 1;
            a.seg[0] = gnn::Seg{cur, nullptr, cur_ld, m.units[l - 1], 0};
            a.W = m.kernel[l];
            a.bias = m.bias[l];
        }
        a.ldw = a.H;
        // softmax rows are finished inside the thin-output kernel (<= 4 classes); wider layers get a second pass
        const bool thin_softmax = m.activation[l] == GNN_ACT_SOFTMAX && thin_dense_applies(a);
        a.act = (m.activation[l] == GNN_ACT_SOFTMAX && !thin_softmax) ? GNN_ACT_LINEAR : m.activation[l];
        if (last) { a.Y = r.Y; a.ldy = r.ldy; a.out_rowidx = r.out_rowidx; }
        else      { a.Y = r.hid[l & 1]; a.ldy = r.ld_hid; a.out_rowidx = nullptr; }
        if (last && r.pred_flag && !r.out_rowidx && a.H <= 64 && a.H > 4 && m.activation[l] != GNN_ACT_SOFTMAX) {
            a.pred_old = r.pred_old; a.ld_pred = r.ld_pred; a.pred_thr = r.pred_thr; a.pred_flag = r.pred_flag;
            a.pred_k = r.pred_k; a.pred_kval = r.pred_kval;
            r.pred_fused = true;
        }
        TRY(launch_segdense(a, st));
        if (m.activation[l] == GNN_ACT_SOFTMAX && !thin_softmax) TRY(launch_softmax(r.gate, a.Y, a.M, a.H, a.ldy, a.out_rowidx, st));
        cur = a.Y; cur_ld = a.ldy;
    }
    return 0;
}

// --- the loop plan ---------------------------------------------------------------------------------------------------
struct TypePlan {
    const gnn_mlp_t *net;
    const int *rows;       // node ids of this type (nullptr = all nodes, homogeneous)
    int count;
    float *Wf, *bf;        // folded first layer [in_dim x H1], [H1]
    float *Wc;             // XC form: [32 x H1] (see Plan::Xc)
    int wrow_state, wrow_agg;
    gnn::Seg cseg[GNN_MAX_SEGS];   // iteration-invariant segments (rowidx filled with `rows`)
    int ncseg;
};

struct Plan {
    int N, E, S, SP, L, A, T;        // S = state width, SP = padded leading dimension of internal state buffers
    int H1max, Hmax_state, Hmax_out, Tout, M, G;
    bool composite, fused;
    int n_heavy;                     // virtual state rows appended after the N real ones (hub segments)
    TypePlan tp[GNN_MAX_TYPES];
    // workspace
    int *flags;
    int *mid_bar;                    // barrier lines of the mid-size whole-loop kernel (inside the zeroed loop words)
    gnn::GroupTab gt;                // convergence groups (one group = the whole graph when the caller gave none)
    int n_groups;                    // the caller's n_groups (0: k_out is one float)
    int group_tiles;                 // 64-node tiles when no tile straddles a group
    int group_max_nodes;             // nodes of the largest group
    int *d_group_tabs;               // device copies of group_node_begin / 64-node-tile offsets / set first group / set size, [n_groups + 1] each
    unsigned long long *set_bar;     // two flag-exchange counters per group set (indexed by the set's first group), zeroed per call
    bool has_sets;                   // some set has more than one group
    int *pred0;                      // state_0's predicate, one word per 64-node tile (written by k_setup_small, read by k_state_small)
    int *err;                        // sticky "an in-launch wait expired" word of the fused kernels, folded into k at the end
    float *agg_arcs, *agg_nodes; int ld_agg_nodes;
    float *C; int ldC;
    float *Xc;                       // constant inputs [N, 32] for the XC kernel variant (xc_ok); their weights: TypePlan::Wc
    float *Wx;                       // state widths 129 .. 256: first-layer weights in MFMA fragment order (kernel_state_xwide.hpp)
    bool xc_ok;
    float *buf[2], *agg;
    float *hid[2]; int ld_hid;
    float *Wf_out, *bf_out;
    float *ohid[2]; int ld_ohid;
    float *out_nodes;
    int *idx_src, *idx_dst;
    size_t bytes;
};

constexpr int GNN_SMALL_MAX_TILES = 512;   // upper bound of CUs a whole-loop launch can cover (one 64-node tile each)
// words behind flags[max_iteration]: [1] last flag, [3..7) barrier counters of the small whole-loop kernel, [12] error word,
// then (128-byte aligned) the mid-size whole-loop kernel's barrier lines; all zeroed by the set-up launch
constexpr int GNN_LOOP_WORDS = 16 + 32 + gnn::MID_BAR_WORDS;

int fused_generation(int SP, int n_nodes, int flags);

// GNN_XC=0 keeps the per-node constant C in the wave-specialised kernel at every size, GNN_XC_MIN_NODES moves the size from
// which the constant inputs are multiplied instead (tuning knobs; both forms are held to the same tests).  Measured, d = 64,
// 10 arcs per node, us per iteration with inputs / with C: 100 k nodes 61.1 / 59.1, 200 k 105.0 / 105.8, 400 k 192.2 / 199.7,
// 600 k 281.6 / 294.0, 1 M (C4) 458 / 480, 4 M 2 215 / 2 329; d = 32 at C4 size 263.2 / 263.7.
bool xc_disabled() {
    int v = -1;      // (read at every call: tests switch it inside one process)
    { const char *e = getenv("GNN_XC"); v = (e && atoi(e) == 0) ? 1 : 0; }
    return v == 1;
}
int xc_min_nodes() {
    int v = -1;      // (read at every call: tests switch it inside one process)
    { const char *e = getenv("GNN_XC_MIN_NODES"); v = e ? atoi(e) : 196608; }
    return v;
}

int state_width(const gnn_loop_args_t &a) { return a.state_dim > 0 ? a.state_dim : a.dim_node_label; }

int make_plan(const gnn_loop_args_t &a, void *ws, Plan &p, bool validate_ptrs) {
    memset(&p, 0, sizeof(p));
    if (a.abi_version != GNN_ABI_VERSION) return fail("abi_version %d != %d", a.abi_version, GNN_ABI_VERSION);
    if (a.n_nodes < 0 || a.n_arcs < 0) return fail("negative graph size");
    if (a.state_dim < 0 || a.max_iteration < 0 || !(a.state_threshold >= 0.0f)) return fail("state_dim, max_iteration, state_threshold must be >= 0");
    p.composite = a.composite != 0;
    p.T = p.composite ? a.n_types : 1;
    if (p.T < 1 || p.T > GNN_MAX_TYPES) return fail("n_types %d out of [1,%d]", p.T, GNN_MAX_TYPES);
    if (p.composite && a.max_iteration < 1) return fail("composite GNN requires max_iteration > 0");
    p.N = a.n_nodes; p.E = a.n_arcs; p.L = a.dim_node_label; p.A = a.dim_arc_label;
    p.S = state_width(a);
    if (p.S < 1) return fail("state width is 0 (state_dim == 0 and dim_node_label == 0)");
    p.SP = state_ld(p.S);
    if (a.focus < GNN_FOCUS_NODE || a.focus > GNN_FOCUS_GRAPH) return fail("unknown focus %d", a.focus);
    p.M = a.n_out;
    if (p.M < 0) return fail("n_out < 0");
    static_assert(gnn::MAX_GROUPS == GNN_MAX_GROUPS, "group table of the kernels and the header differ");
    p.n_groups = a.n_groups;
    if (a.n_groups < 0 || a.n_groups > GNN_MAX_GROUPS_RESIDENT) return fail("n_groups %d out of [0,%d]", a.n_groups, GNN_MAX_GROUPS_RESIDENT);
    if (a.n_groups > 0) {
        if (p.composite) return fail("convergence groups are not supported for composite graphs");
        if (!a.group_node_begin) return fail("group_node_begin is NULL");
        if (a.group_node_begin[0] != 0 || a.group_node_begin[a.n_groups] != a.n_nodes) return fail("group_node_begin must span [0, n_nodes]");
        p.gt.n = a.n_groups <= GNN_MAX_GROUPS ? a.n_groups : 0;
        for (int g = 0; g < a.n_groups; ++g) {
            const int nb = a.group_node_begin[g], ne = a.group_node_begin[g + 1];
            if (ne <= nb) return fail("group %d is empty or group_node_begin is not ascending", g);
            if (g < GNN_MAX_GROUPS && p.gt.n) { p.gt.node_begin[g] = nb; p.gt.tile_begin[g] = p.group_tiles; }
            p.group_tiles += (ne - nb + 63) / 64;
            p.group_max_nodes = std::max(p.group_max_nodes, ne - nb);
        }
        if (p.gt.n) { p.gt.node_begin[a.n_groups] = a.n_nodes; p.gt.tile_begin[a.n_groups] = p.group_tiles; }
        if (a.n_group_sets < 0 || a.n_group_sets > a.n_groups) return fail("n_group_sets %d out of [0, n_groups]", a.n_group_sets);
        if (a.n_group_sets > 0) {
            if (!a.group_set_begin) return fail("group_set_begin is NULL");
            if (a.group_set_begin[0] != 0 || a.group_set_begin[a.n_group_sets] != a.n_groups) return fail("group_set_begin must span [0, n_groups]");
            for (int s2 = 0; s2 < a.n_group_sets; ++s2) {
                const int n_in = a.group_set_begin[s2 + 1] - a.group_set_begin[s2];
                if (n_in < 1) return fail("group set %d is empty or group_set_begin is not ascending", s2);
                if (n_in > 1) p.has_sets = true;
            }
        }
    } else {
        p.gt.n = 1; p.gt.node_begin[0] = 0; p.gt.node_begin[1] = a.n_nodes;
        p.group_tiles = (a.n_nodes + 63) / 64;
        p.gt.tile_begin[0] = 0; p.gt.tile_begin[1] = p.group_tiles;
    }
    int sum_dt = 0;
    for (int t = 0; t < p.T; ++t) {
        TRY(check_mlp(a.net_state[t], "net_state", validate_ptrs));
        const gnn_mlp_t &ns = a.net_state[t];
        int expect;
        if (p.composite) {
            if (a.type_dim_label[t] < 0 || a.type_dim_label[t] > p.L) return fail("type_dim_label[%d]=%d out of [0,%d]", t, a.type_dim_label[t], p.L);
            sum_dt += a.type_dim_label[t];
        }
        (void)expect;
        if (ns.units[ns.n_layers - 1] != p.S) return fail("net_state[%d] output width %d != state width %d", t, ns.units[ns.n_layers - 1], p.S);
        p.H1max = std::max(p.H1max, (int)ns.units[0]);
        p.Hmax_state = std::max(p.Hmax_state, max_units(ns));
    }
    for (int t = 0; t < p.T; ++t) {
        const gnn_mlp_t &ns = a.net_state[t];
        const int expect = p.composite ? a.type_dim_label[t] + 2 * p.S + sum_dt + p.A
                                       : (a.state_dim > 0 ? 2 * p.S + 2 * p.L + p.A : 2 * p.S + p.A);
        if (ns.in_dim != expect) return fail("net_state[%d].in_dim %d != %d expected from the graph dims", t, ns.in_dim, expect);
    }
    TRY(check_mlp(a.net_output, "net_output", validate_ptrs));
    {
        const int node_part = p.composite ? p.S : (a.state_dim > 0 ? p.S + p.L : p.S);
        const int expect = a.focus == GNN_FOCUS_ARC ? 2 * node_part + p.A : node_part;
        if (a.net_output.in_dim != expect) return fail("net_output.in_dim %d != %d expected for this focus", a.net_output.in_dim, expect);
    }
    p.Hmax_out = max_units(a.net_output);
    p.Tout = a.net_output.units[a.net_output.n_layers - 1];
    p.G = a.focus == GNN_FOCUS_GRAPH ? a.nodegraph.n_dst : 0;

    if (validate_ptrs) {
        TRY(check_csr(a.adjacency, "adjacency", p.N, a.nodes_src ? a.adjacency.n_src : p.N));
        TRY(check_csr(a.arcnode, "arcnode", p.N, p.E));
        if (p.N > 0 && !a.nodes) return fail("nodes is NULL");
        if (p.E > 0 && p.A > 0 && !a.arc_labels) return fail("arc_labels is NULL");
        if (a.state_dim > 0 && p.N > 0 && !a.state0) return fail("state0 is required when state_dim > 0");
        if (!a.k_out) return fail("k_out is NULL");
        if (p.N > 0 && !a.state_out) return fail("state_out is NULL");
        if (!a.out && (a.focus == GNN_FOCUS_GRAPH ? a.nodegraph.n_dst : p.M) > 0) return fail("out is NULL");
        if (p.M > 0 && !a.out_index) return fail("out_index is NULL");
        if (a.focus == GNN_FOCUS_ARC && p.E > 0 && (!a.arc_src || !a.arc_dst)) return fail("arc focus needs arc_src / arc_dst");
        if (a.focus == GNN_FOCUS_GRAPH) {
            if (a.nodegraph.n_src != p.M) return fail("graph focus: NodeGraph has %d rows but %d nodes pass the mask (the reference's matmul would fail too)", a.nodegraph.n_src, p.M);
            TRY(check_csr(a.nodegraph, "nodegraph", a.nodegraph.n_dst, p.M));
        }
        if (a.n_heavy_segments > 0) {
            // (on a shard the virtual rows sit behind the adjacency.n_src rows of the exchanged full buffer, which the caller
            // allocates n_heavy_segments rows longer; on one GPU n_src == n_nodes)
            if (!a.heavy_seg_beg || !a.heavy_seg_end) return fail("heavy_seg_beg / heavy_seg_end is NULL");
            TRY(check_csr(a.adjacency_light, "adjacency_light", p.N, (a.nodes_src ? a.adjacency.n_src : p.N) + a.n_heavy_segments));
        }
        if (p.composite) {
            if (!a.type_nodes && p.N > 0) return fail("type_nodes is NULL");
            if (a.type_offsets[0] != 0 || a.type_offsets[p.T] != p.N) return fail("type_offsets must span [0, n_nodes]");
            for (int t = 0; t < p.T; ++t)
                TRY(check_csr(a.composite_adjacency[t], "composite_adjacency", p.N, a.nodes_src ? a.adjacency.n_src : p.N));
        }
    }

    // ---- carve ----
    Carver c(ws);
    p.flags = c.take<int>(a.max_iteration + GNN_LOOP_WORDS);   // behind the flags: the persistent kernel's two 64-bit barrier counters, the error word
    p.err = p.flags ? p.flags + a.max_iteration + 12 : nullptr;
    p.mid_bar = p.flags ? p.flags + ((a.max_iteration + 16 + 31) & ~31) : nullptr;
    p.pred0 = c.take<int>(std::max(GNN_SMALL_MAX_TILES, p.group_tiles));
    p.d_group_tabs = c.take<int>(a.n_groups > 0 ? 4 * ((size_t)a.n_groups + 1) : 0);
    p.set_bar = c.take<unsigned long long>(a.n_groups > 0 ? 2 * (size_t)a.n_groups : 0);
    for (int t = 0; t < p.T; ++t) {
        p.tp[t].Wf = c.take<float>((size_t)a.net_state[t].in_dim * a.net_state[t].units[0]);
        p.tp[t].bf = c.take<float>(a.net_state[t].units[0]);
    }
    p.Wf_out = c.take<float>((size_t)a.net_output.in_dim * a.net_output.units[0]);
    p.bf_out = c.take<float>(a.net_output.units[0]);
    p.agg_arcs = c.take<float>((size_t)p.N * std::max(p.A, 1));
    p.ld_agg_nodes = p.composite ? sum_dt : (a.state_dim > 0 ? p.L : 0);
    p.agg_nodes = c.take<float>((size_t)p.N * std::max(p.ld_agg_nodes, 1));
    p.ldC = (p.H1max + 3) & ~3;      // rows start 16-B aligned: the wave-specialised fused kernel reads C as float4
    p.C = c.take<float>((size_t)p.N * p.ldC);
    // at most 31 constant input columns per node type (one more carries the bias): the wave-specialised kernel may read the
    // inputs (128 B per node) and multiply them instead of reading C (4 H1 bytes per node)
    p.xc_ok = p.SP <= 64 && !(a.flags & GNN_FLAG_UNFUSED) && !xc_disabled() && p.N >= xc_min_nodes();
    for (int t = 0; t < p.T; ++t) {
        const int kc = p.composite ? a.type_dim_label[t] + sum_dt + p.A : (a.state_dim > 0 ? 2 * p.L : 0) + p.A;
        p.xc_ok = p.xc_ok && kc <= 31 && a.net_state[t].n_layers == 1;
    }
    p.Xc = c.take<float>(p.xc_ok ? (size_t)p.N * 32 : 0);
    for (int t = 0; t < p.T; ++t) p.tp[t].Wc = c.take<float>(p.xc_ok ? (size_t)32 * a.net_state[t].units[0] : 0);
    // hub segments: virtual rows for padded widths up to 128; wider states walk the plain adjacency (the un-fused aggregate
    // handles any degree), so the caller may always pass the split
    p.n_heavy = (a.n_heavy_segments > 0 && p.SP <= 128) ? a.n_heavy_segments : 0;
    p.buf[0] = c.take<float>((size_t)(p.N + p.n_heavy) * p.SP + 64);
    p.buf[1] = c.take<float>((size_t)(p.N + p.n_heavy) * p.SP + 64);
    p.agg = c.take<float>((size_t)p.N * p.SP);
    p.Wx = c.take<float>((p.SP > 128 && p.S <= 256) ? gnn::xwide_weight_floats(p.S, p.SP) : 0);
    p.ld_hid = p.Hmax_state;
    p.hid[0] = c.take<float>((size_t)p.N * p.ld_hid);
    p.hid[1] = c.take<float>((size_t)p.N * p.ld_hid);
    p.ld_ohid = p.Hmax_out;
    p.ohid[0] = c.take<float>((size_t)p.M * p.ld_ohid);
    p.ohid[1] = c.take<float>((size_t)p.M * p.ld_ohid);
    p.out_nodes = c.take<float>((size_t)p.M * p.Tout);
    p.idx_src = c.take<int>(std::max(p.M, 1));
    p.idx_dst = c.take<int>(std::max(p.M, 1));
    p.bytes = (c.off + 255) & ~(size_t)255;

    // ---- per-type first-layer layout ----
    for (int t = 0; t < p.T; ++t) {
        TypePlan &tp = p.tp[t];
        tp.net = &a.net_state[t];
        tp.ncseg = 0;
        if (!p.composite) {
            tp.rows = nullptr; tp.count = p.N;
            if (a.state_dim > 0) {   // [state | nodes | agg_state | agg_nodes | agg_arcs]   (GNN.py:222-231)
                tp.wrow_state = 0; tp.wrow_agg = p.S + p.L;
                tp.cseg[tp.ncseg++] = gnn::Seg{a.nodes, nullptr, a.ld_nodes, p.L, p.S};
                tp.cseg[tp.ncseg++] = gnn::Seg{p.agg_nodes, nullptr, p.ld_agg_nodes, p.L, 2 * p.S + p.L};
                if (p.A > 0) tp.cseg[tp.ncseg++] = gnn::Seg{p.agg_arcs, nullptr, p.A, p.A, 2 * p.S + 2 * p.L};
            } else {                 // [state | agg_state | agg_arcs]
                tp.wrow_state = 0; tp.wrow_agg = p.S;
                if (p.A > 0) tp.cseg[tp.ncseg++] = gnn::Seg{p.agg_arcs, nullptr, p.A, p.A, 2 * p.S};
            }
        } else {                     // [nodes[:, :d_t] | state | agg_state | agg_nodes_0.. | agg_arcs]  (CompositeGNN.py:224)
            const int dt = a.type_dim_label[t];
            tp.rows = a.type_nodes ? a.type_nodes + a.type_offsets[t] : nullptr;
            tp.count = a.type_offsets[t + 1] - a.type_offsets[t];
            tp.wrow_state = dt; tp.wrow_agg = dt + p.S;
            if (dt > 0) tp.cseg[tp.ncseg++] = gnn::Seg{a.nodes, tp.rows, a.ld_nodes, dt, 0};
            if (sum_dt > 0) tp.cseg[tp.ncseg++] = gnn::Seg{p.agg_nodes, tp.rows, p.ld_agg_nodes, sum_dt, dt + 2 * p.S};
            if (p.A > 0) tp.cseg[tp.ncseg++] = gnn::Seg{p.agg_arcs, tp.rows, p.A, p.A, dt + 2 * p.S + sum_dt};
        }
    }
    return 0;
}

// the operator the iterations walk: the light one when hub rows were split off
inline const gnn_csr_t &iter_adjacency(const gnn_loop_args_t &a, const Plan &p) { return p.n_heavy > 0 ? a.adjacency_light : a.adjacency; }

// hub pre-pass: virtual rows N .. N + n_heavy of the buffer the iteration is about to read
int launch_heavy(const gnn_loop_args_t &a, const Plan &p, const int *gate, const float *src, hipStream_t st) {
    if (p.n_heavy == 0) return 0;
    float *buf = const_cast<float *>(src);
    const int first_virtual = a.nodes_src ? a.adjacency.n_src : p.N;     // a shard's full buffer: behind every visible row
    const int grid = std::min(p.n_heavy, 256 * 8);
    switch (p.SP) {
        case 16: gnn::k_heavy_segments<16><<<grid, 256, 0, st>>>(gate, a.heavy_seg_beg, a.heavy_seg_end, p.n_heavy, a.adjacency.src, a.adjacency.w, buf, first_virtual); break;
        case 32: gnn::k_heavy_segments<32><<<grid, 256, 0, st>>>(gate, a.heavy_seg_beg, a.heavy_seg_end, p.n_heavy, a.adjacency.src, a.adjacency.w, buf, first_virtual); break;
        case 64: gnn::k_heavy_segments<64><<<grid, 256, 0, st>>>(gate, a.heavy_seg_beg, a.heavy_seg_end, p.n_heavy, a.adjacency.src, a.adjacency.w, buf, first_virtual); break;
        default: gnn::k_heavy_segments<128><<<grid, 256, 0, st>>>(gate, a.heavy_seg_beg, a.heavy_seg_end, p.n_heavy, a.adjacency.src, a.adjacency.w, buf, first_virtual); break;
    }
    LAUNCH_OK();
    return 0;
}

// one un-fused iteration: agg = A^T state ; state_new = net_state([state | agg] + C) per type ; predicate.
int iteration_unfused(const gnn_loop_args_t &a, const Plan &p, const int *gate, const float *src_full, float *dst_full,
                      int row_base, int *flag_next, float *k_out, float k_val, hipStream_t st) {
    GNN_SET_KERNEL_NAME("k_aggregate_vec + k_segdense + k_converge (un-fused)");
    TRY(launch_heavy(a, p, gate, src_full, st));
    // the padded width: pad columns are zero on both sides, and whole 16-B chunks let the vector kernel run for any d
    TRY(launch_aggregate(gate, iter_adjacency(a, p), src_full, p.SP, p.SP, p.agg, p.SP, st));
    const float *src = src_full + (size_t)row_base * p.SP;     // own rows
    float *dst = dst_full + (size_t)row_base * p.SP;
    bool pred_fused = false;
    for (int t = 0; t < p.T; ++t) {
        const TypePlan &tp = p.tp[t];
        if (tp.count == 0) continue;
        MlpRun r;
        r.mlp = tp.net; r.Wf = tp.Wf; r.bf = tp.bf;
        r.nseg = 2;
        r.segs[0] = gnn::Seg{src, tp.rows, p.SP, p.S, tp.wrow_state};
        r.segs[1] = gnn::Seg{p.agg, tp.rows, p.SP, p.S, tp.wrow_agg};
        r.addend = p.C; r.ld_add = p.ldC; r.add_rowidx = tp.rows;
        r.M = tp.count;
        r.hid[0] = p.hid[0]; r.hid[1] = p.hid[1]; r.ld_hid = p.ld_hid;
        r.Y = dst; r.ldy = p.SP; r.out_rowidx = tp.rows;
        r.gate = gate;
        if (flag_next && p.T == 1 && !tp.rows) {        // one network over all rows: the predicate rides in its last layer's epilogue
            r.pred_old = src; r.ld_pred = p.SP; r.pred_thr = a.state_threshold; r.pred_flag = flag_next; r.pred_k = k_out; r.pred_kval = k_val;
        }
        TRY(run_mlp(r, st));
        pred_fused |= r.pred_fused;
    }
    if (flag_next && !pred_fused) TRY(launch_converge(gate, dst, src, p.N, p.S, p.SP, p.SP, a.state_threshold, flag_next, k_out, k_val, st));
    return 0;
}

// `skip_c`: the caller runs every iteration on the XC form of the wave-specialised kernel, which never reads C
// iteration constants handed in by the caller instead of being aggregated here: what the reference's `convergence` receives as
// `aggregated_nodes` / `aggregated_arcs` (GNN.py:217) or, column blocks of one matrix, `aggregated_component` (CompositeGNN.py:214)
struct GivenAgg { const float *nodes; int ld_nodes; const float *arcs; int ld_arcs; };

int setup_constants(const gnn_loop_args_t &a, const Plan &p, hipStream_t st, bool zero_loop_words = false, bool skip_c = false,
                    const GivenAgg *given = nullptr) {
    // BN folding of every first layer: one launch, which also zeroes the flag words / barrier counters and k
    FoldList fl;
    for (int t = 0; t < p.T; ++t) fl.add(a.net_state[t], p.tp[t].Wf, p.tp[t].bf);
    if (a.net_output.kernel[0]) fl.add(a.net_output, p.Wf_out, p.bf_out);      // absent for the standalone state step
    if (zero_loop_words) { fl.fa.zero_a = p.flags; fl.fa.n_a = a.max_iteration + GNN_LOOP_WORDS; fl.fa.zero_b = a.k_out; fl.fa.n_b = 1; }
    TRY(launch_fold_list(fl, st));
    // ArcNode scatter-add (GNN.py:254) and neighbour-label aggregates (GNN.py:258 / CompositeGNN.py:251)
    if (given) {
        if (p.A > 0) TRY(launch_copy2d(nullptr, given->arcs, given->ld_arcs, p.agg_arcs, p.A, p.N, p.A, p.A, st));
        if (p.ld_agg_nodes > 0) TRY(launch_copy2d(nullptr, given->nodes, given->ld_nodes, p.agg_nodes, p.ld_agg_nodes, p.N, p.ld_agg_nodes, p.ld_agg_nodes, st));
    } else {
    if (p.A > 0) TRY(launch_aggregate(nullptr, a.arcnode, a.arc_labels, a.ld_arcs, p.A, p.agg_arcs, p.A, st));
    if (!p.composite) {
        if (a.state_dim > 0) TRY(launch_aggregate(nullptr, a.adjacency, a.nodes_src ? a.nodes_src : a.nodes,
                                                  a.nodes_src ? a.ld_nodes_src : a.ld_nodes, p.L, p.agg_nodes, p.ld_agg_nodes, st));
    } else {
        int col = 0;
        for (int t = 0; t < p.T; ++t) {
            const int dt = a.type_dim_label[t];
            if (dt > 0) TRY(launch_aggregate(nullptr, a.composite_adjacency[t], a.nodes_src ? a.nodes_src : a.nodes,
                                             a.nodes_src ? a.ld_nodes_src : a.ld_nodes, dt, p.agg_nodes + col, p.ld_agg_nodes, st));
            col += dt;
        }
    }
    }
    // C[j] = const segments . Wf[const rows] + bf     (written row-scattered per type; every node has one type)
    for (int t = 0; t < p.T && !skip_c; ++t) {
        const TypePlan &tp = p.tp[t];
        if (tp.count == 0) continue;
        gnn::SegDenseArgs d;
        memset(&d, 0, sizeof(d));
        d.M = tp.count; d.H = tp.net->units[0];
        d.nseg = tp.ncseg;
        for (int s = 0; s < tp.ncseg; ++s) d.seg[s] = tp.cseg[s];
        d.W = tp.Wf; d.ldw = d.H; d.bias = tp.bf; d.act = GNN_ACT_LINEAR;
        d.Y = p.C; d.ldy = p.ldC; d.out_rowidx = tp.rows;
        TRY(launch_segdense(d, st));
    }
    if (p.xc_ok && p.N > 0 && (fused_generation(p.SP, p.N, a.flags) == 4 || a.nodes_src)) {   // only the wave-specialised kernel reads them (shards: the
                                                                                       // overlapped iteration always runs it)
        for (int t = 0; t < p.T; ++t) {
            const TypePlan &tp = p.tp[t];
            if (tp.count == 0) continue;
            gnn::PackSegs ps;
            memset(&ps, 0, sizeof(ps));
            ps.n = tp.ncseg;
            for (int s2 = 0; s2 < tp.ncseg && s2 < 3; ++s2) { ps.ptr[s2] = tp.cseg[s2].ptr; ps.ld[s2] = tp.cseg[s2].ld; ps.width[s2] = tp.cseg[s2].width; ps.wrow[s2] = tp.cseg[s2].wrow; }
            const int H = tp.net->units[0];
            gnn::k_pack_xc<<<(int)std::min<long>(cdiv((long)tp.count * 32, 256), 256 * 16), 256, 0, st>>>(tp.count, tp.rows, ps, p.Xc);
            LAUNCH_OK();
            gnn::k_pack_wc<<<cdiv(32 * H, 256), 256, 0, st>>>(tp.Wf, tp.bf, H, ps, tp.Wc);
            LAUNCH_OK();
        }
    }
    return 0;
}

// Small homogeneous graphs about to run the whole-loop kernel: BN folds, both constant aggregates, C, the predicate of
// state_0 and the zeroing of the loop words in ONE launch (kernels_setup.hpp) instead of five.
bool setup_small_applies(const gnn_loop_args_t &a, const Plan &p) {
    if (p.composite || p.N == 0) return false;
    const int L = a.state_dim > 0 ? p.L : 0, H = a.net_state[0].units[0];
    if (H > 128) return false;
    return gnn::setup_small_lds(H, 2 * L + p.A, std::max(a.net_state[0].in_dim, a.net_output.in_dim)) <= 64 * 1024;
}

int setup_small(const gnn_loop_args_t &a, const Plan &p, hipStream_t st) {
    gnn::SetupArgs sa;
    memset(&sa, 0, sizeof(sa));
    FoldList fl;
    fl.add(a.net_state[0], p.tp[0].Wf, p.tp[0].bf);
    sa.net = fl.fa.job[0];
    if (a.net_output.kernel[0]) { fl.add(a.net_output, p.Wf_out, p.bf_out); sa.out = fl.fa.job[1]; }
    const int L = a.state_dim > 0 ? p.L : 0;
    sa.N = p.N; sa.n_tiles = p.group_tiles; sa.groups = p.gt;
    if (p.n_groups > GNN_MAX_GROUPS || (p.n_groups > 0 && p.gt.n == 0)) {     // many groups: the tables uploaded by the caller of this function
        sa.groups.d_node_begin = p.d_group_tabs; sa.groups.d_tile_begin = p.d_group_tabs + (p.n_groups + 1); sa.groups.n_dev = p.n_groups;
    }
    sa.nodes = a.nodes; sa.ld_nodes = a.ld_nodes; sa.L = L;
    sa.nodes_src = a.nodes_src ? a.nodes_src : a.nodes; sa.ld_nodes_src = a.nodes_src ? a.ld_nodes_src : a.ld_nodes;
    sa.adj = gnn::SetupCsr{a.adjacency.rowptr, a.adjacency.src, a.adjacency.w, a.adjacency.row_scale};
    sa.arc_labels = a.arc_labels; sa.ld_arcs = a.ld_arcs; sa.A = p.A;
    sa.arcnode = gnn::SetupCsr{a.arcnode.rowptr, a.arcnode.src, a.arcnode.w, a.arcnode.row_scale};
    if (a.state_dim > 0) { sa.row_nodes = p.S; sa.row_aggn = 2 * p.S + p.L; sa.row_agga = 2 * p.S + 2 * p.L; }
    else                 { sa.row_agga = 2 * p.S; }
    sa.C = p.C; sa.ldC = p.ldC;
    sa.state0 = a.state_dim > 0 ? a.state0 : a.nodes; sa.ld_s0 = a.state_dim > 0 ? p.S : a.ld_nodes;
    sa.S = p.S; sa.thr = a.state_threshold; sa.pred0 = p.pred0;
    sa.zero_a = p.flags; sa.n_a = a.max_iteration + GNN_LOOP_WORDS; sa.zero_b = a.k_out; sa.n_b = std::max(1, p.n_groups);
    gnn::k_setup_small<<<sa.n_tiles + 1, 256, gnn::setup_small_lds(sa.net.H, 2 * L + p.A, std::max(a.net_state[0].in_dim, a.net_output.in_dim)), st>>>(sa);
    LAUNCH_OK();
    return 0;
}

// a bounded in-launch wait of a fused kernel expired somewhere in the loop: k < 0 tells the caller (same convention as the
// persistent whole-loop kernel, kernel_state_small.hpp)
__global__ void k_fold_error(const int *err, float *k_out) {
    if (*err != 0) *k_out = -1.0e9f;
}

int launch_fold_error(const Plan &p, float *k_out, hipStream_t st) {
    if (!k_out || !p.err) return 0;
    k_fold_error<<<1, 1, 0, st>>>(p.err, k_out);
    LAUNCH_OK();
    return 0;
}

__global__ void k_arc_endpoints(const int *__restrict__ out_index, const int *__restrict__ arc_src,
                                const int *__restrict__ arc_dst, int M, int *__restrict__ idx_src, int *__restrict__ idx_dst) {
    const int m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m < M) { const int e = out_index[m]; idx_src[m] = arc_src[e]; idx_dst[m] = arc_dst[e]; }
}

int output_stage(const gnn_loop_args_t &a, const Plan &p, hipStream_t st) {
    if (p.M == 0) return 0;
    const bool with_labels = !p.composite && a.state_dim > 0;
    MlpRun r;
    r.mlp = &a.net_output; r.Wf = p.Wf_out; r.bf = p.bf_out;
    r.M = p.M;
    int wrow = 0;
    if (a.focus == GNN_FOCUS_ARC) {
        k_arc_endpoints<<<cdiv(p.M, 256), 256, 0, st>>>(a.out_index, a.arc_src, a.arc_dst, p.M, p.idx_src, p.idx_dst);
        LAUNCH_OK();
        const int *ends[2] = {p.idx_src, p.idx_dst};
        for (int s = 0; s < 2; ++s) {
            r.segs[r.nseg++] = gnn::Seg{a.state_out, ends[s], p.S, p.S, wrow}; wrow += p.S;
            if (with_labels) { r.segs[r.nseg++] = gnn::Seg{a.nodes, ends[s], a.ld_nodes, p.L, wrow}; wrow += p.L; }
        }
        if (p.A > 0) r.segs[r.nseg++] = gnn::Seg{a.arc_labels, a.out_index, a.ld_arcs, p.A, wrow};
    } else {
        r.segs[r.nseg++] = gnn::Seg{a.state_out, a.out_index, p.S, p.S, 0};
        if (with_labels) r.segs[r.nseg++] = gnn::Seg{a.nodes, a.out_index, a.ld_nodes, p.L, p.S};
    }
    r.hid[0] = p.ohid[0]; r.hid[1] = p.ohid[1]; r.ld_hid = p.ld_ohid;
    if (a.focus == GNN_FOCUS_GRAPH) { r.Y = p.out_nodes; r.ldy = p.Tout; }
    else                            { r.Y = a.out;       r.ldy = p.Tout; }
    TRY(run_mlp(r, st));
    if (a.focus == GNN_FOCUS_GRAPH)   // GNN.py:345: NodeGraph^T . out_nodes
        TRY(launch_aggregate(nullptr, a.nodegraph, p.out_nodes, p.Tout, p.Tout, a.out, p.Tout, st));
    return 0;
}

// Which generation of the fused iteration kernel runs.  GNN_FUSED_KERNEL unset / 0 = automatic:
//   4  wave-specialised (12 gather waves + 4 matrix waves per workgroup, 32 waves per CU)      <- d > 16 and >= 32768 nodes
//   2  phase-alternating (every wave gathers, then every wave multiplies; 16 waves per CU)      <- d <= 16 or small graphs
//      (measured crossover on ER graphs with 10 arcs / node, d = 64: 17.2 vs 16.4 us at 2e3 nodes, 28.3 vs 28.5 at 3e4,
//       60 vs 74 at 1e5, 480 vs 542 at 1e6)
// (a software-pipelined generation 3 lost to both at every size - profiles/r01_gather_sweep.txt - and was removed)
// A tuning knob, never a correctness switch: both are held to the same parity tests.
int fused_generation(int SP, int n_nodes, int flags) {
    const int pinned = (flags & GNN_FLAG_FUSED_GEN_MASK) >> 4;
    if (pinned == 2 || pinned == 4) return pinned;
    static int v = -1;
    if (v < 0) {
        const char *e = getenv("GNN_FUSED_KERNEL");
        v = e ? atoi(e) : 0;
        if (v != 2 && v != 4) v = 0;
    }
    return v ? v : ((SP > 16 && n_nodes >= 32768) ? 4 : 2);
}

int device_cus() {
    static int n_cu = 0;
    if (n_cu == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 256;
        n_cu = prop.multiProcessorCount;
    }
    return n_cu;
}

// the fused kernel that runs ONE iteration: the size-based / pinned choice, except that two-layer state networks go to the
// wave-specialised kernel at any size (the only per-iteration kernel that carries a second Dense) unless pinned elsewhere
// the state network's LAST activation is a softmax (legal: reference MLP.py:12-78 takes any Keras activation): only the
// wave-specialised kernel's row-major epilogue has whole rows in one lane group, so that kernel runs whatever is pinned
bool state_softmax(const gnn_loop_args_t &a, const Plan &p) {
    for (int t = 0; t < p.T; ++t) {
        const gnn_mlp_t &m = a.net_state[t];
        if (m.n_layers >= 1 && m.activation[m.n_layers - 1] == GNN_ACT_SOFTMAX) return true;
    }
    return false;
}

int iteration_generation(const gnn_loop_args_t &a, const Plan &p) {
    if (state_softmax(a, p)) return 4;
    const int gen = fused_generation(p.SP, p.N, a.flags);
    bool two = false;
    for (int t = 0; t < p.T; ++t) two |= a.net_state[t].n_layers == 2;
    if (!two) return gen;
    const int pinned = (a.flags & GNN_FLAG_FUSED_GEN_MASK) >> 4;
    static int env = -1;
    if (env < 0) { const char *e = getenv("GNN_FUSED_KERNEL"); env = e ? atoi(e) : 0; }
    const int want = pinned ? pinned : env;
    return (p.SP > 16 && (want == 0 || want >= 4)) ? 4 : gen;
}

// what a fused kernel needs to know about node type t (the second Dense of a two-layer state network included)
gnn::FusedType fused_type(const gnn_loop_args_t &a, const Plan &p, int t) {
    const gnn_mlp_t &m = a.net_state[t];
    const bool two = m.n_layers == 2;
    return gnn::FusedType{p.tp[t].rows, p.tp[t].count, p.tp[t].Wf, p.tp[t].wrow_state, p.tp[t].wrow_agg,
                          (int)m.units[0], (int)m.activation[0],
                          two ? m.kernel[1] : nullptr, two ? m.bias[1] : nullptr, two ? (int)m.activation[1] : 0, p.tp[t].Wc, 0};
}

// State networks with two or more hidden layers (reference MLP.py:83-139 `hidden_units=[h1, h2, ..]`): the first TWO Dense layers run
// fused with the aggregate in the wave-specialised kernel's two-layer form, which then writes the second hidden layer's activations
// (not a state) - the remaining layers are plain dense launches over that matrix, the last one carrying the predicate.  Homogeneous
// models on one GPU, hidden widths within the padded state width.  Un-fused before: aggregate + one dense launch per layer.
bool prefix_applies(const gnn_loop_args_t &a, const Plan &p) {
    if (a.flags & GNN_FLAG_UNFUSED) return false;
    static int off = -1;
    if (off < 0) { const char *e = getenv("GNN_PREFIX_FUSION"); off = (e && e[0] == '0') ? 1 : 0; }
    if (off || p.composite || p.T != 1 || p.tp[0].rows || a.nodes_src || p.n_groups > 0) return false;
    const gnn_mlp_t &m = a.net_state[0];
    if (m.n_layers < 3 || (p.SP != 32 && p.SP != 64)) return false;
    if (m.units[0] > p.SP || m.units[1] > p.SP || m.activation[0] == GNN_ACT_SOFTMAX || m.activation[1] == GNN_ACT_SOFTMAX) return false;
    const int pinned = (a.flags & GNN_FLAG_FUSED_GEN_MASK) >> 4;
    if (pinned != 0 && pinned != 4) return false;
    if ((size_t)(p.N + p.n_heavy) * p.SP * 4 >= ((size_t)1 << 32) || (size_t)p.N * p.ldC * 4 >= ((size_t)1 << 32)) return false;
    if ((size_t)iter_adjacency(a, p).nnz * 4 >= ((size_t)1 << 32)) return false;
    return true;
}

int iteration_prefix(const gnn_loop_args_t &a, const Plan &p, const int *gate, const float *src, float *dst, int *flag_next, float *k_out,
                     float k_val, hipStream_t st) {
    TRY(launch_heavy(a, p, gate, src, st));
    const gnn_csr_t &adj = iter_adjacency(a, p);
    const gnn_mlp_t &m = a.net_state[0];
    gnn::Fused2Args fa;
    memset(&fa, 0, sizeof(fa));
    fa.gate = gate; fa.n_gate = gate ? 1 : 0; fa.gate_stride = 0;
    fa.rowptr = adj.rowptr; fa.src = adj.src; fa.w = adj.w; fa.row_scale = adj.row_scale;
    fa.state_in = src; fa.state_out = p.agg; fa.row_base = 0;          // p.agg: [N, SP], free on this path, receives the hidden activations
    fa.C = p.C; fa.ldC = p.ldC;
    fa.n_types = 1;
    fa.tp[0] = gnn::FusedType{nullptr, p.tp[0].count, p.tp[0].Wf, p.tp[0].wrow_state, p.tp[0].wrow_agg, (int)m.units[0], (int)m.activation[0],
                              m.kernel[1], m.bias[1], (int)m.activation[1], nullptr, (int)m.units[1]};
    fa.S = p.S; fa.thr = a.state_threshold;
    fa.flag_next = nullptr; fa.k_out = nullptr; fa.err = p.err;
    FUSED_OK(gnn::launch_fused4(fa, p.SP, device_cus(), st));
    GNN_SET_KERNEL_NAME("k_state_fused4<.., L2> (aggregate + two Dense layers) + k_segdense (remaining layers)");
    gnn_mlp_t tail;
    memset(&tail, 0, sizeof(tail));
    tail.n_layers = m.n_layers - 2; tail.in_dim = m.units[1];
    for (int l = 0; l < tail.n_layers; ++l) { tail.units[l] = m.units[l + 2]; tail.activation[l] = m.activation[l + 2]; tail.kernel[l] = m.kernel[l + 2]; tail.bias[l] = m.bias[l + 2]; }
    MlpRun r;
    r.mlp = &tail; r.Wf = tail.kernel[0]; r.bf = tail.bias[0];
    r.nseg = 1;
    r.segs[0] = gnn::Seg{p.agg, nullptr, p.SP, (int)m.units[1], 0};
    r.M = p.N;
    r.hid[0] = p.hid[0]; r.hid[1] = p.hid[1]; r.ld_hid = p.ld_hid;
    r.Y = dst; r.ldy = p.SP;
    r.gate = gate;
    if (flag_next) { r.pred_old = src; r.ld_pred = p.SP; r.pred_thr = a.state_threshold; r.pred_flag = flag_next; r.pred_k = k_out; r.pred_kval = k_val; }
    TRY(run_mlp(r, st));
    if (flag_next && !r.pred_fused) TRY(launch_converge(gate, dst, src, p.N, p.S, p.SP, p.SP, a.state_threshold, flag_next, k_out, k_val, st));
    return 0;
}

// one fused iteration over every node type (one launch per type)
int iteration_fused(const gnn_loop_args_t &a, const Plan &p, const int *gate, int n_gate, int gate_stride,
                    const float *src, float *dst, int row_base, int *flag_next, float *k_out, float k_val,
                    hipStream_t st, const gnn_csr_t *adj_override = nullptr, const float *agg_init = nullptr,
                    const int *rows_override = nullptr, int count_override = 0, const gnn_peer_set_t *peers = nullptr) {
    auto type_of = [&](int t) { return fused_type(a, p, t); };
    TRY(launch_heavy(a, p, n_gate == 1 ? gate : nullptr, src, st));
    const gnn_csr_t &adj = adj_override ? *adj_override : iter_adjacency(a, p);
    gnn::Fused2Args fa;
    memset(&fa, 0, sizeof(fa));
    fa.gate = n_gate ? gate : nullptr; fa.n_gate = n_gate; fa.gate_stride = gate_stride;
    fa.rowptr = adj.rowptr; fa.src = adj.src; fa.w = adj.w; fa.row_scale = adj.row_scale;
    fa.state_in = src; fa.state_out = dst; fa.row_base = row_base;
    fa.C = p.C; fa.ldC = p.ldC;
    fa.n_types = 0;
    for (int t = 0; t < p.T; ++t)
        if (p.tp[t].count > 0) fa.tp[fa.n_types++] = type_of(t);
    if (rows_override && fa.n_types == 1) {      // a SUB-RANGE of a homogeneous shard's rows (the pipelined exchange): the row-list form
        fa.tp[0].rows = rows_override; fa.tp[0].count = count_override;
        if (count_override == 0) fa.n_types = 0;
    }
    fa.S = p.S; fa.thr = a.state_threshold;
    fa.flag_next = flag_next;
    fa.k_out = k_out; fa.k_val = k_val;
    fa.err = p.err;
    fa.agg_init = agg_init;
    if (peers) {
        if (peers->n_peers < 0 || peers->n_peers > GNN_MAX_PEERS) return fail("n_peers %d out of [0, %d]", peers->n_peers, GNN_MAX_PEERS);
        fa.n_peers = peers->n_peers;
        for (int i = 0; i < peers->n_peers; ++i) fa.peer_out[i] = peers->state_out_full[i];
    }
    if (p.xc_ok && !adj.w && (agg_init || fused_generation(p.SP, p.N, a.flags) == 4)) fa.Xc = p.Xc;
    if (fa.n_types == 0) {                      // no nodes at all: only the iteration counter moves
        if (k_out) TRY(launch_converge(fa.gate, src, src, 0, p.S, p.SP, p.SP, a.state_threshold, flag_next, k_out, k_val, st));
        return 0;
    }
    if (fa.n_peers > 0 && (p.SP == 128 || p.T != 1 || adj.w || a.net_state[0].n_layers != 1 || (p.SP != 32 && p.SP != 64) || p.n_heavy != 0))
        return fail("peer stores: homogeneous one-layer shards without per-arc weights or hub rows, state widths 17 .. 64");
    if (p.SP == 128) { FUSED_OK(gnn::launch_wide(fa, device_cus(), st, 0)); return 0; }
    const int gen = (agg_init || fa.n_peers > 0) ? 4 : iteration_generation(a, p);
    if (gen == 4) FUSED_OK(gnn::launch_fused4(fa, p.SP, device_cus(), st));
    else FUSED_OK(gnn::launch_fused2(fa, p.SP, 8, device_cus(), st));
    return 0;
}

// The whole loop in ONE launch (kernel_state_small.hpp) for graphs of at most 64 nodes per CU.  `persistent_applies` is
// the static part of the decision (hub rows, d <= 16, too many tiles, pinned to another kernel: one launch per iteration).
bool persistent_applies(const gnn_loop_args_t &a, const Plan &p) {
    const int pinned = (a.flags & GNN_FLAG_FUSED_GEN_MASK) >> 4;
    if (pinned != 0 && pinned != 5) return false;
    static int env = -1;
    if (env < 0) { const char *e = getenv("GNN_FUSED_KERNEL"); env = e ? atoi(e) : 0; }
    if (pinned == 0 && env != 0 && env != 5) return false;
    if (p.n_heavy != 0 || a.max_iteration < 1 || (p.SP != 16 && p.SP != 32 && p.SP != 64)) return false;
    int tiles = 0;
    for (int t = 0; t < p.T; ++t) tiles += (p.tp[t].count + 63) / 64;
    if (p.n_groups > 0) tiles = p.group_tiles;
    return tiles > 0 && tiles <= std::min(device_cus(), GNN_SMALL_MAX_TILES);
}

// returns 2 when the kernel does not apply after all and the caller launches per iteration.  `pred0` (optional): the
// predicate of state_0 as one word per tile; `state_final` (optional): the caller's compact result buffer.
int loop_persistent(const gnn_loop_args_t &a, const Plan &p, const float *first, float *const B[2], const int *pred0,
                    int n_pred0, float *state_final, hipStream_t st) {
    if (!persistent_applies(a, p)) return 2;
    gnn::SmallArgs sa;
    memset(&sa, 0, sizeof(sa));
    gnn::Fused2Args &fa = sa.f;
    const gnn_csr_t &adj = iter_adjacency(a, p);
    fa.rowptr = adj.rowptr; fa.src = adj.src; fa.w = adj.w; fa.row_scale = adj.row_scale;
    fa.state_in = first; fa.row_base = 0;
    fa.C = p.C; fa.ldC = p.ldC;
    for (int t = 0; t < p.T; ++t)
        if (p.tp[t].count > 0)
            fa.tp[fa.n_types++] = fused_type(a, p, t);
    if (fa.n_types == 0) return 2;
    fa.S = p.S; fa.thr = a.state_threshold; fa.k_out = a.k_out;
    sa.buf[0] = B[0]; sa.buf[1] = B[1];
    sa.max_iteration = a.max_iteration;
    sa.no_exit = (a.flags & GNN_FLAG_NO_EARLY_EXIT) != 0;
    sa.flags = p.flags;
    sa.pred0 = pred0; sa.n_pred0 = n_pred0;
    sa.state_final = state_final; sa.ld_final = p.S;
    sa.bar = reinterpret_cast<unsigned long long *>(p.flags + ((a.max_iteration + 3) & ~1));     // 8-byte aligned (flags is 256-B aligned)
    if (p.n_groups > 0) {              // one pair of arrival counters per group: the (otherwise idle) mid-size kernel's lines
        static_assert(4 * GNN_MAX_GROUPS <= gnn::MID_BAR_WORDS, "group counters must fit the zeroed barrier words");
        sa.groups = p.gt;
        sa.bar = reinterpret_cast<unsigned long long *>(p.mid_bar);
    }
    const int rc = gnn::launch_small(sa, p.SP, device_cus(), st, p.n_groups > 0 ? p.group_tiles : 0);
    if (rc == 1) return fail("persistent loop kernel: launch failed (%s)", hipGetErrorString(hipGetLastError()));
    return rc;
}

// Mid-size graphs: the whole loop in one launch with several tiles per workgroup (kernel_state_mid.hpp).  Between the
// small whole-loop kernel's range (one tile per CU) and GNN_MID_MAX_NODES.  The default is where one launch per iteration
// of the wave-specialised kernel catches up (profiles/r02_mid_sweep.txt, 10 arcs per node: d = 64 25.4 vs 28.5 us at
// 30 000 nodes, 40.1 vs 35.7 at 45 000; d = 32 16.4 vs 17.6 at 30 000, 26.3 vs 26.1 at 60 000); pinned with
// GNN_FLAG_FUSED_GEN6 at any size.
int mid_max_nodes() {
    static int v = -1;
    if (v < 0) { const char *e = getenv("GNN_MID_MAX_NODES"); v = e ? atoi(e) : 36000; }
    return v;
}

bool mid_applies(const gnn_loop_args_t &a, const Plan &p) {
    const int pinned = (a.flags & GNN_FLAG_FUSED_GEN_MASK) >> 4;
    if (pinned != 0 && pinned != 6) return false;
    static int env = -1;
    if (env < 0) { const char *e = getenv("GNN_FUSED_KERNEL"); env = e ? atoi(e) : 0; }
    if (pinned == 0 && env != 0 && env != 6) return false;
    if (p.n_heavy != 0 || a.max_iteration < 1 || (p.SP != 32 && p.SP != 64) || p.N == 0) return false;
    return pinned == 6 || env == 6 || p.N <= mid_max_nodes();
}

int loop_mid(const gnn_loop_args_t &a, const Plan &p, const float *first, float *const B[2], hipStream_t st) {
    gnn::SmallArgs sa;
    memset(&sa, 0, sizeof(sa));
    gnn::Fused2Args &fa = sa.f;
    const gnn_csr_t &adj = iter_adjacency(a, p);
    fa.rowptr = adj.rowptr; fa.src = adj.src; fa.w = adj.w; fa.row_scale = adj.row_scale;
    fa.state_in = first; fa.row_base = 0;
    fa.C = p.C; fa.ldC = p.ldC;
    for (int t = 0; t < p.T; ++t)
        if (p.tp[t].count > 0)
            fa.tp[fa.n_types++] = fused_type(a, p, t);
    if (fa.n_types == 0) return 2;
    fa.S = p.S; fa.thr = a.state_threshold; fa.k_out = a.k_out;
    sa.buf[0] = B[0]; sa.buf[1] = B[1];
    sa.max_iteration = a.max_iteration;
    sa.no_exit = (a.flags & GNN_FLAG_NO_EARLY_EXIT) != 0;
    sa.flags = p.flags;
    sa.bar = reinterpret_cast<unsigned long long *>(p.mid_bar);
    const int rc = gnn::launch_mid(sa, p.SP, device_cus(), st);
    if (rc == 1) return fail("mid-size loop kernel: launch failed (%s)", hipGetErrorString(hipGetLastError()));
    return rc;
}

// Groups whose whole state fits the LDS of one CU: one workgroup per group, no grid barrier (kernel_state_lds.hpp).
bool lds_applies(const gnn_loop_args_t &a, const Plan &p) {
    const int pinned = (a.flags & GNN_FLAG_FUSED_GEN_MASK) >> 4;
    static int env = -1;
    if (env < 0) { const char *e = getenv("GNN_FUSED_KERNEL"); env = e ? atoi(e) : 0; }
    if ((pinned != 0 && pinned != 7) || (pinned == 0 && env != 0 && env != 7)) return false;
    if (p.n_groups < 1 || p.composite || p.n_heavy != 0 || a.max_iteration < 1 || (p.SP != 16 && p.SP != 32)) return false;
    if (a.net_state[0].n_layers != 1 || a.net_state[0].activation[0] == GNN_ACT_SOFTMAX || (a.flags & GNN_FLAG_UNFUSED)) return false;
    if (p.has_sets && p.n_groups > device_cus()) return false;      // groups that wait for each other must all be resident
    return gnn::lds_group_fits(p.group_max_nodes, p.SP);
}

int loop_lds(const gnn_loop_args_t &a, const Plan &p, hipStream_t st) {
    gnn::LdsArgs la;
    memset(&la, 0, sizeof(la));
    la.node_begin = p.d_group_tabs; la.tile64_begin = p.d_group_tabs + (p.n_groups + 1);
    la.set_first = p.d_group_tabs + 2 * (p.n_groups + 1); la.set_size = p.d_group_tabs + 3 * (p.n_groups + 1);
    la.set_bar = p.has_sets ? p.set_bar : nullptr;
    la.pred0 = p.pred0;
    la.rowptr = a.adjacency.rowptr; la.src = a.adjacency.src; la.w = a.adjacency.w; la.row_scale = a.adjacency.row_scale;
    la.state0 = a.state_dim > 0 ? a.state0 : a.nodes; la.ld_s0 = a.state_dim > 0 ? p.S : a.ld_nodes;
    la.C = p.C; la.ldC = p.ldC;
    la.Wf = p.tp[0].Wf; la.wrow_state = p.tp[0].wrow_state; la.wrow_agg = p.tp[0].wrow_agg;
    la.H = a.net_state[0].units[0]; la.act = a.net_state[0].activation[0];
    la.S = p.S; la.max_iteration = a.max_iteration; la.no_exit = (a.flags & GNN_FLAG_NO_EARLY_EXIT) != 0;
    la.thr = a.state_threshold;
    la.stage = p.buf[0];
    la.state_out = a.state_out; la.k_out = a.k_out;
    const int rc = gnn::launch_lds(la, p.SP, p.n_groups, p.group_max_nodes, st);
    if (rc == 1) return fail("LDS-resident loop kernel: launch failed (%s)", hipGetErrorString(hipGetLastError()));
    return rc;
}

// 0: un-fused kernels; 1: any fused kernel; 2: two-layer state networks - only the wave-specialised kernel and the
// persistent whole-loop kernel carry the second Dense; 3: state width 65 .. 128 - the wide kernel, whatever generation is pinned.
int fusable(const gnn_loop_args_t &a, const Plan &p) {
    if (a.flags & GNN_FLAG_UNFUSED) return 0;
    if (p.SP > 128) return 0;      // W1 = [2SP x SP] floats must fit the CU's LDS: 128 KB at SP = 128 (kernel_state_wide.hpp)
    // the fused kernel addresses state rows and C with 32-bit byte offsets off a scalar base
    if ((size_t)(std::max(a.adjacency.n_src, p.N) + p.n_heavy) * p.SP * 4 >= ((size_t)1 << 32) || (size_t)p.N * p.ldC * 4 >= ((size_t)1 << 32)) return 0;
    // ... and the CSR arrays through 4 GiB buffer windows
    if ((size_t)iter_adjacency(a, p).nnz * 4 >= ((size_t)1 << 32) || ((size_t)p.N + 1) * 4 >= ((size_t)1 << 32)) return 0;
    // One Dense layer everywhere, or two with at most SP hidden units (the matrix waves have the time, and LDS holds a
    // second weight matrix in place of two ring slots).
    bool two = false;
    for (int t = 0; t < p.T; ++t) {
        const gnn_mlp_t &m = a.net_state[t];
        if (m.n_layers < 1 || m.n_layers > 2) return 0;
        if (m.activation[m.n_layers - 1] == GNN_ACT_SOFTMAX && p.SP > 64) return 0;   // a softmax state: the wave-specialised kernel only (widths up to 64)
        if (m.n_layers == 2 && (m.activation[0] == GNN_ACT_SOFTMAX || m.units[0] > p.SP)) return 0;
        if (t > 0 && (m.n_layers == 2) != two) return 0;                // every node type the same depth
        two = m.n_layers == 2;
    }
    if (p.SP == 128) return two ? 0 : 3;     // state widths 65 .. 128: one-layer state networks on the wide kernel
    return two ? 2 : 1;
}

// State widths 129 .. 256 (kernel_state_xwide.hpp): homogeneous graphs, one Dense layer, no hub split (make_plan keeps hub rows
// off these widths), every array within the 32-bit byte offsets of a buffer window.  GNN_XWIDE=0 keeps the un-fused path.
bool xwide_applies(const gnn_loop_args_t &a, const Plan &p) {
    static int env = -1;
    if (env < 0) { const char *e = getenv("GNN_XWIDE"); env = (e && atoi(e) == 0) ? 0 : 1; }
    if (!env || (a.flags & GNN_FLAG_UNFUSED)) return false;
    if (p.SP <= 128 || p.S > 256 || p.composite || p.T != 1 || p.tp[0].rows || p.n_heavy != 0 || p.N < 1 || !p.Wx) return false;
    if (a.n_heavy_segments > 0) return false;      // hub rows (the caller's split says so): one gather wave would walk a hub two rows at a time

    const gnn_mlp_t &m = a.net_state[0];
    if (m.n_layers != 1 || m.activation[0] == GNN_ACT_SOFTMAX || (int)m.units[0] != p.S) return false;
    if ((size_t)std::max(a.adjacency.n_src, p.N) * p.SP * 4 >= ((size_t)1 << 32) || (size_t)p.N * p.ldC * 4 >= ((size_t)1 << 32)) return false;
    if ((size_t)a.adjacency.nnz * 4 >= ((size_t)1 << 32) || ((size_t)p.N + 1) * 4 >= ((size_t)1 << 32)) return false;
    return true;
}

#ifdef XB_EXPERIMENT
int xwide_matrix_waves() { const char *e = getenv("GNN_XWIDE_MW"); return e ? atoi(e) : 0; }
#else
int xwide_matrix_waves() { return 0; }
#endif

int setup_xwide(const gnn_loop_args_t &a, const Plan &p, hipStream_t st) {
    return gnn::launch_xwide_weights_b3(p.tp[0].Wf, (int)a.net_state[0].units[0], p.S, p.tp[0].wrow_state, p.tp[0].wrow_agg, p.SP, p.Wx, st);
}

int iteration_xwide(const gnn_loop_args_t &a, const Plan &p, const int *gate, const float *src, float *dst, int *flag_next, float *k_out,
                    float k_val, hipStream_t st) {
    gnn::XWideArgs xa;
    memset(&xa, 0, sizeof(xa));
    xa.gate = gate; xa.n_gate = gate ? 1 : 0; xa.gate_stride = 0;
    xa.rowptr = a.adjacency.rowptr; xa.src = a.adjacency.src; xa.w = a.adjacency.w; xa.row_scale = a.adjacency.row_scale;
    xa.state_in = src; xa.state_out = dst;
    xa.C = p.C; xa.ldC = p.ldC; xa.Wb = p.Wx;
    xa.N = p.N; xa.S = p.S; xa.SP = p.SP;
    xa.act = (int)a.net_state[0].activation[0];
    xa.thr = a.state_threshold; xa.flag_next = flag_next; xa.k_out = k_out; xa.k_val = k_val; xa.err = p.err;
#ifdef XB_EXPERIMENT
    { const char *e = getenv("GNN_XB_DBG"); xa.dbg = e ? atoi(e) : 0; }
#endif
    FUSED_OK(gnn::launch_xwide_b3(xa, device_cus(), xwide_matrix_waves(), st));
    return 0;
}

// may ONE ITERATION of this model run in a fused launch? (the per-iteration entry points and the loop's fallback)
bool can_fuse(const gnn_loop_args_t &a, const Plan &p) {
    const int f = fusable(a, p);
    return f == 1 || f == 3 || (f == 2 && iteration_generation(a, p) == 4);
}

}  // namespace

// =====================================================================================================================
extern "C" {

#ifdef GNN_F4_PROFILE
// experiment-only: cycle totals of the wave-specialised kernel's phases (not part of the ABI)
int gnn_f4_profile(unsigned long long *out8, int reset) {
    if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(gnn::g_f4_prof), 64) != hipSuccess) return 1;
    if (reset) { unsigned long long z[8] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(gnn::g_f4_prof), z, 64) != hipSuccess) return 1; }
    return 0;
}
#endif
#ifdef XW_PROFILE
// experiment-only: shader-clock totals of k_state_xwide's phases (not part of the ABI)
int gnn_xw_profile(unsigned long long *out8, int reset) {
    if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(gnn::g_xw_prof), 64) != hipSuccess) return 1;
    if (reset) { unsigned long long z[8] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(gnn::g_xw_prof), z, 64) != hipSuccess) return 1; }
    return 0;
}
#endif
#ifdef GNN_F4_TIMELINE
// per-workgroup timestamps of the last wave-specialised launch: out[1024][4] = entry, after the W1 fill, first deposit, exit
int gnn_f4_wg_times(unsigned long long *out4096) {
    return hipMemcpyFromSymbol(out4096, HIP_SYMBOL(gnn::g_f4_wg), 1024 * 4 * 8) == hipSuccess ? 0 : 1;
}
#endif

const char *gnn_last_error(void) { return g_err; }
const char *gnn_last_kernel_name(void) { return gnn::last_kernel_name(); }
int gnn_abi_version(void) { return GNN_ABI_VERSION; }
size_t gnn_struct_size(int which) {
    switch (which) {
        case 0: return sizeof(gnn_csr_t);
        case 1: return sizeof(gnn_mlp_t);
        case 2: return sizeof(gnn_loop_args_t);
        case 3: return offsetof(gnn_loop_args_t, flags);
        case 4: return sizeof(gnn_train_args_t);
        case 5: return offsetof(gnn_train_args_t, tape);
        case 6: return sizeof(gnn_ragged_desc_t);
        case 7: return sizeof(gnn_shard_loop_args_t);
        default: return 0;
    }
}

size_t gnn_loop_workspace_bytes(const gnn_loop_args_t *args) {
    if (!args) { fail("args is NULL"); return 0; }
    Plan p;
    if (make_plan(*args, nullptr, p, false)) return 0;
    return p.bytes;
}

int gnn_loop_forward(const gnn_loop_args_t *args) {
    if (!args) return fail("args is NULL");
    const gnn_loop_args_t &a = *args;
    Plan p;
    TRY(make_plan(a, a.workspace, p, true));
    if (!a.workspace || a.workspace_bytes < p.bytes) return fail("workspace too small: %zu < %zu bytes", a.workspace_bytes, p.bytes);
    if (((uintptr_t)a.workspace & 255) != 0) return fail("workspace must be 256-byte aligned");
    hipStream_t st = (hipStream_t)a.stream;

    // Graphs the whole-loop kernel covers: one set-up launch, the loop, the output stage.  Everything else: the general
    // set-up (one launch per constant) and one launch per iteration.
    const int fz = fusable(a, p);
    // Groups that fit the LDS of one CU: set-up launch, one workgroup per group, output stage.
    if (p.n_groups > 0 && fz == 1 && lds_applies(a, p) && setup_small_applies(a, p)) {
        // the group tables go to the device through a pinned staging buffer of this thread (an asynchronous copy out of pageable
        // memory would have to outlive this call); its event says when the previous call's copy has left it
        static thread_local struct { int *buf; size_t cap; hipEvent_t ev; } stage = {nullptr, 0, nullptr};
        const size_t n_tab = 4 * ((size_t)p.n_groups + 1);
        if (stage.ev) HIP_OK(hipEventSynchronize(stage.ev));
        else HIP_OK(hipEventCreateWithFlags(&stage.ev, hipEventDisableTiming));
        if (stage.cap < n_tab) {
            if (stage.buf) HIP_OK(hipHostFree(stage.buf));
            stage.buf = nullptr; stage.cap = 0;
            HIP_OK(hipHostMalloc((void **)&stage.buf, std::max<size_t>(n_tab, 4096) * sizeof(int), hipHostMallocDefault));
            stage.cap = std::max<size_t>(n_tab, 4096);
        }
        int *tabs = stage.buf;
        int tiles = 0;
        for (int g = 0; g <= p.n_groups; ++g) {
            tabs[g] = a.group_node_begin[g];
            tabs[p.n_groups + 1 + g] = tiles;
            tabs[2 * (p.n_groups + 1) + g] = g; tabs[3 * (p.n_groups + 1) + g] = 1;       // its own set unless listed below
            if (g < p.n_groups) tiles += (a.group_node_begin[g + 1] - a.group_node_begin[g] + 63) / 64;
        }
        for (int s2 = 0; s2 < a.n_group_sets; ++s2)
            for (int g = a.group_set_begin[s2]; g < a.group_set_begin[s2 + 1]; ++g) {
                tabs[2 * (p.n_groups + 1) + g] = a.group_set_begin[s2];
                tabs[3 * (p.n_groups + 1) + g] = a.group_set_begin[s2 + 1] - a.group_set_begin[s2];
            }
        HIP_OK(hipMemcpyAsync(p.d_group_tabs, tabs, n_tab * sizeof(int), hipMemcpyHostToDevice, st));
        if (p.has_sets) HIP_OK(hipMemsetAsync(p.set_bar, 0, sizeof(unsigned long long) * 2 * (size_t)p.n_groups, st));
        HIP_OK(hipEventRecord(stage.ev, st));
        Plan q = p;
        q.gt.n = 0;                                 // the set-up kernel reads the device tables
        TRY(setup_small(a, q, st));
        if (a.ev_loop_begin) HIP_OK(hipEventRecord((hipEvent_t)a.ev_loop_begin, st));
        const int rc = loop_lds(a, p, st);
        if (rc != 0) return rc == 2 ? fail("LDS-resident loop kernel does not cover this shape") : 1;
        if (a.ev_loop_end) HIP_OK(hipEventRecord((hipEvent_t)a.ev_loop_end, st));
        return output_stage(a, p, st);
    }
    if (p.has_sets) return fail("convergence group sets need the one-CU-per-group form (gnn_loop_groups_supported() != 2 for these args)");
    if (p.n_groups > GNN_MAX_GROUPS) return fail("more than %d convergence groups need every group to fit one CU's LDS (gnn_loop_groups_supported() != 2)", GNN_MAX_GROUPS);
    const bool soft = state_softmax(a, p);          // (one launch per iteration on the wave-specialised kernel)
    const bool whole_loop = fz != 0 && !soft && persistent_applies(a, p);
    const bool small_setup = whole_loop && setup_small_applies(a, p);
    if (p.n_groups > 0 && !small_setup) return fail("convergence groups need the whole-loop kernel (gnn_loop_groups_supported() == 0 for these args)");
    // every iteration on the XC form of the wave-specialised kernel (large homogeneous / composite graphs): C is never read
    const bool xc_loop = p.xc_ok && !whole_loop && fz == 1 && (soft || !mid_applies(a, p)) && p.SP != 128 && !iter_adjacency(a, p).w &&
                         iteration_generation(a, p) == 4 && fused_generation(p.SP, p.N, a.flags) == 4;
    if (small_setup) TRY(setup_small(a, p, st));
    else             TRY(setup_constants(a, p, st, /*zero_loop_words=*/true, /*skip_c=*/xc_loop));   // flags, barrier counters and k start from zero

    // state_0 (GNN.py:256-259) into the padded buffer; state_old_0 = ones is implicit in the first predicate (:261)
    // When the caller's state_0 already has the padded layout (d a multiple of 16, no hub rows behind the real ones) the
    // first iteration reads it in place: no copy of N x d floats.
    const float *first = p.buf[0];
    if (a.state_dim > 0 && p.S == p.SP && p.n_heavy == 0 && (reinterpret_cast<uintptr_t>(a.state0) & 15) == 0) first = a.state0;
    else if (a.state_dim > 0) TRY(launch_copy2d(nullptr, a.state0, p.S, p.buf[0], p.SP, p.N, p.S, p.SP, st));
    else                      TRY(launch_copy2d(nullptr, a.nodes, a.ld_nodes, p.buf[0], p.SP, p.N, p.S, p.SP, st));
    if (p.SP != p.S) HIP_OK(hipMemsetAsync(p.buf[1], 0, sizeof(float) * (size_t)p.N * p.SP, st));
    if (!small_setup) TRY(launch_converge(nullptr, first, nullptr, p.N, p.S, p.SP, 0, a.state_threshold, p.flags, nullptr, 0.f, st));

    const bool fused = can_fuse(a, p);
    const bool prefix = !fused && prefix_applies(a, p);
    const bool xwide = !fused && !prefix && xwide_applies(a, p);
    if (xwide) TRY(setup_xwide(a, p, st));
    const bool no_exit = (a.flags & GNN_FLAG_NO_EARLY_EXIT) != 0;
    // Ping-pong buffers.  When the caller's state_out has the padded layout too, it stands in for the buffer the LAST
    // iteration writes (B[max_iteration & 1]): a loop that runs to max_iteration leaves the result where the caller
    // wants it and the final copy below returns at once (an early stop may still need it).
    float *B[2] = {p.buf[0], p.buf[1]};
    if (first == a.state0 && a.state_out != a.state0 && (reinterpret_cast<uintptr_t>(a.state_out) & 15) == 0 && a.max_iteration > 0)
        B[a.max_iteration & 1] = a.state_out;
    if (a.ev_loop_begin) HIP_OK(hipEventRecord((hipEvent_t)a.ev_loop_begin, st));
    int persistent = 2;
    if (whole_loop) {                              // (two-layer state networks too); the kernel also writes state_out
        persistent = loop_persistent(a, p, first, B, small_setup ? p.pred0 : nullptr, small_setup ? p.group_tiles : 0, a.state_out, st);
        if (persistent == 1) return 1;
        if (persistent == 2 && small_setup) return fail("whole-loop kernel refused a graph its set-up kernel accepted");
    }
    bool loop_done = persistent == 0;
    if (!loop_done && fz == 1 && !soft && mid_applies(a, p)) {
        const int rc = loop_mid(a, p, first, B, st);
        if (rc == 1) return 1;
        loop_done = rc == 0;
    }
    for (int it = 0; !loop_done && it < a.max_iteration; ++it) {
        const int *gate = no_exit ? nullptr : p.flags + it;
        const float *src = it == 0 ? first : B[it & 1];
        float *dst = B[(it + 1) & 1];
        if (fused)       TRY(iteration_fused(a, p, gate, gate ? 1 : 0, 0, src, dst, 0, p.flags + it + 1, a.k_out, (float)(it + 1), st));
        else if (prefix) TRY(iteration_prefix(a, p, gate, src, dst, p.flags + it + 1, a.k_out, (float)(it + 1), st));
        else if (xwide)  TRY(iteration_xwide(a, p, gate, src, dst, p.flags + it + 1, a.k_out, (float)(it + 1), st));
        else             TRY(iteration_unfused(a, p, gate, src, dst, 0, p.flags + it + 1, a.k_out, (float)(it + 1), st));
    }

    if (a.ev_loop_end) HIP_OK(hipEventRecord((hipEvent_t)a.ev_loop_end, st));

    // converged state -> caller's compact [N, S] buffer; k (device) tells which of the two buffers holds it
    if (persistent != 0) {
        const long total = (long)p.N * p.S;
        if (total > 0) {
            gnn::k_select_state<<<std::min(cdiv(total, 256), 256 * 16), 256, 0, st>>>(a.k_out, first, B[0], B[1], p.SP, a.state_out, p.S, p.N, p.S);
            LAUNCH_OK();
        }
    }
    TRY(output_stage(a, p, st));
    if (!loop_done && (fused || xwide) && a.max_iteration > 0) TRY(launch_fold_error(p, a.k_out, st));
    return 0;
}

int gnn_loop_groups_supported(const gnn_loop_args_t *args) {
    if (!args || args->n_groups < 1) return 0;
    Plan p;
    if (make_plan(*args, nullptr, p, false)) return 0;
    const int fz = fusable(*args, p);
    if (fz == 1 && lds_applies(*args, p) && setup_small_applies(*args, p)) return 2;
    if (args->n_groups > GNN_MAX_GROUPS) return 0;
    return fz != 0 && !state_softmax(*args, p) && persistent_applies(*args, p) && setup_small_applies(*args, p) ? 1 : 0;
}

int gnn_aggregate(const gnn_csr_t *csr, const float *X, int32_t ldx, int32_t F, float *out, int32_t ldo, void *stream) {
    if (!csr) return fail("csr is NULL");
    if (F < 0 || ldx < F || ldo < F) return fail("bad F / leading dimensions");
    TRY(check_csr(*csr, "csr", csr->n_dst, csr->n_src));
    if (csr->n_dst > 0 && F > 0 && (!X || !out)) return fail("X / out is NULL");
    return launch_aggregate(nullptr, *csr, X, ldx, F, out, ldo, (hipStream_t)stream);
}

size_t gnn_mlp_workspace_bytes(const gnn_mlp_t *mlp, int32_t M) {
    if (!mlp || check_mlp(*mlp, "mlp", false)) return 0;
    Carver c(nullptr);
    c.take<float>((size_t)mlp->in_dim * mlp->units[0]);
    c.take<float>(mlp->units[0]);
    const int h = max_units(*mlp);
    c.take<float>((size_t)std::max(M, 1) * h);
    c.take<float>((size_t)std::max(M, 1) * h);
    return (c.off + 255) & ~(size_t)255;
}

int gnn_mlp_forward(const gnn_mlp_t *mlp, const float *X, int32_t ldx, int32_t M, float *Y, int32_t ldy,
                    void *workspace, size_t workspace_bytes, void *stream) {
    if (!mlp) return fail("mlp is NULL");
    TRY(check_mlp(*mlp, "mlp", true));
    if (M < 0 || ldx < mlp->in_dim || ldy < mlp->units[mlp->n_layers - 1]) return fail("bad M / leading dimensions");
    if (M > 0 && (!X || !Y)) return fail("X / Y is NULL");
    const size_t need = gnn_mlp_workspace_bytes(mlp, M);
    if (!workspace || workspace_bytes < need) return fail("workspace too small: %zu < %zu bytes", workspace_bytes, need);
    hipStream_t st = (hipStream_t)stream;
    Carver c(workspace);
    float *Wf = c.take<float>((size_t)mlp->in_dim * mlp->units[0]);
    float *bf = c.take<float>(mlp->units[0]);
    const int h = max_units(*mlp);
    MlpRun r;
    r.hid[0] = c.take<float>((size_t)std::max(M, 1) * h);
    r.hid[1] = c.take<float>((size_t)std::max(M, 1) * h);
    r.ld_hid = h;
    TRY(launch_fold(*mlp, Wf, bf, st));
    r.mlp = mlp; r.Wf = Wf; r.bf = bf;
    r.nseg = 1;
    r.segs[0] = gnn::Seg{X, nullptr, ldx, mlp->in_dim, 0};
    r.M = M; r.Y = Y; r.ldy = ldy;
    return run_mlp(r, st);
}

int gnn_converged(const float *state, const float *state_old, int32_t n, int32_t dim, int32_t ld, float threshold,
                  int32_t *flag, void *stream) {
    if (n < 0 || dim < 1 || ld < dim) return fail("bad n / dim / ld");
    if (!flag || (n > 0 && !state)) return fail("state / flag is NULL");
    hipStream_t st = (hipStream_t)stream;
    HIP_OK(hipMemsetAsync(flag, 0, sizeof(int32_t), st));
    return launch_converge(nullptr, state, state_old, n, dim, ld, ld, threshold, flag, nullptr, 0.f, st);
}

static int state_step_impl(const gnn_loop_args_t *args, const float *state_in, const GivenAgg *given, float *state_out, int32_t *flag_out) {
    if (!args) return fail("args is NULL");
    const gnn_loop_args_t &a = *args;
    Plan p;
    TRY(make_plan(a, a.workspace, p, false));
    TRY(check_csr(a.adjacency, "adjacency", p.N, p.N));
    if (!given) TRY(check_csr(a.arcnode, "arcnode", p.N, p.E));
    for (int t = 0; t < p.T; ++t) TRY(check_mlp(a.net_state[t], "net_state", true));
    if (p.composite) {      // one step of CompositeGNNnodeBased.convergence (CompositeGNN.py:215-234): per-type networks on per-type row lists
        if (!a.type_nodes && p.N > 0) return fail("type_nodes is NULL");
        if (a.type_offsets[0] != 0 || a.type_offsets[p.T] != p.N) return fail("type_offsets must span [0, n_nodes]");
        for (int t = 0; t < p.T && !given; ++t) TRY(check_csr(a.composite_adjacency[t], "composite_adjacency", p.N, p.N));
    }
    if (p.N > 0 && !a.nodes) return fail("nodes is NULL");
    if (!state_in || !state_out) return fail("state_in / state_out is NULL");
    if (given && p.N > 0) {
        if (p.A > 0 && (!given->arcs || given->ld_arcs < p.A)) return fail("aggregated_arcs is NULL or narrower than dim_arc_label = %d", p.A);
        if (p.ld_agg_nodes > 0 && (!given->nodes || given->ld_nodes < p.ld_agg_nodes))
            return fail("aggregated_nodes is NULL or narrower than the %d aggregated label columns", p.ld_agg_nodes);
    }
    if (!a.workspace || a.workspace_bytes < p.bytes) return fail("workspace too small: %zu < %zu bytes", a.workspace_bytes, p.bytes);
    hipStream_t st = (hipStream_t)a.stream;
    TRY(setup_constants(a, p, st, false, false, given));
    TRY(launch_copy2d(nullptr, state_in, p.S, p.buf[0], p.SP, p.N, p.S, p.SP, st));
    if (p.SP != p.S) HIP_OK(hipMemsetAsync(p.buf[1], 0, sizeof(float) * (size_t)p.N * p.SP, st));
    if (flag_out) HIP_OK(hipMemsetAsync(flag_out, 0, sizeof(int32_t), st));
    if (can_fuse(a, p))
        TRY(iteration_fused(a, p, nullptr, 0, 0, p.buf[0], p.buf[1], 0, flag_out, nullptr, 0.f, st));
    else if (xwide_applies(a, p)) {
        TRY(setup_xwide(a, p, st));
        TRY(iteration_xwide(a, p, nullptr, p.buf[0], p.buf[1], flag_out, nullptr, 0.f, st));
    } else
        TRY(iteration_unfused(a, p, nullptr, p.buf[0], p.buf[1], 0, flag_out, nullptr, 0.f, st));
    return launch_copy2d(nullptr, p.buf[1], p.SP, state_out, p.S, p.N, p.S, p.S, st);
}

int gnn_state_step(const gnn_loop_args_t *args, const float *state_in, float *state_out, int32_t *flag_out) {
    return state_step_impl(args, state_in, nullptr, state_out, flag_out);
}

int gnn_state_step_agg(const gnn_loop_args_t *args, const float *state_in, const float *aggregated_nodes, int32_t ld_aggregated_nodes,
                       const float *aggregated_arcs, int32_t ld_aggregated_arcs, float *state_out, int32_t *flag_out) {
    const GivenAgg g{aggregated_nodes, ld_aggregated_nodes, aggregated_arcs, ld_aggregated_arcs};
    return state_step_impl(args, state_in, &g, state_out, flag_out);
}

__global__ void k_or_flags(const int *gate, int n_gate, int gate_stride, int *out) {
    int v = 0;
    for (int i = threadIdx.x; i < n_gate; i += blockDim.x) v |= gate[(size_t)i * gate_stride] != 0;
    v = __syncthreads_or(v);
    if (threadIdx.x == 0) *out = v;
}

// Test hook (tests/test_gpu_round4.py): `n_workgroups` workgroups that each hold `lds_bytes` of LDS (160 KB = a whole CU) and one
// wave, doing nothing until `milliseconds` of wall-clock time have passed - a co-tenant that keeps CUs away from the persistent
// kernels.  Bounded by construction (the 100 MHz wall clock), so it cannot hang the GPU.
// `release` != NULL: the workgroups also leave as soon as *release != 0 (a word the test sets from another stream: the co-tenant goes when
// the TEST says so, not when a clock does - `ticks` stays as the bound that keeps the kernel from hanging the GPU)
__global__ void __launch_bounds__(64) k_debug_occupy(unsigned long long ticks, int *sink, const int *release) {
    extern __shared__ int occ_smem[];
    const unsigned long long t0 = wall_clock64();
    int polls = 0;
    while (wall_clock64() - t0 < ticks) {
        if (release && __hip_atomic_load(release, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0) break;      // (system scope: the word may be host memory)
        __builtin_amdgcn_s_sleep(64); ++polls;
    }
    if (threadIdx.x == 0 && polls < 0) { occ_smem[0] = polls; *sink = occ_smem[0]; }      // (keeps the LDS allocation alive; never taken)
}

int gnn_debug_occupy(int32_t n_workgroups, int32_t lds_bytes, int32_t milliseconds, void *stream) {
    if (n_workgroups < 1 || lds_bytes < 0 || lds_bytes > 160 * 1024 || milliseconds < 0 || milliseconds > 10000) return fail("gnn_debug_occupy: bad arguments");
    static bool attr = false;
    if (!attr) { HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_debug_occupy), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); attr = true; }
    k_debug_occupy<<<n_workgroups, 64, (size_t)lds_bytes, (hipStream_t)stream>>>((unsigned long long)milliseconds * 100000ull, nullptr, nullptr);
    LAUNCH_OK();
    return 0;
}

int gnn_debug_host_flag(int32_t **host_out, int32_t **dev_out) {
    static int *h = nullptr, *d = nullptr;
    if (!host_out || !dev_out) return fail("gnn_debug_host_flag: NULL pointer");
    if (!h) {
        HIP_OK(hipHostMalloc((void **)&h, 64, hipHostMallocMapped));
        HIP_OK(hipHostGetDevicePointer((void **)&d, h, 0));
    }
    *h = 0;
    *host_out = h; *dev_out = d;
    return 0;
}

int gnn_debug_expiry_beacon(int32_t **beacon_out, int32_t reset, void *stream) {
    if (!beacon_out) return fail("gnn_debug_expiry_beacon: beacon_out is NULL");
    void *ptr = nullptr;
    HIP_OK(hipGetSymbolAddress(&ptr, HIP_SYMBOL(gnn::g_wait_expired_beacon)));
    if (reset == 1) HIP_OK(hipMemsetAsync(ptr, 0, sizeof(int), (hipStream_t)stream));
    else if (reset == 2) HIP_OK(hipMemsetAsync(ptr, 1, sizeof(int), (hipStream_t)stream));       // (non-zero: "a wait has expired" raised by hand - a test's last resort)
    *beacon_out = (int32_t *)ptr;
    return 0;
}

int gnn_debug_occupy_until(int32_t n_workgroups, int32_t lds_bytes, int32_t max_milliseconds, const int32_t *release_flag, void *stream) {
    if (n_workgroups < 1 || lds_bytes < 0 || lds_bytes > 160 * 1024 || max_milliseconds < 0 || max_milliseconds > 20000 || !release_flag)
        return fail("gnn_debug_occupy_until: bad arguments");
    static bool attr = false;
    if (!attr) { HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_debug_occupy), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); attr = true; }
    k_debug_occupy<<<n_workgroups, 64, (size_t)lds_bytes, (hipStream_t)stream>>>((unsigned long long)max_milliseconds * 100000ull, nullptr, release_flag);
    LAUNCH_OK();
    return 0;
}

int32_t gnn_state_ld(int32_t state_width) { return state_width > 0 ? state_ld(state_width) : 0; }

int gnn_gather_rows(const float *src, int32_t ld_src, const int32_t *idx, int32_t M, int32_t width, float *dst,
                    int32_t ld_dst, void *stream) {
    if (M < 0 || width < 1 || ld_src < width || ld_dst < width) return fail("bad arguments");
    if (M == 0) return 0;
    if (!src || !idx || !dst) return fail("NULL pointer");
    const long total = (long)M * ((width & 3) == 0 ? width / 4 : width);
    gnn::k_gather_rows<<<std::min(cdiv(total, 256), 256 * 16), 256, 0, (hipStream_t)stream>>>(src, ld_src, idx, M, width, dst, ld_dst);
    LAUNCH_OK();
    return 0;
}

int gnn_ragged_copy(const gnn_ragged_desc_t *desc, int32_t n_desc, const int32_t *blk_begin, int32_t n_blocks, void *stream) {
    if (n_desc < 0 || n_blocks < 0) return fail("bad n_desc / n_blocks");
    if (n_desc == 0 || n_blocks == 0) return 0;
    if (!desc || !blk_begin) return fail("desc / blk_begin is NULL");
    static_assert(gnn::RC_CHUNK == GNN_RC_CHUNK, "chunk size of the header and the kernel differ");
    gnn::k_ragged_copy<<<n_blocks, 256, 0, (hipStream_t)stream>>>(desc, n_desc, blk_begin);
    LAUNCH_OK();
    return 0;
}

int gnn_shard_setup(const gnn_loop_args_t *args) {
    if (!args) return fail("args is NULL");
    const gnn_loop_args_t &a = *args;
    Plan p;
    TRY(make_plan(a, a.workspace, p, true));
    if (!a.nodes_src) return fail("gnn_shard_setup: nodes_src is NULL (not a shard)");
    if (!a.workspace || a.workspace_bytes < p.bytes) return fail("workspace too small: %zu < %zu bytes", a.workspace_bytes, p.bytes);
    hipStream_t st = (hipStream_t)a.stream;
    HIP_OK(hipMemsetAsync(p.flags, 0, sizeof(int) * (a.max_iteration + GNN_LOOP_WORDS), st));
    HIP_OK(hipMemsetAsync(a.k_out, 0, sizeof(float), st));
    return setup_constants(a, p, st);
}

int gnn_shard_iteration(const gnn_loop_args_t *args, const float *state_in_full, float *state_out_full,
                        int32_t row_base, const int32_t *gate, int32_t n_gate, int32_t gate_stride, int32_t *flag_out,
                        int32_t iteration) {
    if (!args) return fail("args is NULL");
    const gnn_loop_args_t &a = *args;
    if (!state_in_full || !state_out_full || !flag_out) return fail("state buffers / flag_out are NULL");
    if (iteration < 0 || iteration >= a.max_iteration) return fail("iteration %d out of [0, max_iteration)", iteration);
    if (n_gate < 0 || (n_gate > 0 && !gate)) return fail("bad gate list");
    Plan p;
    TRY(make_plan(a, a.workspace, p, false));
    if (row_base < 0 || row_base + p.N > a.adjacency.n_src) return fail("row_base out of the full buffer");
    hipStream_t st = (hipStream_t)a.stream;
    const int *g = nullptr;
    if (n_gate > 0 && !(a.flags & GNN_FLAG_NO_EARLY_EXIT)) {
        k_or_flags<<<1, 64, 0, st>>>(gate, n_gate, gate_stride, p.flags + iteration);
        LAUNCH_OK();
        g = p.flags + iteration;
    }
    HIP_OK(hipMemsetAsync(flag_out, 0, sizeof(int32_t), st));
    if (can_fuse(a, p))
        return iteration_fused(a, p, g, g ? 1 : 0, 0, state_in_full, state_out_full, row_base, flag_out, a.k_out, (float)(iteration + 1), st);
    return iteration_unfused(a, p, g, state_in_full, state_out_full, row_base, flag_out, a.k_out, (float)(iteration + 1), st);
}

// ---- the exchange inside the iteration kernel (include/gnnloop.h: gnn_shard_iteration_peers, gnn_peer_wait / _publish, gnn_ipc_*) --------
int gnn_shard_iteration_peers(const gnn_loop_args_t *args, const float *state_in_full, float *state_out_full, int32_t row_base,
                              const int32_t *gate, int32_t n_gate, int32_t gate_stride, int32_t *flag_out, int32_t iteration,
                              const gnn_peer_set_t *peers) {
    if (!args || !peers) return fail("args / peers is NULL");
    const gnn_loop_args_t &a = *args;
    if (!state_in_full || !state_out_full || !flag_out) return fail("state buffers / flag_out are NULL");
    if (iteration < 0 || iteration >= a.max_iteration) return fail("iteration %d out of [0, max_iteration)", iteration);
    if (n_gate < 0 || (n_gate > 0 && !gate)) return fail("bad gate list");
    Plan p;
    TRY(make_plan(a, a.workspace, p, false));
    if (row_base < 0 || row_base + p.N > a.adjacency.n_src) return fail("row_base out of the full buffer");
    if (!can_fuse(a, p)) return fail("peer stores need the fused iteration kernel");
    hipStream_t st = (hipStream_t)a.stream;
    const int *g = nullptr;
    if (n_gate > 0 && !(a.flags & GNN_FLAG_NO_EARLY_EXIT)) {
        k_or_flags<<<1, 64, 0, st>>>(gate, n_gate, gate_stride, p.flags + iteration);
        LAUNCH_OK();
        g = p.flags + iteration;
    }
    HIP_OK(hipMemsetAsync(flag_out, 0, sizeof(int32_t), st));
    return iteration_fused(a, p, g, g ? 1 : 0, 0, state_in_full, state_out_full, row_base, flag_out, a.k_out, (float)(iteration + 1), st,
                           nullptr, nullptr, nullptr, 0, peers);
}

// buffers whose BASE pointer a host can export (a framework's caching allocator hands out interior pointers of larger segments)
int gnn_device_malloc(void **device_ptr, size_t bytes) {
    if (!device_ptr || bytes == 0) return fail("gnn_device_malloc: bad arguments");
    HIP_OK(hipMalloc(device_ptr, bytes));
    HIP_OK(hipMemset(*device_ptr, 0, bytes));
    return 0;
}
int gnn_device_free(void *device_ptr) {
    if (device_ptr) HIP_OK(hipFree(device_ptr));
    return 0;
}

int gnn_ipc_export(const void *device_ptr, void *handle64) {
    static_assert(sizeof(hipIpcMemHandle_t) == 64, "hipIpcMemHandle_t is 64 bytes");
    if (!device_ptr || !handle64) return fail("gnn_ipc_export: NULL pointer");
    hipIpcMemHandle_t h;
    HIP_OK(hipIpcGetMemHandle(&h, const_cast<void *>(device_ptr)));
    memcpy(handle64, &h, 64);
    return 0;
}
int gnn_ipc_open(const void *handle64, void **device_ptr) {
    if (!handle64 || !device_ptr) return fail("gnn_ipc_open: NULL pointer");
    hipIpcMemHandle_t h;
    memcpy(&h, handle64, 64);
    HIP_OK(hipIpcOpenMemHandle(device_ptr, h, hipIpcMemLazyEnablePeerAccess));
    return 0;
}
int gnn_ipc_close(void *device_ptr) {
    if (!device_ptr) return 0;
    HIP_OK(hipIpcCloseMemHandle(device_ptr));
    return 0;
}

// every peer has published `value`: lane p polls arrive_local[p] (system scope: the peers' stores come from other processes / devices)
__global__ void __launch_bounds__(64) k_peer_wait(const int *arrive_local, int world, int rank, int value, unsigned long long wait_ticks, float *k_err) {
    const int p = threadIdx.x;
    if (p >= world || p == rank) return;
    const bool ok = gnn::wait_until(wait_ticks, [&]() { return __hip_atomic_load(arrive_local + p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) - value >= 0; });
    if (!ok && k_err) atomicAdd(k_err, -1.0e9f);              // (k < 0: an arrival never came - the caller's check_last_k raises)
}
// the slice's trailing flag row to every peer, then - behind a system-scope release - this rank's arrival word, everywhere
__global__ void __launch_bounds__(64) k_peer_publish(gnn_peer_set_t ps, int *arrive_local, const float *flag_row, long long row_off, int row_floats, int rank, int value) {
#pragma unroll
    for (int pi = 0; pi < GNN_MAX_PEERS; ++pi)
        if (pi < ps.n_peers)
            for (int c = threadIdx.x; c < row_floats; c += 64) ps.state_out_full[pi][row_off + c] = flag_row[c];
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_store(arrive_local + rank, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
#pragma unroll
        for (int pi = 0; pi < GNN_MAX_PEERS; ++pi)
            if (pi < ps.n_peers) __hip_atomic_store(ps.arrive[pi] + rank, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}
int gnn_peer_wait(const int32_t *arrive_local, int32_t world_size, int32_t rank, int32_t value, float *k_out_error, void *stream) {
    if (!arrive_local || world_size < 1 || world_size > GNN_MAX_PEERS + 1 || rank < 0 || rank >= world_size) return fail("gnn_peer_wait: bad arguments");
    if (world_size == 1) return 0;
    k_peer_wait<<<1, 64, 0, (hipStream_t)stream>>>(arrive_local, world_size, rank, value, gnn::wait_ticks(), k_out_error);
    LAUNCH_OK();
    return 0;
}
int gnn_peer_publish(const gnn_peer_set_t *peers, int32_t *arrive_local, const float *flag_row_local, int64_t flag_row_offset_floats,
                     int32_t row_floats, int32_t rank, int32_t value, void *stream) {
    if (!peers || !arrive_local || peers->n_peers < 0 || peers->n_peers > GNN_MAX_PEERS || rank < 0 || rank > GNN_MAX_PEERS)
        return fail("gnn_peer_publish: bad arguments");
    if (row_floats > 0 && !flag_row_local) return fail("gnn_peer_publish: flag_row_local is NULL");
    k_peer_publish<<<1, 64, 0, (hipStream_t)stream>>>(*peers, arrive_local, flag_row_local, (long long)flag_row_offset_floats, row_floats, rank, value);
    LAUNCH_OK();
    return 0;
}

// ---- overlap of the exchange with own-range work (gnnkeras_amd/distributed.py) ----------------------------
int gnn_shard_can_split(const gnn_loop_args_t *args) {
    if (!args) return 0;
    Plan p;
    if (make_plan(*args, args->workspace, p, false)) return 0;
    const int f = fusable(*args, p);
    if (p.n_heavy != 0) return 0;
    if (f == 3) return 1;                                          // state widths 65 .. 128: the wide kernel, whatever is pinned
    if (f != 1 || p.SP <= 16) return 0;                            // one-layer state nets on the wave-specialised kernel
    const int pinned = (args->flags & GNN_FLAG_FUSED_GEN_MASK) >> 4;
    return pinned == 0 || pinned == 4;
}

int gnn_shard_partial(const gnn_loop_args_t *args, const gnn_csr_t *adjacency_own, const float *state_in_full,
                      float *agg_partial) {
    if (!args || !adjacency_own) return fail("args / adjacency_own is NULL");
    const gnn_loop_args_t &a = *args;
    if (!state_in_full || !agg_partial) return fail("state_in_full / agg_partial is NULL");
    const int SP = state_ld(state_width(a));
    TRY(check_csr(*adjacency_own, "adjacency_own", a.n_nodes, adjacency_own->n_src));
    // the UN-SCALED (or per-arc weighted) sum: the row scale of 'average' / 'normalized' is applied once, by the iteration
    gnn_csr_t c = *adjacency_own;
    c.row_scale = nullptr;
    return launch_aggregate(nullptr, c, state_in_full, SP, SP, agg_partial, SP, (hipStream_t)a.stream);
}

int gnn_shard_iteration_split(const gnn_loop_args_t *args, const gnn_csr_t *adjacency_halo, const float *agg_partial,
                              const float *state_in_full, float *state_out_full, int32_t row_base, const int32_t *gate,
                              int32_t n_gate, int32_t gate_stride, int32_t *flag_out, int32_t iteration) {
    if (!args || !adjacency_halo || !agg_partial) return fail("args / adjacency_halo / agg_partial is NULL");
    const gnn_loop_args_t &a = *args;
    if (!state_in_full || !state_out_full || !flag_out) return fail("state buffers / flag_out are NULL");
    if (iteration < 0 || iteration >= a.max_iteration) return fail("iteration %d out of [0, max_iteration)", iteration);
    if (n_gate < 0 || (n_gate > 0 && !gate)) return fail("bad gate list");
    if (!gnn_shard_can_split(args)) return fail("gnn_shard_iteration_split: this model / shard does not run on the wave-specialised kernel");
    Plan p;
    TRY(make_plan(a, a.workspace, p, false));
    TRY(check_csr(*adjacency_halo, "adjacency_halo", p.N, a.adjacency.n_src));
    if (row_base < 0 || row_base + p.N > a.adjacency.n_src) return fail("row_base out of the full buffer");
    hipStream_t st = (hipStream_t)a.stream;
    const int *g = nullptr;
    if (n_gate > 0 && !(a.flags & GNN_FLAG_NO_EARLY_EXIT)) {
        k_or_flags<<<1, 64, 0, st>>>(gate, n_gate, gate_stride, p.flags + iteration);
        LAUNCH_OK();
        g = p.flags + iteration;
    }
    HIP_OK(hipMemsetAsync(flag_out, 0, sizeof(int32_t), st));
    return iteration_fused(a, p, g, g ? 1 : 0, 0, state_in_full, state_out_full, row_base, flag_out, a.k_out,
                           (float)(iteration + 1), st, adjacency_halo, agg_partial);
}

int gnn_shard_iteration_split_rows(const gnn_loop_args_t *args, const gnn_csr_t *adjacency_halo, const float *agg_partial,
                                   const float *state_in_full, float *state_out_full, int32_t row_base, const int32_t *gate,
                                   int32_t n_gate, int32_t gate_stride, int32_t *flag_out, int32_t iteration,
                                   const int32_t *node_ids, int32_t n_ids, int32_t first_chunk) {
    if (!args || !adjacency_halo || !agg_partial) return fail("args / adjacency_halo / agg_partial is NULL");
    const gnn_loop_args_t &a = *args;
    if (!state_in_full || !state_out_full || !flag_out) return fail("state buffers / flag_out are NULL");
    if (iteration < 0 || iteration >= a.max_iteration) return fail("iteration %d out of [0, max_iteration)", iteration);
    if (n_gate < 0 || (n_gate > 0 && !gate)) return fail("bad gate list");
    if (n_ids < 0 || n_ids > a.n_nodes || (n_ids > 0 && !node_ids)) return fail("bad node_ids / n_ids");
    if (a.composite) return fail("gnn_shard_iteration_split_rows: homogeneous models");
    if (!gnn_shard_can_split(args)) return fail("gnn_shard_iteration_split_rows: this model / shard does not run on the wave-specialised kernel");
    Plan p;
    TRY(make_plan(a, a.workspace, p, false));
    if (p.SP > 64) return fail("gnn_shard_iteration_split_rows: state widths up to 64");
    TRY(check_csr(*adjacency_halo, "adjacency_halo", p.N, a.adjacency.n_src));
    if (row_base < 0 || row_base + p.N > a.adjacency.n_src) return fail("row_base out of the full buffer");
    hipStream_t st = (hipStream_t)a.stream;
    const int *g = nullptr;
    if (n_gate > 0 && !(a.flags & GNN_FLAG_NO_EARLY_EXIT)) {
        if (first_chunk) {          // the gate word of this iteration: evaluated once, read by every chunk's launch
            k_or_flags<<<1, 64, 0, st>>>(gate, n_gate, gate_stride, p.flags + iteration);
            LAUNCH_OK();
        }
        g = p.flags + iteration;
    }
    if (first_chunk) HIP_OK(hipMemsetAsync(flag_out, 0, sizeof(int32_t), st));      // (later chunks OR into it)
    if (n_ids == 0) return 0;
    return iteration_fused(a, p, g, g ? 1 : 0, 0, state_in_full, state_out_full, row_base, flag_out, a.k_out,
                           (float)(iteration + 1), st, adjacency_halo, agg_partial, node_ids, n_ids);
}

int gnn_shard_output(const gnn_loop_args_t *args, const float *buf0_full, const float *buf1_full, int32_t row_base) {
    if (!args) return fail("args is NULL");
    const gnn_loop_args_t &a = *args;
    if (!buf0_full || !buf1_full) return fail("state buffers are NULL");
    Plan p;
    TRY(make_plan(a, a.workspace, p, true));
    hipStream_t st = (hipStream_t)a.stream;
    const long total = (long)p.N * p.S;
    if (total > 0) {
        gnn::k_select_state<<<std::min(cdiv(total, 256), 256 * 16), 256, 0, st>>>(
            a.k_out, buf0_full + (size_t)row_base * p.SP, buf0_full + (size_t)row_base * p.SP, buf1_full + (size_t)row_base * p.SP, p.SP,
            a.state_out, p.S, p.N, p.S);
        LAUNCH_OK();
    }
    TRY(output_stage(a, p, st));
    return can_fuse(a, p) ? launch_fold_error(p, a.k_out, st) : 0;
}

}  // extern "C"

#include "shard_loop.hpp"
#include "train_api.hpp"
#include "train_loop.hpp"
#include "train_composite.hpp"
