// gnn_train_step for HETEROGENEOUS models on LARGE graphs (reference GNN/Models/CompositeGNN.py:275-304 over the loop of :215-234; BASELINE
// C5: 500 k nodes / 5 M arcs, 3 node types): the row-streaming kernels of the homogeneous large-graph step (kernels_train_big.hpp) with one
// weight set per node type.
//
// The step runs in POSITION SPACE: position i = node type_nodes[i] of the caller's graph (the caller's per-type node lists ARE that
// permutation), so the rows of a type are one contiguous range [type_offsets[t], type_offsets[t + 1]) of every [N, S] array of the tape, and
// "network t on the rows of its type" (CompositeGNN.py:223-228: boolean_mask, :229-232: scatter_nd + reduce_sum) is the homogeneous kernel
// launched on that range with network t's weights, BatchNormalization statistics (taken over ITS rows) and gradient shares - no row-index
// indirection inside the kernels, whose loads are contiguous windows (LDS-DMA rings, buffer descriptors of the arrays' exact sizes).
// What crosses types is what crosses rows anyway: the neighbour sums.  They walk the adjacency RE-LABELLED into positions - by destination
// for the forward sums, by source for the transposed ones - built here on the device once per step (a count, a scan, a fill: the arcs of a
// row keep their order, so every sum adds the same terms in the same order as the caller's CSR would).
//
// Per executed iteration and type: forward k_aggregate_stats (the type's rows of Adj^T state + their column statistics) -> k_fold_bn ->
// k_train_fwd_b6<.., ADD> (the rows' new state, its statistics, the loop condition); backward k_train_wgrad_b6<.., XT> -> k_reduce_partials ->
// k_first_layer_param_grads -> k_train_bwd_dx_b6 -> k_aggregate_dz.  Two differences from the homogeneous step, both because a type's constant
// inputs [labels[:, :d_t] | aggregated_component] are wider than the 31 columns of its 128-byte constants line (C5: 43 / 37 / 33):
//   * forward: their share of the first layer, Cc[n, :] = b + sum_c (a_c (x_c - mean_c) + beta_c) W[c, :], does not change between the
//     iterations of a step (their batch statistics do not) - it is computed once and ADDED to the pre-activations (TrainFwdArgs::addend)
//     instead of multiplied on the matrix cores every iteration: 256 bytes a row instead of 128, one k block of products less;
//   * backward: the weight gradient contracts a line of 64 floats (k_train_wgrad_b6<.., XT = 2>; 32 when every type fits).
// The output network runs on the converged state in the caller's node order, as the general composite step does.
#pragma once
// (included by train_composite.hpp: shares the helpers of train_loop.hpp / train_composite.hpp)

namespace {

inline bool composite_big_enabled() {      // GNN_TRAIN_COMPOSITE_BIG=0: large heterogeneous graphs stay on the general kernels (one launch per layer, type and iteration)
    int v = -1;      // (read at every call: tests switch it inside one process)
    { const char *e = getenv("GNN_TRAIN_COMPOSITE_BIG"); v = (e && e[0] == '0') ? 0 : 1; }
    return v != 0;
}

// ---- the adjacency re-labelled into positions ------------------------------------------------------------------------------------------
// rp[i + 1] - rp[i] = degree of node perm[i]: an exclusive scan of the permuted degrees in chunks of 2 048 (chunk sums, one workgroup scans
// them, the chunks finish) - 3 launches, N ints read twice.
constexpr int SCAN_CHUNK = 2048;
__device__ __forceinline__ int perm_degree(const int *__restrict__ rowptr, const int *__restrict__ perm, int i, int n) {
    if (i >= n) return 0;
    const int j = perm[i];
    return rowptr[j + 1] - rowptr[j];
}
__device__ __forceinline__ int block_exclusive_scan_256(int v, int *sh /* [256] */, int *total) {
    const int tid = threadIdx.x;
    sh[tid] = v;
    __syncthreads();
    for (int off = 1; off < 256; off <<= 1) {
        const int t = tid >= off ? sh[tid - off] : 0;
        __syncthreads();
        sh[tid] += t;
        __syncthreads();
    }
    const int incl = sh[tid];
    *total = sh[255];
    __syncthreads();
    return incl - v;
}
__global__ void __launch_bounds__(256) k_permdeg_chunk_sums(const int *__restrict__ rowptr, const int *__restrict__ perm, int n, int *__restrict__ sums) {
    __shared__ int sh[256];
    const int base = blockIdx.x * SCAN_CHUNK + threadIdx.x * 8;
    int s = 0;
#pragma unroll
    for (int u = 0; u < 8; ++u) s += perm_degree(rowptr, perm, base + u, n);
    int total;
    block_exclusive_scan_256(s, sh, &total);
    if (threadIdx.x == 0) sums[blockIdx.x] = total;
}
__global__ void __launch_bounds__(256) k_scan_sums_inplace(int *__restrict__ sums, int nb) {      // one workgroup: sums -> exclusive prefix, sums[nb] = total
    __shared__ int sh[256];
    int carry = 0;
    for (int b0 = 0; b0 < nb; b0 += 256) {
        const int i = b0 + threadIdx.x;
        const int v = i < nb ? sums[i] : 0;
        int total;
        const int ex = block_exclusive_scan_256(v, sh, &total);
        if (i < nb) sums[i] = carry + ex;
        carry += total;
    }
    if (threadIdx.x == 0) sums[nb] = carry;
}
__global__ void __launch_bounds__(256) k_permdeg_scan_chunks(const int *__restrict__ rowptr, const int *__restrict__ perm, int n, const int *__restrict__ sums,
                                                             int *__restrict__ rp) {
    __shared__ int sh[256];
    const int base = blockIdx.x * SCAN_CHUNK + threadIdx.x * 8;
    int d[8], s = 0;
#pragma unroll
    for (int u = 0; u < 8; ++u) { d[u] = perm_degree(rowptr, perm, base + u, n); s += d[u]; }
    int total;
    int run = sums[blockIdx.x] + block_exclusive_scan_256(s, sh, &total);
#pragma unroll
    for (int u = 0; u < 8; ++u) { if (base + u < n) rp[base + u] = run; run += d[u]; }
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 255) rp[n] = sums[gridDim.x];
}
// the arcs of position i = the arcs of node perm[i], in their order, their other endpoint re-labelled (16 lanes a row)
__global__ void __launch_bounds__(256) k_perm_fill(const int *__restrict__ rowptr, const int *__restrict__ src, const float *__restrict__ w, const int *__restrict__ perm,
                                                   const int *__restrict__ inv, const int *__restrict__ rp, int n, int *__restrict__ src_out, float *__restrict__ w_out) {
    const int l = threadIdx.x & 15;
    for (int i = blockIdx.x * 16 + (threadIdx.x >> 4); i < n; i += gridDim.x * 16) {
        const int node = perm[i], b = rowptr[node], e = rowptr[node + 1], o = rp[i];
        for (int j = b + l; j < e; j += 16) {
            src_out[o + (j - b)] = inv[src[j]];
            if (w) w_out[o + (j - b)] = w[j];
        }
    }
}
__global__ void __launch_bounds__(256) k_gather_f32(const float *__restrict__ x, const int *__restrict__ perm, int n, float *__restrict__ out) {
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) out[i] = x[perm[i]];
}

// the loop's first dZ: out[i, :] = G[perm[i], :] (.) act'(Y[i, :]) with the activation of position i's node type (S % 4 == 0; 16-byte pieces)
struct ActRanges { int begin[GNN_MAX_TYPES + 1], act[GNN_MAX_TYPES], n; };
__global__ void __launch_bounds__(256) k_gather_rows_dz(const float *__restrict__ G, const int *__restrict__ perm, const float *__restrict__ Y, int N, int S, ActRanges ar,
                                                        float *__restrict__ out) {
    const int pr = S / 4;
    const size_t total = (size_t)N * pr;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        const int i = (int)(idx / pr), c4 = (int)(idx % pr);
        int act = ar.act[0];
#pragma unroll
        for (int t = 1; t < GNN_MAX_TYPES; ++t) if (t < ar.n && i >= ar.begin[t]) act = ar.act[t];
        const gnn::f32x4 g = *reinterpret_cast<const gnn::f32x4 *>(G + (size_t)perm[i] * S + 4 * c4), y = *reinterpret_cast<const gnn::f32x4 *>(Y + (size_t)i * S + 4 * c4);
        gnn::f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = g[e] * gnn::activate_grad_from_output(act, y[e]);
        *reinterpret_cast<gnn::f32x4 *>(out + (size_t)i * S + 4 * c4) = o;
    }
}

struct PosCsr { int *rp, *src; float *w, *row_scale; int nnz; };

// `c` (n_dst = n_src = N) re-labelled: row i = the row of node perm[i], entries inv[.]
int build_pos_csr(const gnn_csr_t &c, const int *perm, const int *inv, int N, PosCsr &o, int *scan_tmp, hipStream_t st) {
    const int nb = cdiv(N, SCAN_CHUNK);
    k_permdeg_chunk_sums<<<nb, 256, 0, st>>>(c.rowptr, perm, N, scan_tmp);
    LAUNCH_OK();
    k_scan_sums_inplace<<<1, 256, 0, st>>>(scan_tmp, nb);
    LAUNCH_OK();
    k_permdeg_scan_chunks<<<nb, 256, 0, st>>>(c.rowptr, perm, N, scan_tmp, o.rp);
    LAUNCH_OK();
    k_perm_fill<<<std::min(cdiv(N, 16), 256 * 32), 256, 0, st>>>(c.rowptr, c.src, c.w, perm, inv, o.rp, N, o.src, o.w);
    LAUNCH_OK();
    if (c.row_scale) { k_gather_f32<<<std::min(cdiv(N, 256), 4096), 256, 0, st>>>(c.row_scale, perm, N, o.row_scale); LAUNCH_OK(); }
    o.nnz = c.nnz;
    return 0;
}

// the constants line of the rows of one type, in POSITION order: Xc[m] = [segment columns of node rows[m] .. | 1 | 0 ..]  (XW floats)
template <int XW>
__global__ void __launch_bounds__(256) k_pack_xc_pos(int count, const int *__restrict__ rows, gnn::PackSegs ps, float *__restrict__ Xc) {
    const int Kc = ps.width[0] + ps.width[1] + ps.width[2];
    const size_t total = (size_t)count * XW;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t m = i / XW;
        const size_t j = (size_t)rows[m];
        const int c = (int)(i % XW);
        float v = 0.0f;
        if (c < ps.width[0]) v = ps.ptr[0][j * ps.ld[0] + c];
        else if (c < ps.width[0] + ps.width[1]) v = ps.ptr[1][j * ps.ld[1] + (c - ps.width[0])];
        else if (c < Kc) v = ps.ptr[2][j * ps.ld[2] + (c - ps.width[0] - ps.width[1])];
        else if (c == Kc) v = 1.0f;
        Xc[i] = v;
    }
}

// ---- launchers of the kernel forms only this path uses ---------------------------------------------------------------------------------
// (the dynamic-LDS limit of a kernel is raised once per device and kernel)
int raise_dynamic_lds(const void *fn, size_t lds) {
    static std::vector<std::pair<int, const void *>> done;
    static std::mutex mtx;
    int dev = 0;
    HIP_OK(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lock(mtx);
    const std::pair<int, const void *> key(dev, fn);
    if (std::find(done.begin(), done.end(), key) != done.end()) return 0;
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return fail("cannot raise the dynamic LDS limit to %zu bytes", lds);
    done.push_back(key);
    return 0;
}

template <int SQ, int ACT>
int launch_train_fwd_b6_add_sa(const gnn::TrainFwdArgs &fa, int grid, hipStream_t st) {
    const size_t lds = gnn::train_fwd_b6_lds<SQ, true>();
    TRY(raise_dynamic_lds((const void *)gnn::k_train_fwd_b6<SQ, ACT, true>, lds));
    gnn::k_train_fwd_b6<SQ, ACT, true><<<grid, 64 * gnn::TB_WAVES, lds, st>>>(fa);
    return hipGetLastError() == hipSuccess ? 0 : fail("k_train_fwd_b6<ADD> launch failed");
}
template <int SQ>
int launch_train_fwd_b6_add_s(const gnn::TrainFwdArgs &fa, int grid, hipStream_t st) {
    switch (fa.act) {
        case GNN_ACT_LINEAR: return launch_train_fwd_b6_add_sa<SQ, GNN_ACT_LINEAR>(fa, grid, st);
        case GNN_ACT_RELU: return launch_train_fwd_b6_add_sa<SQ, GNN_ACT_RELU>(fa, grid, st);
        case GNN_ACT_SELU: return launch_train_fwd_b6_add_sa<SQ, GNN_ACT_SELU>(fa, grid, st);
        case GNN_ACT_TANH: return launch_train_fwd_b6_add_sa<SQ, GNN_ACT_TANH>(fa, grid, st);
        case GNN_ACT_SIGMOID: return launch_train_fwd_b6_add_sa<SQ, GNN_ACT_SIGMOID>(fa, grid, st);
        case GNN_ACT_ELU: return launch_train_fwd_b6_add_sa<SQ, GNN_ACT_ELU>(fa, grid, st);
        case GNN_ACT_SOFTPLUS: return launch_train_fwd_b6_add_sa<SQ, GNN_ACT_SOFTPLUS>(fa, grid, st);
        default: return fail("k_train_fwd_b6<ADD>: no instance for activation %d", fa.act);
    }
}
int launch_train_fwd_add(const gnn::TrainFwdArgs &fa, int S, hipStream_t st, int *grid_out) {
    const int n_tiles16 = (fa.M + 15) / 16;
    const int grid = std::max(1, std::min(std::min(device_cus(), BIG_FWD_BLOCKS), cdiv(n_tiles16, gnn::TB_WAVES)));     // one 8-wave workgroup per CU
    *grid_out = grid;
    return S == 64 ? launch_train_fwd_b6_add_s<4>(fa, grid, st) : launch_train_fwd_b6_add_s<2>(fa, grid, st);
}

// the weight gradient over [state | agg | constants line of 32 XT floats] of rows that carry dZ (the LINEAR instance)
template <int NB, int XT>
int launch_train_wgrad_b6_xt_nb(const gnn::TrainWgradArgs &wa, int grid, hipStream_t st) {
    const size_t lds = gnn::train_wgrad_b6_lds<NB, GNN_ACT_LINEAR, XT>();
    TRY(raise_dynamic_lds((const void *)gnn::k_train_wgrad_b6<NB, GNN_ACT_LINEAR, XT>, lds));
    gnn::k_train_wgrad_b6<NB, GNN_ACT_LINEAR, XT><<<grid, 256, lds, st>>>(wa);
    return hipGetLastError() == hipSuccess ? 0 : fail("k_train_wgrad_b6 launch failed");
}
int launch_train_wgrad_xt(const gnn::TrainWgradArgs &wa, int S, int XT, int grid, hipStream_t st) {
    if (wa.Y || wa.act != GNN_ACT_LINEAR) return fail("k_train_wgrad_b6<XT>: the dZ form only");
    if (S == 64) return XT == 2 ? launch_train_wgrad_b6_xt_nb<2, 2>(wa, grid, st) : launch_train_wgrad_b6_xt_nb<2, 1>(wa, grid, st);
    return XT == 2 ? launch_train_wgrad_b6_xt_nb<1, 2>(wa, grid, st) : launch_train_wgrad_b6_xt_nb<1, 1>(wa, grid, st);
}

// ---- one launch for the rows of every node type (gnn::TypeLaunch) ---------------------------------------------------------------------------
// workgroups of a launch of `total` dealt to the types in proportion to their rows (>= 1 each where there are rows, never more than `total` together:
// a second round of workgroups on a CU would double the launch's time)
void split_blocks(int total, const int *count, int T, int *out) {
    long N = 0; int live = 0;
    for (int t = 0; t < T; ++t) { N += count[t]; live += count[t] > 0; }
    int used = 0, big = 0;
    for (int t = 0; t < T; ++t) {
        out[t] = count[t] > 0 ? std::max(1, (int)((long)std::max(total - live, 0) * count[t] / std::max<long>(N, 1))) : 0;
        used += out[t];
        if (count[t] > count[big]) big = t;
    }
    if (used < total && N > 0) out[big] += total - used;
}

template <int SQ, int ACT>
int launch_train_fwd_types_sa(const gnn::TypeLaunch<gnn::TrainFwdArgs> &m, hipStream_t st) {
    const size_t lds = gnn::train_fwd_b6_lds<SQ, true>();
    TRY(raise_dynamic_lds((const void *)gnn::k_train_fwd_b6_types<SQ, ACT, true>, lds));
    gnn::k_train_fwd_b6_types<SQ, ACT, true><<<m.blk_begin[m.n], 64 * gnn::TB_WAVES, lds, st>>>(m);
    return hipGetLastError() == hipSuccess ? 0 : fail("k_train_fwd_b6_types launch failed");
}
template <int SQ>
int launch_train_fwd_types_s(const gnn::TypeLaunch<gnn::TrainFwdArgs> &m, int act, hipStream_t st) {
    switch (act) {
        case GNN_ACT_LINEAR: return launch_train_fwd_types_sa<SQ, GNN_ACT_LINEAR>(m, st);
        case GNN_ACT_RELU: return launch_train_fwd_types_sa<SQ, GNN_ACT_RELU>(m, st);
        case GNN_ACT_SELU: return launch_train_fwd_types_sa<SQ, GNN_ACT_SELU>(m, st);
        case GNN_ACT_TANH: return launch_train_fwd_types_sa<SQ, GNN_ACT_TANH>(m, st);
        case GNN_ACT_SIGMOID: return launch_train_fwd_types_sa<SQ, GNN_ACT_SIGMOID>(m, st);
        case GNN_ACT_ELU: return launch_train_fwd_types_sa<SQ, GNN_ACT_ELU>(m, st);
        case GNN_ACT_SOFTPLUS: return launch_train_fwd_types_sa<SQ, GNN_ACT_SOFTPLUS>(m, st);
        default: return fail("k_train_fwd_b6_types: no instance for activation %d", act);
    }
}
int launch_train_fwd_types(const gnn::TypeLaunch<gnn::TrainFwdArgs> &m, int act, int S, hipStream_t st) {
    return S == 64 ? launch_train_fwd_types_s<4>(m, act, st) : launch_train_fwd_types_s<2>(m, act, st);
}

template <int NB, int XT>
int launch_train_wgrad_types_nb(const gnn::TypeLaunch<gnn::TrainWgradArgs> &m, hipStream_t st) {
    const size_t lds = gnn::train_wgrad_b6_lds<NB, GNN_ACT_LINEAR, XT>();
    TRY(raise_dynamic_lds((const void *)gnn::k_train_wgrad_b6_types<NB, GNN_ACT_LINEAR, XT>, lds));
    gnn::k_train_wgrad_b6_types<NB, GNN_ACT_LINEAR, XT><<<m.blk_begin[m.n], 256, lds, st>>>(m);
    return hipGetLastError() == hipSuccess ? 0 : fail("k_train_wgrad_b6_types launch failed");
}
int launch_train_wgrad_types(const gnn::TypeLaunch<gnn::TrainWgradArgs> &m, int S, int XT, hipStream_t st) {
    if (S == 64) return XT == 2 ? launch_train_wgrad_types_nb<2, 2>(m, st) : launch_train_wgrad_types_nb<2, 1>(m, st);
    return XT == 2 ? launch_train_wgrad_types_nb<1, 2>(m, st) : launch_train_wgrad_types_nb<1, 1>(m, st);
}

template <int HQ>
int launch_train_bwd_types_h(const gnn::TypeLaunch<gnn::TrainBwdArgs> &m, hipStream_t st) {
    const size_t lds = gnn::train_bwd_b6_lds<HQ>();
    TRY(raise_dynamic_lds((const void *)gnn::k_train_bwd_dx_b6_types<HQ, GNN_ACT_LINEAR>, lds));
    gnn::k_train_bwd_dx_b6_types<HQ, GNN_ACT_LINEAR><<<m.blk_begin[m.n], 256, lds, st>>>(m);
    return hipGetLastError() == hipSuccess ? 0 : fail("k_train_bwd_dx_b6_types launch failed");
}
int launch_train_bwd_types(const gnn::TypeLaunch<gnn::TrainBwdArgs> &m, int S, hipStream_t st) {      // (the dZ form: Y == NULL in every entry)
    return S == 64 ? launch_train_bwd_types_h<4>(m, st) : launch_train_bwd_types_h<2>(m, st);
}

// ---- the plan's large-graph part -----------------------------------------------------------------------------------------------------------
struct CBig {
    int XT, XW;                          // 32-column tiles / floats of a constants line
    int *scan_tmp;
    PosCsr d, s;                         // the adjacency in positions: by destination, by source
    float *xc, *Cc, *Gpos;
    float *part_a[GNN_MAX_TYPES], *part_y[GNN_MAX_TYPES], *part_w[GNN_MAX_TYPES];      // every type its own partials: one k_stats_finish / k_reduce_partials for all
    int Kc[GNN_MAX_TYPES]; gnn::ConstCols cc[GNN_MAX_TYPES];
    int fwd_blocks[GNN_MAX_TYPES], bwd_blocks[GNN_MAX_TYPES], wgrad_blocks[GNN_MAX_TYPES];   // workgroups of the merged launches
    bool one_act; int act;               // every type with rows has the same activation: ONE forward launch an iteration
};

// whether the large-graph kernels take this step (decided before the tape is carved)
bool composite_big_applies(const gnn_train_args_t &ta, int N, int S, int W_comp, int *XT_out) {
    const gnn_loop_args_t &a = ta.loop;
    if (!composite_big_enabled() || N < train_big_min_nodes() || (S != 32 && S != 64)) return false;
    if (!train_bf16x6_enabled() || !train_wgrad_b6_enabled() || !train_dz_enabled() || !train_wgrad_enabled()) return false;
    if ((size_t)N * 2 * S * 4 >= 0xFFFFFFF0ull || (size_t)a.adjacency.nnz * 4 >= 0xFFFFFFF0ull) return false;       // (4 GiB buffer windows)
    int XT = 1;
    for (int t = 0; t < a.n_types; ++t) {
        const gnn_mlp_t &m = a.net_state[t];
        if (m.n_layers != 1 || m.units[0] != S || m.activation[0] == GNN_ACT_SOFTMAX) return false;
        const int kc = a.type_dim_label[t] + W_comp;
        if (kc > 63) return false;
        if (kc > 31) XT = 2;
    }
    *XT_out = XT;
    return true;
}

}  // namespace
