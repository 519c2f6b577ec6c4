/* gnnloop.h — C ABI of libgnnloop.so: the MI355X (gfx950) implementation of GNNkeras' convergent message-passing loop.
 *
 * Drop-in boundary.  The reference has no FFI: its seam is the Python method
 * `GNNnodeBased.Loop` (reference GNN/Models/GNN.py:245-274) and the TensorFlow ops it dispatches.  Each entry point
 * below replaces one of those call sites and is what a binding for this path binds (ctypes stub: INTEGRATION.md;
 * in-tree binding: gnnkeras_amd/_native.py).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer (HBM of the current HIP device) unless marked HOST;
 *   - inputs are borrowed and never written; outputs are caller-allocated;
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); nothing in this library synchronises the
 *     host with the device: the loop runs `max_iteration` gated launches and the iteration count comes back as a
 *     device scalar;
 *   - all arithmetic is float32 (tf.keras.backend.floatx(), reference graph_class.py:43); ids are int32;
 *   - return value: 0 = ok, non-zero = error, message via gnn_last_error() (thread-local). Never aborts.
 */
#ifndef GNNLOOP_H
#define GNNLOOP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GNN_ABI_VERSION 9

/* Keras activation names accepted by the reference MLP builder (GNN/Models/MLP.py:16). */
enum gnn_activation {
    GNN_ACT_LINEAR = 0, GNN_ACT_RELU = 1, GNN_ACT_SELU = 2, GNN_ACT_TANH = 3, GNN_ACT_SIGMOID = 4,
    GNN_ACT_ELU = 5, GNN_ACT_SOFTPLUS = 6, GNN_ACT_SOFTMAX = 7
};

/* problem focus: GNNnodeBased / GNNarcBased / GNNgraphBased (GNN.py:8, :312, :336). */
enum gnn_focus { GNN_FOCUS_NODE = 0, GNN_FOCUS_ARC = 1, GNN_FOCUS_GRAPH = 2 };

enum gnn_flags {
    GNN_FLAG_UNFUSED = 1,      /* run the iteration as separate aggregate / dense / predicate kernels            */
    GNN_FLAG_NO_EARLY_EXIT = 2,/* debugging: ignore the convergence predicate (always max_iteration iterations)  */
    /* testing / tuning: pin the generation of the fused iteration kernel instead of the size-based choice
     * (2 = phase-alternating, 4 = wave-specialised, 5 = whole loop in one persistent launch, small graphs only, 6 = the same
     * with several tiles per workgroup, 7 = convergence groups with one CU per group and its state in LDS - each falls back to
     * the size-based choice when it does not apply).  Results are the same within float32
     * summation order; the GNN_FUSED_KERNEL environment variable has the same effect process-wide. */
    GNN_FLAG_FUSED_GEN2 = 2 << 4, GNN_FLAG_FUSED_GEN4 = 4 << 4,
    GNN_FLAG_FUSED_GEN5 = 5 << 4, GNN_FLAG_FUSED_GEN6 = 6 << 4, GNN_FLAG_FUSED_GEN7 = 7 << 4, GNN_FLAG_FUSED_GEN_MASK = 7 << 4
};

/* A sparse operator A (n_src x n_dst, COO in the reference: tf.SparseTensor) stored as the CSR of its transpose:
 * the form in which `tf.sparse.sparse_dense_matmul(A, X, adjoint_a=True)` (GNN.py:228,254,258,345) walks it.
 * Entry values are either per entry (`w`), or one value per destination (`row_scale`), or all 1 (both NULL). */
typedef struct gnn_csr {
    int32_t n_dst;            /* columns of A = rows of the result                                              */
    int32_t n_src;            /* rows of A    = rows of the dense operand                                       */
    int32_t nnz;
    const int32_t *rowptr;    /* [n_dst + 1]                                                                    */
    const int32_t *src;       /* [nnz] row of A (source node / arc id), ascending inside a destination          */
    const float *w;           /* [nnz] or NULL                                                                  */
    const float *row_scale;   /* [n_dst] or NULL                                                                */
} gnn_csr_t;

/* Keras Sequential built by the reference MLP() (MLP.py:12-78): optional BatchNormalization first, then Dense x n.
 * Arrays are in Keras get_weights() order: BN -> gamma, beta, moving_mean, moving_variance; Dense -> kernel
 * (row-major [in][out]), bias [out]. */
#define GNN_MAX_LAYERS 8
typedef struct gnn_mlp {
    int32_t in_dim;
    int32_t n_layers;
    int32_t units[GNN_MAX_LAYERS];
    int32_t activation[GNN_MAX_LAYERS];
    const float *kernel[GNN_MAX_LAYERS];
    const float *bias[GNN_MAX_LAYERS];
    int32_t has_bn;
    float bn_eps;
    const float *bn_gamma, *bn_beta, *bn_mean, *bn_var;   /* [in_dim] each */
} gnn_mlp_t;

#define GNN_MAX_TYPES 8
#define GNN_MAX_PEERS 7       /* the other GPUs of a fully connected 8-GPU node                                   */
/* Whole forward pass of one (merged) graph.
 * Homogeneous: replaces GNNnodeBased.Loop (GNN.py:245-274), GNNarcBased.apply_filters (:317-330),
 *              GNNgraphBased.Loop (:341-346).            n_types = 1, type_* unused.
 * Composite:   replaces CompositeGNNnodeBased.Loop (CompositeGNN.py:242-272), arc (:315-327), graph (:338-343).
 *              n_types = T > 1 (or composite != 0). */
typedef struct gnn_loop_args {
    int32_t abi_version;      /* GNN_ABI_VERSION                                                               */
    int32_t composite;        /* 0 homogeneous input layout, 1 composite input layout                           */
    /* graph --------------------------------------------------------------------------------------------------- */
    int32_t n_nodes, n_arcs;
    int32_t dim_node_label;   /* width of `nodes` (homogeneous: L; composite: max_t d_t)                        */
    int32_t dim_arc_label;    /* A                                                                              */
    const float *nodes;       /* [n_nodes, ld_nodes]                                                            */
    int32_t ld_nodes;
    const float *arc_labels;  /* &arcs[0][2], row stride ld_arcs (= 2 + A for the reference arcs matrix)        */
    int32_t ld_arcs;
    gnn_csr_t adjacency;      /* n_src = n_dst = n_nodes                                                        */
    gnn_csr_t arcnode;        /* n_src = n_arcs, n_dst = n_nodes                                                */
    /* composite only ------------------------------------------------------------------------------------------ */
    int32_t n_types;
    int32_t type_dim_label[GNN_MAX_TYPES];   /* d_t                                                              */
    const int32_t *type_nodes;               /* node ids grouped by type, ascending inside a type [n_nodes]      */
    int32_t type_offsets[GNN_MAX_TYPES + 1]; /* HOST values: type t owns type_nodes[off[t] : off[t+1]]           */
    gnn_csr_t composite_adjacency[GNN_MAX_TYPES];
    /* networks ------------------------------------------------------------------------------------------------ */
    gnn_mlp_t net_state[GNN_MAX_TYPES];      /* [0] for homogeneous                                              */
    gnn_mlp_t net_output;
    int32_t state_dim;        /* state_vect_dim d >= 0; 0 => state0 = nodes (GNN.py:259)                        */
    int32_t max_iteration;
    float state_threshold;
    const float *state0;      /* [n_nodes, state_dim] row-major, required when state_dim > 0 (replaces the
                                 tf.random.normal draw of GNN.py:257)                               */
    /* output stage --------------------------------------------------------------------------------------------- */
    int32_t focus;            /* enum gnn_focus                                                                 */
    int32_t n_out;            /* rows that pass `set_mask & output_mask` (GNN.py:269)                           */
    const int32_t *out_index; /* [n_out] ascending node ids (node/graph focus) or arc ids (arc focus)           */
    const int32_t *arc_src;   /* [n_arcs] arc focus: adjacency.indices[:,0] in arcs order                       */
    const int32_t *arc_dst;   /* [n_arcs] arc focus: adjacency.indices[:,1]                                     */
    gnn_csr_t nodegraph;      /* graph focus: n_src = n_out, n_dst = #graphs                                    */
    /* results -------------------------------------------------------------------------------------------------- */
    float *k_out;             /* [1] iterations executed, float like the reference; NEGATIVE when a bounded
                                 in-launch wait expired: the persistent whole-loop launch could not get all its
                                 workgroups resident (grid barrier timed out, e.g. too many such loops overlapped on
                                 other streams), or a slot hand-off of the wave-specialised kernel was lost: results
                                 are then invalid - rerun with GNN_FLAG_FUSED_GEN2                                 */
    float *state_out;         /* [n_nodes, S] S = state_dim, or dim of the state when state_dim == 0            */
    float *out;               /* [n_out, T] (node / arc focus) or [#graphs, T] (graph focus)                    */
    /* execution ------------------------------------------------------------------------------------------------ */
    void *workspace;          /* >= gnn_loop_workspace_bytes(args) bytes, 256-byte aligned                      */
    size_t workspace_bytes;
    void *stream;
    int32_t flags;            /* enum gnn_flags                                                                 */
    /* node-range shard of a larger graph (multi-GPU; all zero / NULL on one GPU) -------------------------------------
     * The shard owns n_nodes destination nodes; adjacency.n_src rows of the exchanged full state buffer are visible.
     * `nodes` holds the OWN labels [n_nodes, L]; `nodes_src` the labels of every source row [adjacency.n_src, L] in
     * the row order of the full state buffer (the one-off halo of the label aggregate, GNN.py:258). */
    const float *nodes_src;
    int32_t ld_nodes_src;
    /* hub nodes (optional; n_heavy_segments == 0 = absent) ----------------------------------------------------------
     * Destination rows with very many in-arcs are cut into segments [heavy_seg_beg[s], heavy_seg_end[s]) of the arcs of
     * `adjacency` (its src / w arrays); before every iteration a pre-pass sums each segment with a whole workgroup into
     * the virtual state row n_nodes + s, and the iterations walk `adjacency_light`, in which a hub row lists its virtual
     * rows (n_src = n_nodes + n_heavy_segments) instead of its arcs.
     * A binder should split every row above ~512 in-arcs (gnnkeras_amd/sparse.py: split_heavy does): besides the
     * load balance, the fused kernels' in-launch hand-offs poll with a bound (~0.4 s) sized for rows of that order. */
    gnn_csr_t adjacency_light;
    const int32_t *heavy_seg_beg, *heavy_seg_end;
    int32_t n_heavy_segments;
    /* measurement (optional) ------------------------------------------------------------------------------------ */
    void *ev_loop_begin;      /* hipEvent_t or NULL: recorded on `stream` right before the first iteration launch */
    void *ev_loop_end;        /* hipEvent_t or NULL: recorded right after the last iteration launch               */
    /* independent convergence groups (optional; n_groups == 0 = one loop over the whole graph) ----------------------
     * The graph is the merge of n_groups batches (block-diagonal operators, nodes of group g contiguous in
     * [group_node_begin[g], group_node_begin[g + 1])), and the reference would have run `Loop` once per batch: its
     * `while` (GNN.py:265) stops a batch when no node OF THAT BATCH moves.  With groups, one call runs all those loops at
     * once - each group has its own predicate, its own iteration counter (k_out is [n_groups]) and stops on its own -
     * so many small batches fill the GPU in one launch instead of one underfilled launch each.  Results per group equal
     * a separate call on that batch alone.  Supported where the whole-loop kernel applies (gnn_loop_groups_supported);
     * everything before and after the loop (constants, output network, pooling) is per node / per graph anyway. */
    const int32_t *group_node_begin;   /* HOST array [n_groups + 1], ascending, [0] = 0, [n_groups] = n_nodes         */
    int32_t n_groups;                  /* see gnn_loop_groups_supported                                                */
    /* group SETS (optional, ABI 4; n_group_sets == 0 = every group is its own set).  A batch whose state does not fit the LDS of
     * one CU is cut - along graph boundaries, the merged batch is block-diagonal (graph_class.py:399-408) - into several
     * groups that run on one CU each and share nothing but the `reduce_any` of the reference's condition (GNN.py:212): the
     * groups of set s, [group_set_begin[s], group_set_begin[s + 1]), exchange one flag word per iteration and leave the loop
     * together, after the same k.  k_out stays [n_groups] (the groups of a set report the same value).  Only the
     * one-CU-per-group form (gnn_loop_groups_supported() == 2) takes sets, and only when all groups are resident at once
     * (n_groups <= CUs) whenever some set has more than one group. */
    const int32_t *group_set_begin;    /* HOST array [n_group_sets + 1], ascending, [0] = 0, [n_group_sets] = n_groups    */
    int32_t n_group_sets;
} gnn_loop_args_t;
#define GNN_MAX_GROUPS 32                 /* groups of a call that spreads every group over several CUs                  */
#define GNN_MAX_GROUPS_RESIDENT (1 << 20) /* groups of a call that keeps every group's state in the LDS of one CU        */

const char *gnn_last_error(void);
/* Name (with template arguments) of the state-transition kernel this thread launched last, e.g.
 * "k_state_fused4<64,false,4,4,false>" - what a rocprofv3 kernel trace of the same call shows; "" before any launch. */
const char *gnn_last_kernel_name(void);
int gnn_abi_version(void);
/* sizeof() of the ABI structs as this library was compiled (0 gnn_csr_t, 1 gnn_mlp_t, 2 gnn_loop_args_t; 3 = offsetof
 * (gnn_loop_args_t, flags); 4 gnn_train_args_t; 5 = offsetof(gnn_train_args_t, tape)): lets a foreign-language binding verify
 * its struct layout at load time. */
size_t gnn_struct_size(int which);

/* bytes of scratch HBM gnn_loop_forward needs for `args` (only sizes / dims of `args` are read). */
size_t gnn_loop_workspace_bytes(const gnn_loop_args_t *args);

/* (k, state, out) = Loop(...)  — see gnn_loop_args. */
int gnn_loop_forward(const gnn_loop_args_t *args);
/* Non-zero when gnn_loop_forward accepts these args with n_groups > 0, else 0 (the caller then runs one call per batch).
 *   2: every group's state fits the LDS of one CU (nodes_g * padded width * 4 <= 156 KB, width <= 32, one-layer state network):
 *      one workgroup per group, any number of groups up to GNN_MAX_GROUPS_RESIDENT - the more the better, 256 run at once;
 *   1: homogeneous model, one- or two-layer state network of width <= 64, at most GNN_MAX_GROUPS groups whose 64-node tiles are
 *      all resident at once: sum_g ceil(nodes_g / 64) <= CUs.
 * Reads dims, flags and the host group array only. */
int gnn_loop_groups_supported(const gnn_loop_args_t *args);

/* out[j, 0:F] = sum_{e in row j} w_e * X[src_e, 0:F]   == tf.sparse.sparse_dense_matmul(A, X, adjoint_a=True)
 * (ArcNode scatter-add GNN.py:254, label aggregate :258, state aggregate :228, graph pooling :345). */
int gnn_aggregate(const gnn_csr_t *csr, const float *X, int32_t ldx, int32_t F, float *out, int32_t ldo, void *stream);

/* Y[M, units_last] = Sequential(X[M, in_dim])  — Keras inference call of a reference MLP (GNN.py:234, :273).
 * workspace: >= gnn_mlp_workspace_bytes(mlp, M). */
size_t gnn_mlp_workspace_bytes(const gnn_mlp_t *mlp, int32_t M);
int gnn_mlp_forward(const gnn_mlp_t *mlp, const float *X, int32_t ldx, int32_t M, float *Y, int32_t ldy,
                    void *workspace, size_t workspace_bytes, void *stream);

/* *flag = any_j( ||state_j - old_j||_2 > threshold * ||old_j||_2 )   (GNN.py:196-212; `old` NULL = all ones,
 * the state_old of iteration 0, GNN.py:261).  flag is an int32 device word, overwritten with 0 / 1. */
int gnn_converged(const float *state, const float *state_old, int32_t n, int32_t dim, int32_t ld, float threshold,
                  int32_t *flag, void *stream);

/* One state-transition step: GNNnodeBased.convergence (GNN.py:217-236)
 *   state_new = net_state([state | nodes (if state_dim>0) | A^T state | agg_nodes | agg_arcs])
 * or, with args->composite, CompositeGNNnodeBased.convergence (CompositeGNN.py:215-234): per node type t
 *   state_new[type t rows] = net_state[t]([nodes[:, :d_t] | state | A^T state | agg_nodes_0.. | agg_arcs][type t rows])
 * and *flag_out = predicate(state_new, state) for the next iteration.  Mostly for tests / LGNN-style callers;
 * gnn_loop_forward does the same with the constants folded once.  `args` supplies graph, nets and workspace;
 * state_in/state_out are [n_nodes, S] row-major. */
int gnn_state_step(const gnn_loop_args_t *args, const float *state_in, float *state_out, int32_t *flag_out);

/* The same step with the iteration constants HANDED IN, i.e. the reference's own argument list: `convergence(k, state, state_old,
 * nodes, adjacency, aggregated_nodes, aggregated_arcs, training)` (GNN.py:217) receives the label / arc aggregates that `Loop`
 * formed once (GNN.py:254-258) and threads them through `tf.while_loop` (:265); the composite form receives them as the column
 * blocks of one `aggregated_component` = [aggregated_nodes_0 | .. | aggregated_nodes_{T-1} | aggregated_arcs] (CompositeGNN.py:214,
 * :251-253).  aggregated_nodes: [n_nodes, ld] with dim_node_label columns (homogeneous, state_dim > 0; unused when state_dim == 0)
 * or sum_t type_dim_label[t] columns (composite); aggregated_arcs: [n_nodes, ld] with dim_arc_label columns.  args->arcnode,
 * args->arc_labels and args->composite_adjacency are not read (n_arcs may be 0); everything else as gnn_state_step. */
int gnn_state_step_agg(const gnn_loop_args_t *args, const float *state_in, const float *aggregated_nodes, int32_t ld_aggregated_nodes,
                       const float *aggregated_arcs, int32_t ld_aggregated_arcs, float *state_out, int32_t *flag_out);

/* Testing aid, not part of the path: occupies the GPU the way a co-tenant would - `n_workgroups` workgroups of one wave holding
 * `lds_bytes` of LDS each (163 840 = a whole CU) for `milliseconds` (<= 10 000) of wall-clock time on `stream`.  The whole-loop
 * kernels wait for each other inside a launch; their waits are bounded by GNN_WAIT_MS (environment, default 2 000; 0 = expire at
 * once) and an expired wait comes back as k < 0 (gnn_loop_args_t::k_out) - the recovery tests drive both with this. */
int gnn_debug_occupy(int32_t n_workgroups, int32_t lds_bytes, int32_t milliseconds, void *stream);
/* ... until the DEVICE word *release_flag becomes non-zero (the test writes it from another stream when ITS condition holds - a co-tenant
 * that leaves on a handshake, not on a clock), at the latest after max_milliseconds (<= 20 000: the kernel cannot hang the GPU). */
int gnn_debug_occupy_until(int32_t n_workgroups, int32_t lds_bytes, int32_t max_milliseconds, const int32_t *release_flag, void *stream);
/* One pinned HOST word mapped into the device (zeroed by the call): *host_out for the test to write with a plain store, *dev_out for
 * gnn_debug_occupy_until's release_flag - a release that needs no launch, so it cannot queue behind the co-tenant it is meant to release
 * (streams share a few hardware queues). */
int gnn_debug_host_flag(int32_t **host_out, int32_t **dev_out);
/* The library's expiry beacon: a DEVICE word the whole-loop / persistent kernels set when one of their bounded waits runs out (never cleared
 * by a kernel; `reset` = 1 zeroes it on `stream`, 2 raises it by hand, 0 leaves it).  Handed to gnn_debug_occupy_until as the release flag it makes a co-tenant that leaves
 * exactly when the first wait of the launch under test has expired - a device-side handshake, no clock on the host. */
int gnn_debug_expiry_beacon(int32_t **beacon_out, int32_t reset, void *stream);

/* ---- node-range sharded loop -------------------------------------------------------------------------
 * One process per GPU; rank r owns a contiguous node range and, per iteration, (1) runs gnn_shard_iteration on its
 * rows reading the full (all-gathered) state buffer and writing its own slice of the other full buffer, (2) the host
 * all-gathers the slices (RCCL, torch.distributed).  The convergence flag of a slice travels inside it (one trailing
 * flag row per slice), so there is exactly one exchange per iteration and still no host synchronisation.
 * Full state buffers: [adjacency.n_src, gnn_state_ld(S)] float32, rows of rank r at row_base_r .. */
int32_t gnn_state_ld(int32_t state_width);          /* padded row length (floats) of exchanged state buffers */
/* dst[m, :width] = src[idx[m], :width]: packs the state rows a peer needs (its halo) into a contiguous send buffer */
int gnn_gather_rows(const float *src, int32_t ld_src, const int32_t *idx, int32_t M, int32_t width, float *dst,
                    int32_t ld_dst, void *stream);
int gnn_shard_setup(const gnn_loop_args_t *args);   /* once per forward: BN folding, label / arc aggregates, C */
/* gates: run iff OR_i gate[i * gate_stride] != 0 (the flag words of every slice of state_in_full), i < n_gate.
 * flag_out: this shard's flag word inside state_out_full (zeroed, then raised if any own node still moves). */
int gnn_shard_iteration(const gnn_loop_args_t *args, const float *state_in_full, float *state_out_full,
                        int32_t row_base, const int32_t *gate, int32_t n_gate, int32_t gate_stride, int32_t *flag_out,
                        int32_t iteration);
/* picks the buffer holding the state after k iterations (k read on the device from args->k_out), copies the own rows
 * to args->state_out [n_nodes, S] and runs the output network on them into args->out. */
int gnn_shard_output(const gnn_loop_args_t *args, const float *buf0_full, const float *buf1_full, int32_t row_base);
/* Overlap of the exchange with own-range work.  The shard's adjacency is split by where a source row lives:
 * `adjacency_own` (sources in this rank's own range: final as soon as the rank's own kernel has written them) and
 * `adjacency_halo` (sources received from peers); both are CSRs over the same n_nodes destinations and index rows of the
 * full state buffer.  While the exchange of iteration i is in flight the caller runs gnn_shard_partial (un-scaled /
 * per-arc-weighted partial sums of the own-range arcs into agg_partial [n_nodes, gnn_state_ld(S)]); once it has landed,
 * gnn_shard_iteration_split walks only the halo arcs, starting every row's sum from agg_partial (summation order: own-range
 * arcs, then halo arcs - within float32 re-association of the single-GPU result).  gnn_shard_can_split tells whether this
 * model / shard runs on a kernel that supports it (one-layer state networks, state width 17..128, no hub rows); when it
 * returns 0 use gnn_shard_iteration. */
int gnn_shard_can_split(const gnn_loop_args_t *args);
int gnn_shard_partial(const gnn_loop_args_t *args, const gnn_csr_t *adjacency_own, const float *state_in_full,
                      float *agg_partial);
int gnn_shard_iteration_split(const gnn_loop_args_t *args, const gnn_csr_t *adjacency_halo, const float *agg_partial,
                              const float *state_in_full, float *state_out_full, int32_t row_base, const int32_t *gate,
                              int32_t n_gate, int32_t gate_stride, int32_t *flag_out, int32_t iteration);
/* The same second phase for a SUB-RANGE of the shard's rows (`node_ids`: device array of n_ids ascending LOCAL node ids, e.g. a slice
 * of an iota array): a rank cuts its rows into chunks, launches the halo kernel chunk by chunk and starts sending a chunk's rows to
 * its peers as soon as that chunk is written, so the exchange runs under the remaining chunks instead of behind the whole kernel
 * (gnnkeras_amd/distributed.py: pipeline_chunks).  `first_chunk` != 0: this call evaluates the iteration's gate and clears flag_out
 * (every chunk ORs its "some node still moves" into it).  Per row the arithmetic is that of gnn_shard_iteration_split: the chunks
 * together give the same bits.  Homogeneous models, state widths up to 64. */
int gnn_shard_iteration_split_rows(const gnn_loop_args_t *args, const gnn_csr_t *adjacency_halo, const float *agg_partial,
                                   const float *state_in_full, float *state_out_full, int32_t row_base, const int32_t *gate,
                                   int32_t n_gate, int32_t gate_stride, int32_t *flag_out, int32_t iteration,
                                   const int32_t *node_ids, int32_t n_ids, int32_t first_chunk);

/* ---- the exchange inside the iteration kernel (ABI 8; SURVEY 8e: "each rank writes its slice to all 7 peers, 1 hop") ---------------------
 * Instead of an all-gather BEHIND the kernel, the kernel's epilogue stores every new state row to the rank's own full buffer and, at the
 * same offset, to the full buffers of the other ranks, mapped into this process with gnn_ipc_open (hipIpcOpenMemHandle; the handles travel
 * over any host channel).  Ordering between ranks is a monotonically increasing ARRIVAL word per rank in every rank's `arrive` array:
 *     gnn_peer_wait(value i)      - before iteration i may write: every peer has published i, i.e. has finished iteration i - 1 (its rows
 *                                   of the buffer iteration i reads have landed here, and it no longer reads the buffer iteration i writes);
 *     gnn_shard_iteration_peers   - the iteration (gate, kernel with peer stores);
 *     gnn_peer_publish(value i+1) - behind the kernel on the same stream: the slice's trailing flag row goes to the peers, then - system-scope
 *                                   release - arrive[rank] = i + 1 in every rank's array.
 * The values keep growing across forwards (base + i): nothing is ever reset, so no rank can mistake an old arrival for a new one.  The waits
 * are bounded by GNN_WAIT_MS and raise the sticky error word (k < 0) like every other in-launch wait.  One-layer homogeneous shards without
 * per-arc weights, state widths 17 .. 64 (the wave-specialised kernel); RCCL stays the default transport (gnnkeras_amd/distributed.py) -
 * this path has only ever run between two PROCESSES sharing one GPU (tests/test_gpu_peer.py): no performance claim attaches to it. */
int gnn_device_malloc(void **device_ptr, size_t bytes);              /* hipMalloc + zero fill: an allocation whose BASE pointer can be exported */
int gnn_device_free(void *device_ptr);
int gnn_ipc_export(const void *device_ptr, void *handle64);          /* 64-byte hipIpcMemHandle_t of an allocation's BASE pointer */
int gnn_ipc_open(const void *handle64, void **device_ptr);
int gnn_ipc_close(void *device_ptr);
typedef struct gnn_peer_set {
    int32_t n_peers;
    float *state_out_full[GNN_MAX_PEERS];      /* the peers' full buffer the iteration WRITES (same layout as the own one)             */
    int32_t *arrive[GNN_MAX_PEERS];            /* the peers' arrival arrays [world_size]                                                */
} gnn_peer_set_t;
int gnn_shard_iteration_peers(const gnn_loop_args_t *args, const float *state_in_full, float *state_out_full, int32_t row_base,
                              const int32_t *gate, int32_t n_gate, int32_t gate_stride, int32_t *flag_out, int32_t iteration,
                              const gnn_peer_set_t *peers);
int gnn_peer_wait(const int32_t *arrive_local, int32_t world_size, int32_t rank, int32_t value, float *k_out_error, void *stream);
int gnn_peer_publish(const gnn_peer_set_t *peers, int32_t *arrive_local, const float *flag_row_local, int64_t flag_row_offset_floats,
                     int32_t row_floats, int32_t rank, int32_t value, void *stream);

/* ---- the sharded loop driven from native code (ABI 7; gnnkeras_amd/csrc/shard_loop.hpp) ------------------------------------------------
 * All iterations of a rank in ONE call: own-range partial sums, the halo kernel (whole or in chunk launches), the exchange of the rows
 * just written over the RCCL C API on an exchange stream of the library's own, the stream dependencies between them - the launches
 * gnnkeras_amd/distributed.py issues one at a time from the interpreter, in the same order (bit-identical results), without the
 * interpreter between a chunk's kernel and its sends.  No host synchronisation; the convergence gates stay device words.
 * The communicator is the library's own: rank 0 draws an id (gnn_comm_unique_id, 128 bytes), hands it to the other ranks over whatever
 * the host already has (torch.distributed), every rank calls gnn_comm_create (collective).  RCCL is reached through dlopen: the copy
 * the process has already loaded (librccl.so.1; GNN_RCCL_LIB overrides). */
int gnn_comm_unique_id(void *out128);
int gnn_comm_create(int32_t nranks, int32_t rank, const void *unique_id128, void **comm_out);
int gnn_comm_destroy(void *comm);
typedef struct gnn_shard_loop_args {
    const gnn_loop_args_t *loop;                     /* as for gnn_shard_iteration (gnn_shard_setup has run; `stream` = the compute stream)  */
    const gnn_csr_t *adjacency_own, *adjacency_halo; /* both non-NULL (+ agg_partial): the own-range / halo split with overlap               */
    float *agg_partial;
    float *buf[2];                                   /* the two full state buffers; iteration i reads buf[i & 1], writes buf[(i + 1) & 1]     */
    int32_t row_base, rows_per_slice, chunk;         /* this rank's first row; rows of a slice incl. padding + flag row; the flag row's index  */
    int32_t world_size, rank, SP;                    /* SP = gnn_state_ld(S)                                                                   */
    int32_t first_iteration, n_iterations;           /* iterations [first, first + n) of the loop; n < 0: through loop->max_iteration.  Calls chain:
                                                      * the own-range partial sums of iteration i + 1 are issued at the end of iteration i    */
    int32_t transport;                               /* 0: RCCL all-gather of whole slices; 1: R - 1 point-to-point pairs in one group         */
    int32_t n_chunks;                                /* > 1: chunk launches (gnn_shard_iteration_split_rows), every chunk sent as soon as written */
    const int32_t *chunk_begin;                      /* HOST [n_chunks + 1]: tile-aligned row bounds inside the nominal slice                  */
    const int32_t *node_iota;                        /* DEVICE [n_nodes]: 0, 1, 2, ..                                                          */
    int32_t emulated;                                /* != 0: world_size > 1 without a communicator (one GPU runs one rank's launches: timing) */
    void *comm;                                      /* gnn_comm_create, or NULL at world size 1 / emulated                                    */
} gnn_shard_loop_args_t;
int gnn_shard_loop(const gnn_shard_loop_args_t *args);

/* Keras Dropout / AlphaDropout (the `dropout_rate` / `dropout_pos` / `alphadropout` arguments of the reference MLP builder,
 * MLP.py:25-27, :60-66) in training mode, forward (backward = 0: y = the layer's output for input x) or backward (backward = 1:
 * y = d loss / d input for x = d loss / d output).  The keep mask is a pure function of (key, row, column) - a counter hash,
 * keep <=> lowbias32(lowbias32(key + row) ^ column * 0x9E3779B1) >= rate * 2^32 - so the backward pass of a call regenerates
 * the mask from the key of that call; nothing is stored.  x and y may alias. */
int gnn_dropout(const float *x, int32_t ldx, float *y, int32_t ldy, int32_t M, int32_t H, float rate, uint32_t key, int32_t alpha,
                int32_t backward, void *stream);

/* ---- training building blocks (reference train_step, GNN.py:277-306: tape.gradient through the unrolled loop) ------
 * The backward pass is orchestrated by the host (gnnkeras_amd/Models/training.py) one iteration at a time out of these
 * device primitives; each is a hand-written gfx950 kernel (kernels_train.hpp), float32, no host synchronisation. */
#define GNN_MAX_SEGMENTS 6
/* Y[orow(m), :H] = act( sum_s X_s[row_s(m), :width_s] . W[wrow_s : wrow_s + width_s, :H] + bias + addend[arow(m), :H] )
 * — one Dense layer over a virtual column concatenation (f32 MFMA); the forward layers and dX = dY . W^T of backprop. */
typedef struct gnn_dense_args {
    int32_t M, H, n_segments;
    const float *seg_ptr[GNN_MAX_SEGMENTS];
    const int32_t *seg_rowidx[GNN_MAX_SEGMENTS];   /* NULL = identity */
    int32_t seg_ld[GNN_MAX_SEGMENTS], seg_width[GNN_MAX_SEGMENTS], seg_wrow[GNN_MAX_SEGMENTS];
    const float *W; int32_t ldw;                   /* row-major [K_total][ldw], ldw = 0 means H */
    const float *bias;                             /* [H] or NULL */
    const float *addend; int32_t ld_addend; const int32_t *addend_rowidx;
    int32_t activation;
    float *Y; int32_t ldy; const int32_t *out_rowidx;
    const int32_t *gate;                           /* skip the launch when *gate == 0 */
    void *stream;
    const float *in_center;                        /* ABI 7, optional: in_center[weight row] is subtracted from every input value as it is
                                                    * staged - the consumer of a CENTRED fold (gnn_fold_bn(centred = 1)): BatchNormalization
                                                    * as a (x - mean) + beta instead of a x + (beta - mean a), which cancels the mean against
                                                    * itself and costs digits when the inputs sit far from zero */
} gnn_dense_args_t;
int gnn_dense(const gnn_dense_args_t *args);
/* Wf = diag(gamma/sqrt(var+eps)) W, bf = b + (beta - mean*gamma/sqrt(var+eps)) W  (gamma NULL: plain copy).
 * centred != 0 (ABI 7): bf = b + beta W - the consumer subtracts `mean` from its inputs itself (gnn_dense_args_t::in_center). */
int gnn_fold_bn(const float *W, const float *b, int32_t K, int32_t H, const float *gamma, const float *beta,
                const float *mean, const float *var, float eps, float *Wf, float *bf, int32_t centred, void *stream);
/* P[K,H] (+)= (X[rows] - center)^T dZ ; q[H] (+)= colsum(dZ)   — deterministic two-stage reduction.  `center` (ABI 7; [K] or NULL): the
 * column means of a training-mode BatchNormalization, subtracted as the rows are staged (gnn_first_layer_param_grads(centered = 1)). */
size_t gnn_dense_grad_workspace_bytes(int32_t K, int32_t H, int32_t M);
int gnn_dense_grad(const float *X, int32_t ldx, const int32_t *rowidx, int32_t K, const float *dZ, int32_t ldz, int32_t H,
                   int32_t M, float *P, float *q, int32_t accumulate, const float *center, void *workspace, size_t workspace_bytes,
                   void *stream);
/* dZ = G (.) act'(Y) from the layer output (softmax: Y (.) (G - <G,Y>)) */
int gnn_act_grad(const float *G, int32_t ldg, const float *Y, int32_t ldy, float *dZ, int32_t ldz, int32_t M, int32_t H,
                 int32_t activation, void *stream);
/* batch mean / biased variance of X[rows, :K] (BatchNormalization training=True) + Keras moving-average update */
size_t gnn_colstats_workspace_bytes(int32_t K, int32_t M);
int gnn_colstats(const float *X, int32_t ldx, const int32_t *rowidx, int32_t K, int32_t M, float *mean, float *var,
                 float *moving_mean, float *moving_var, float momentum, const int32_t *gate, void *workspace,
                 size_t workspace_bytes, void *stream);
/* dW, db, dgamma, dbeta (and the two BN input-gradient moments m1, m2) of [BN +] first Dense from P and q.
 * centered != 0 (ABI 7, with BatchNormalization): P is (X - mean)^T dZ already (gnn_dense_grad with `center`): dW = a P + beta q^T and
 * d gamma = rstd sum_h W P without a `- mean (W q)` that would cancel the leading digits of two sums over all rows. */
int gnn_first_layer_param_grads(const float *P, const float *q, const float *W, int32_t K, int32_t H, const float *gamma,
                                const float *beta, const float *mean, const float *var, float eps, int32_t M, float *dW,
                                float *db, float *dgamma, float *dbeta, float *m1, float *m2, int32_t accumulate, int32_t centered,
                                void *stream);
int gnn_bn_input_grad(const float *dy, int32_t ld_dy, const float *x, int32_t ld_x, const int32_t *x_rowidx, int32_t M, int32_t width, int32_t k0,
                      const float *gamma, const float *mean, const float *var, float eps, const float *m1, const float *m2,
                      float *dx, int32_t ld_dx, void *stream);
int gnn_scatter_add_rows(const float *D, int32_t ldd, const int32_t *idx, int32_t M, int32_t width, float *G, int32_t ldg, void *stream);
int gnn_axpby(float a, const float *x, float b, const float *y, float *out, size_t n, void *stream);
/* Keras losses with sample weights, reduction SUM_OVER_BATCH_SIZE: kind 0 categorical_crossentropy, 1 binary_crossentropy,
 * 2 mse, 3 mae.  dp = d loss / d prediction, loss_rows[m] = weighted per-row loss (caller sums / M). */
int gnn_loss_grad(int32_t kind, const float *y, const float *p, const float *sample_weight, int32_t M, int32_t T, float *dp,
                  float *loss_rows, void *stream);
/* Optimizer updates (tf.keras.optimizers.Adam / SGD defaults).  `gate` (ABI 7; NULL = always): a DEVICE word - the launch leaves
 * parameters and slots untouched when *gate == 0.  gnn_train_step hands out such a word (`grads_ok_dev`): 1 when the gradients it left
 * are valid, 0 when its backward launch failed (a grid barrier of the persistent small-graph kernel expired) - so a failed step can
 * never reach the weights, without a host synchronisation between the step and its update. */
int gnn_adam_step(float *p, const float *g, float *m, float *v, size_t n, float lr, float beta1, float beta2, float eps,
                  int32_t step, const int32_t *gate, void *stream);
/* the same update for every variable of a model in one launch: HOST arrays of n_vars device pointers / element counts */
int gnn_adam_multi(float *const *p, const float *const *g, float *const *m, float *const *v, const size_t *n, int32_t n_vars, float lr,
                   float beta1, float beta2, float eps, int32_t step, const int32_t *gate, void *stream);
int gnn_sgd_step(float *p, const float *g, float *velocity, size_t n, float lr, float momentum, const int32_t *gate, void *stream);
int gnn_converged_gated(const float *state, const float *state_old, int32_t n, int32_t dim, int32_t ld, float threshold,
                        const int32_t *gate, int32_t *flag, float *k_out, float k_val, void *stream);
int gnn_aggregate_gated(const gnn_csr_t *csr, const float *X, int32_t ldx, int32_t F, float *out, int32_t ldo,
                        const int32_t *gate, void *stream);

/* ---- batch assembly on the device (reference GraphSequencers.py:42-46, :123-127 -> GraphObject.merge, graph_class.py:386-413) ----
 * A merged batch is the block-diagonal concatenation of its graphs: every array of the merged graph is a run of per-graph
 * segments of the dataset's device-resident arrays, with a per-segment offset added to node / arc ids.  gnn_ragged_copy executes
 * a table of such segment operations in ONE launch.  `desc` and `blk_begin` are DEVICE arrays (the caller uploads them with one
 * small copy per batch): blk_begin[d] = first workgroup of descriptor d (GNN_RC_CHUNK elements per workgroup), blk_begin[n_desc]
 * = n_blocks. */
enum gnn_ragged_kind {
    GNN_RC_COPY_F32 = 0,        /* dst[i] = src[i]                                 (float32)                          */
    GNN_RC_COPY_I32_ADD = 1,    /* dst[i] = src[i] + iadd                          (int32: ids, row pointers)          */
    GNN_RC_COPY_ROWS_ADD2 = 2,  /* float32 rows of `width` columns, columns 0 and 1 += fval (the arcs matrix: float ids)*/
    GNN_RC_FILL_F32 = 3,        /* dst[i] = fval                                                                       */
    GNN_RC_FILL_I32 = 4,        /* dst[i] = iadd                                                                       */
    GNN_RC_IOTA_I32 = 5,        /* dst[i] = iadd + i                                                                   */
    GNN_RC_COPY_U8 = 6          /* dst[i] = src[i]                                 (bytes: bool masks)                 */
};
#define GNN_RC_CHUNK 2048
typedef struct gnn_ragged_desc {
    const void *src;          /* NULL for fills                                                                        */
    void *dst;
    int64_t count;            /* elements                                                                              */
    int32_t kind;             /* enum gnn_ragged_kind                                                                  */
    int32_t iadd;
    float fval;
    int32_t width;            /* GNN_RC_COPY_ROWS_ADD2: columns per row                                                */
} gnn_ragged_desc_t;
int gnn_ragged_copy(const gnn_ragged_desc_t *desc, int32_t n_desc, const int32_t *blk_begin, int32_t n_blocks, void *stream);

/* ---- one whole training step inside the library (reference GNN.py:277-306, CompositeGNN.py:275-304) ------------------------
 * Training-mode forward (BatchNormalization on batch statistics, moving averages updated once per executed iteration for
 * the state network and once for the output network), Keras loss with sample weights, back-propagation through the k
 * executed iterations, optional 1/k on the state-network gradients (`average_st_grads`, :295).  The optimizer update stays
 * with the caller (gnn_adam_step / gnn_sgd_step per variable).  The same arithmetic as the building blocks above driven
 * from Python, with the per-iteration orchestration and the per-segment launches folded into the library: a MUTAG-sized
 * step is bound by launch count and host time, not by bandwidth.
 * `loop` describes graph, networks (their BatchNormalization moving_mean / moving_variance arrays ARE UPDATED in place),
 * focus, out_index, nodegraph, state0, max_iteration, state_threshold and stream exactly as for gnn_loop_forward; its
 * k_out / state_out / out / workspace fields are ignored.  Heterogeneous (composite) models are covered since ABI 6
 * (grad_state_types; gnnkeras_amd/csrc/train_composite.hpp), Dropout layers behind Dense layers since ABI 8 (gnn_dropout_spec_t);
 * Dropout in front of the first Dense and LGNN label gradients are not: the caller uses the building blocks for those.
 * The call synchronises the stream ONCE (to learn k, as the reference does when it divides by k). */
/* ABI 8: the Dropout / AlphaDropout layers of one network (reference MLP.py:25-27, :60-66: `dropout_rate`, `dropout_pos`, `alphadropout`) in
 * a training step.  Layer i of the list sits at POSITION pos[i]: in front of Dense pos[i] (pos = n_layers: behind the last Dense).  The
 * keep mask of a call is the counter hash of gnn_dropout with key = mix32(drop_seed, net_id, call, index[i]) - call = the iteration for a
 * state network, 0 for the output network; mix32(v..): h = 0x9E3779B9; for v: h = lowbias32(h ^ v) - so a host that knows the step's seed
 * can reproduce every mask (oracle/torch_train.py does).  Position 0 (Dropout in front of the first Dense, behind BatchNormalization) is
 * not taken by gnn_train_step: such networks train through the building blocks. */
#define GNN_MAX_DROPOUT 8
typedef struct gnn_dropout_spec {
    int32_t n;                                 /* dropout layers with rate > 0 (0: none)                            */
    int32_t alpha;                             /* != 0: AlphaDropout                                                */
    int32_t net_id;                            /* enters the key: 0.. for state networks (the node type), 1000 for the output network */
    int32_t pos[GNN_MAX_DROPOUT];              /* ascending, 1 .. n_layers                                          */
    int32_t index[GNN_MAX_DROPOUT];            /* the layer's index among the network's dropout layers (enters the key) */
    float rate[GNN_MAX_DROPOUT];               /* 0 < rate < 1                                                      */
} gnn_dropout_spec_t;

typedef struct gnn_mlp_grads {
    float *dgamma, *dbeta;                     /* [in_dim] each; NULL without BatchNormalization                   */
    float *dkernel[GNN_MAX_LAYERS];            /* same shapes as the network's kernels / biases                     */
    float *dbias[GNN_MAX_LAYERS];
} gnn_mlp_grads_t;

typedef struct gnn_train_args {
    gnn_loop_args_t loop;
    gnn_csr_t adjacency_by_source;             /* CSR of Adjacency itself (arcs grouped by SOURCE): transposed aggregate */
    gnn_csr_t nodegraph_by_source;             /* graph focus: CSR of NodeGraph itself (n_dst = n_out nodes, n_src = #graphs) */
    const float *targets;                      /* [n_rows, T]: n_rows = n_out (node / arc focus) or #graphs          */
    const float *sample_weight;                /* [n_rows] or NULL                                                   */
    int32_t loss_kind;                         /* 0 categorical_crossentropy, 1 binary_crossentropy, 2 mse, 3 mae    */
    int32_t average_st_grads;
    float bn_momentum;                         /* 0.99 in Keras                                                      */
    gnn_mlp_grads_t grad_state, grad_output;   /* OUT (overwritten)                                                  */
    float *y_pred;                             /* OUT [n_rows, T] training-mode prediction                            */
    float *state;                              /* OUT [n_nodes, S] state after the k executed iterations              */
    float *loss;                               /* OUT [1] device scalar                                               */
    int32_t *k_host;                           /* OUT host int: iterations executed                                   */
    void *tape; size_t tape_bytes;             /* >= gnn_train_workspace_bytes(args), 256-byte aligned                */
    /* optional (ABI 5): the batch as diagonal blocks.  A merge of small graphs (reference graph_class.py:386-413) has block-
     * diagonal operators; the caller may hand over TILES of at most 64 consecutive nodes cut at graph boundaries, so that no arc
     * of `adjacency` joins two tiles: tile b = nodes [tile_node_begin[b], tile_node_begin[b + 1]).  The small-graph training
     * kernels then keep a tile's state in the LDS of one CU for the whole loop and exchange only the BatchNormalization
     * statistics and the loop condition between tiles.  Same results as without tiles (n_tiles == 0).  An arc that leaves its
     * tile is an error (the call fails).  Ignored when the persistent kernels do not apply (wide / deep state networks, more
     * tiles than CUs, large graphs). */
    const int32_t *tile_node_begin;            /* HOST array [n_tiles + 1], ascending, [0] = 0, [n_tiles] = n_nodes  */
    int32_t n_tiles;
    /* ABI 6: heterogeneous models (loop.composite != 0; reference CompositeGNN.py:275-304 `train_step`): one state network per node
     * type, net_state[t] applied to the rows of type t with ITS BatchNormalization statistics - the gradients of network t land in
     * grad_state_types[t] (grad_state is not used then).  Node, graph and (round 6) arc focus (CompositeGNN.py:315-327); LGNN label
     * gradients train through the building blocks.  From GNN_TRAIN_BIG_MIN_NODES nodes on the step runs the row-streaming kernels in
     * position space (position i = node type_nodes[i]; train_composite_big.hpp): same arguments, same results. */
    gnn_mlp_grads_t grad_state_types[GNN_MAX_TYPES];
    /* ABI 7: validity of the step's gradients WITHOUT a second host synchronisation.  The persistent small-graph backward kernel waits
     * at grid barriers with a bound (GNN_WAIT_MS); when such a wait expires (GPU shared with other long-running work) its gradients are
     * poisoned (NaN) - and the call has returned long before.  So the library keeps one device word at the START of the tape: 1 when
     * every launch of the step completed its protocol, else 0; the BatchNormalization moving averages of the step are updated BEHIND
     * the backward launch and only when the word is 1, and the optimizer entry points take the word as `gate`: a failed step changes
     * nothing.  The caller learns about it for free at the NEXT call's one synchronisation:
     *   grads_ok_dev        optional OUT (host pointer variable): the device word of THIS call (valid until the tape is reused)
     *   prev_grads_ok_host  optional OUT (host int): the word the PREVIOUS call on this tape left, fetched before this call resets it
     *                       (NULL for a fresh tape, whose first word is not a validity word yet).  Written by the time the call returns
     *                       whether it succeeds or fails behind its first launches (a failing call synchronises the stream first);
     *                       a call that fails before them (argument checks) leaves *prev_grads_ok_host as the caller set it - start
     *                       from a sentinel - and the word on the tape untouched. */
    const int32_t **grads_ok_dev;
    int32_t *prev_grads_ok_host;
    /* ABI 8: Dropout layers (all-zero: none).  A state network with Dropout runs the general kernels (one launch per layer and iteration);
     * Dropout inside the OUTPUT network alone leaves the loop on whichever fast path applies. */
    gnn_dropout_spec_t drop_state[GNN_MAX_TYPES];   /* [0] for homogeneous models                                   */
    gnn_dropout_spec_t drop_output;
    uint32_t drop_seed;                             /* the step's seed (every mask of the step derives from it)     */
    /* ABI 9: the training-mode FORWARD alone - `Loop(..., training=True)` (reference GNN.py:245-274: BatchNormalization on the batch
     * statistics of THIS call + its moving-average updates, Dropout masks of `drop_seed`), what LGNN's serial fit() runs on every graph
     * between its layers (LGNN.py:325-337; with focus NODE on a graph-focused model it is the per-node output LGNN feeds to the next layer).
     * No loss, no gradients: targets, sample_weight, loss, grad_*, adjacency_by_source, nodegraph_by_source and the validity word are not
     * touched (they may be NULL / zero; prev_grads_ok_host must be NULL - a forward has no step before it to judge).  y_pred, state and
     * k_host are written as by a step.  Homogeneous models only. */
    int32_t forward_only;
} gnn_train_args_t;
/* Arithmetic: float32 throughout.  On graphs of >= GNN_TRAIN_BIG_MIN_NODES (32 768) nodes the first Dense's forward product and dZ . W^T run on
 * the bf16 matrix cores with every float32 operand split into three bf16 terms (six products, float32 accumulation: the accuracy of a
 * float32 product chain, not its bits; GNN_TRAIN_BF16X6=0 selects float32-input MFMAs), and every kernel of that path subtracts the
 * BatchNormalization column mean from a row as it arrives. */
size_t gnn_train_workspace_bytes(const gnn_train_args_t *args);
int gnn_train_step(const gnn_train_args_t *args);

#ifdef __cplusplus
}
#endif
#endif /* GNNLOOP_H */
