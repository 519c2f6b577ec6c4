// libgnnkeras_torch.so — the PyTorch-ROCm custom-op boundary of the loop (BASELINE.json north_star: "Python host code calls
// hand-written HIP kernels through PyTorch-ROCm custom ops"; SURVEY.md §8b "One custom op (torch TORCH_LIBRARY schema ...), a
// C-ABI shim of the same shape for non-torch callers").
//
//   torch.ops.gnnkeras.loop_forward   (k, state, out) = Loop(...)      reference GNN/Models/GNN.py:245-274, :317-330, :341-346,
//                                                                       CompositeGNN.py:242-272, :315-343
//                                     (+ group_node_begin: merged batches as independent loops; loop_groups_supported asks first)
//   torch.ops.gnnkeras.aggregate      A^T . X                           sparse_dense_matmul(adjoint_a=True): GNN.py:228, :254, :258
//   torch.ops.gnnkeras.pool           NodeGraph^T . out                 GNN.py:345
//   torch.ops.gnnkeras.converged      the predicate of `condition`      GNN.py:196-212
//   torch.ops.gnnkeras.state_step     one `convergence` step            GNN.py:217-236 (aggregated_nodes / aggregated_arcs given:
//                                     the reference's own 8 arguments, gnn_state_step_agg)
//   torch.ops.gnnkeras.mlp_forward    Keras Sequential inference call   GNN.py:234, :273
//
// Thin registrations over the C ABI of libgnnloop.so (include/gnnloop.h): this file owns argument checking (device / dtype /
// shape / contiguity errors are TORCH_CHECKs = RuntimeError), output and workspace allocation through torch's caching
// allocator, and the current HIP stream; every kernel lives in libgnnloop.so.  Registered for the CUDA dispatch key only
// (= HIP on ROCm builds of PyTorch): CPU tensors are rejected by the dispatcher - there is no CPU path.
//
// Encodings shared by the schemas:
//   sparse operator (CSR of A^T, what SparseMatrix.device_csr() holds):  Tensor?[4] {rowptr i32[n_dst+1], src i32[nnz],
//       w f32[nnz]?, row_scale f32[n_dst]?}  +  int[3] {n_dst, n_src, nnz};  an empty list = operator absent
//   network (Keras Sequential of the reference MLP builder):  Tensor[] in get_weights() order {gamma, beta, mean, var}? +
//       {kernel, bias} per Dense  +  int[] spec {in_dim, has_bn, n_layers, units[n_layers], activation ids[n_layers]}
#include <ATen/ATen.h>
#include <c10/hip/HIPStream.h>
#include <torch/library.h>

#include <vector>

#include "../../include/gnnloop.h"

namespace {

using OptTensorList = c10::List<std::optional<at::Tensor>>;

const float *f32(const at::Tensor &t, const char *name, const at::Device &dev) {
    TORCH_CHECK(t.device() == dev, name, ": expected a tensor on ", dev, ", got ", t.device());
    TORCH_CHECK(t.scalar_type() == at::kFloat, name, ": expected float32, got ", t.scalar_type());
    TORCH_CHECK(t.is_contiguous(), name, ": expected a contiguous tensor");
    return t.const_data_ptr<float>();
}
const float *f32(const std::optional<at::Tensor> &t, const char *name, const at::Device &dev) {
    return (t.has_value() && t->defined()) ? f32(*t, name, dev) : nullptr;
}
const int32_t *i32(const at::Tensor &t, const char *name, const at::Device &dev) {
    TORCH_CHECK(t.device() == dev, name, ": expected a tensor on ", dev, ", got ", t.device());
    TORCH_CHECK(t.scalar_type() == at::kInt, name, ": expected int32, got ", t.scalar_type());
    TORCH_CHECK(t.is_contiguous(), name, ": expected a contiguous tensor");
    return t.const_data_ptr<int32_t>();
}
const int32_t *i32(const std::optional<at::Tensor> &t, const char *name, const at::Device &dev) {
    return (t.has_value() && t->defined()) ? i32(*t, name, dev) : nullptr;
}

gnn_csr_t csr_of(const OptTensorList &t, at::IntArrayRef dims, const char *name, const at::Device &dev) {
    gnn_csr_t c{};
    if (t.size() == 0 && dims.size() == 0) return c;
    TORCH_CHECK(t.size() == 4 && dims.size() == 3, name, ": a sparse operator is {rowptr, src, w?, row_scale?} + {n_dst, n_src, nnz}");
    c.n_dst = (int32_t)dims[0]; c.n_src = (int32_t)dims[1]; c.nnz = (int32_t)dims[2];
    TORCH_CHECK(c.n_dst >= 0 && c.n_src >= 0 && c.nnz >= 0, name, ": negative dimension");
    const std::optional<at::Tensor> rp = t.get(0), sr = t.get(1), w = t.get(2), rs = t.get(3);
    TORCH_CHECK(rp.has_value() && rp->defined(), name, ": rowptr is required");
    c.rowptr = i32(*rp, name, dev);
    TORCH_CHECK(rp->numel() == (int64_t)c.n_dst + 1, name, ": rowptr must have n_dst + 1 = ", c.n_dst + 1, " entries, got ", rp->numel());
    c.src = i32(sr, name, dev);
    TORCH_CHECK(c.nnz == 0 || (c.src && sr->numel() == c.nnz), name, ": src must have nnz = ", c.nnz, " entries");
    c.w = f32(w, name, dev);
    TORCH_CHECK(!c.w || w->numel() == c.nnz, name, ": w must have nnz entries");
    c.row_scale = f32(rs, name, dev);
    TORCH_CHECK(!c.row_scale || rs->numel() == c.n_dst, name, ": row_scale must have n_dst entries");
    return c;
}

// fills `m` from (weights, spec) starting at weights[*wpos] / spec[*spos]; advances both (composite: T networks in a row)
void mlp_of(gnn_mlp_t &m, at::TensorList weights, at::IntArrayRef spec, size_t *wpos, size_t *spos, double bn_eps, const char *name,
            const at::Device &dev) {
    m = gnn_mlp_t{};
    TORCH_CHECK(*spos + 3 <= spec.size(), name, ": spec is {in_dim, has_bn, n_layers, units..., activations...}");
    m.in_dim = (int32_t)spec[*spos]; m.has_bn = (int32_t)spec[*spos + 1]; m.n_layers = (int32_t)spec[*spos + 2];
    TORCH_CHECK(m.n_layers >= 1 && m.n_layers <= GNN_MAX_LAYERS, name, ": between 1 and ", GNN_MAX_LAYERS, " Dense layers");
    TORCH_CHECK(*spos + 3 + 2 * (size_t)m.n_layers <= spec.size(), name, ": spec too short for ", m.n_layers, " layers");
    const size_t need = (m.has_bn ? 4 : 0) + 2 * (size_t)m.n_layers;
    TORCH_CHECK(*wpos + need <= weights.size(), name, ": expected ", need, " weight tensors (Keras get_weights() order)");
    if (m.has_bn) {
        m.bn_eps = (float)bn_eps;
        const float **dst[4] = {&m.bn_gamma, &m.bn_beta, &m.bn_mean, &m.bn_var};
        for (int i = 0; i < 4; ++i) {
            const at::Tensor &t = weights[*wpos + i];
            *dst[i] = f32(t, name, dev);
            TORCH_CHECK(t.numel() == m.in_dim, name, ": BatchNormalization arrays must have in_dim = ", m.in_dim, " entries");
        }
        *wpos += 4;
    }
    int fan_in = m.in_dim;
    for (int l = 0; l < m.n_layers; ++l) {
        m.units[l] = (int32_t)spec[*spos + 3 + l];
        m.activation[l] = (int32_t)spec[*spos + 3 + m.n_layers + l];
        const at::Tensor &W = weights[*wpos], &b = weights[*wpos + 1];
        m.kernel[l] = f32(W, name, dev); m.bias[l] = f32(b, name, dev);
        TORCH_CHECK(W.dim() == 2 && W.size(0) == fan_in && W.size(1) == m.units[l], name, ": kernel ", l, " must be [", fan_in, ", ",
                    m.units[l], "], got ", W.sizes());
        TORCH_CHECK(b.numel() == m.units[l], name, ": bias ", l, " must have ", m.units[l], " entries");
        fan_in = m.units[l];
        *wpos += 2;
    }
    *spos += 3 + 2 * (size_t)m.n_layers;
}

void check_rc(int rc) { TORCH_CHECK(rc == 0, "libgnnloop: ", gnn_last_error()); }

void *current_stream(const at::Device &dev) { return (void *)c10::hip::getCurrentHIPStream(dev.index()).stream(); }

// 256-byte aligned workspace out of the caching allocator (stream-ordered reuse: freed when the op returns, handed out again
// only to later work of the same stream)
at::Tensor workspace(size_t bytes, const at::Device &dev, void **ptr) {
    at::Tensor ws = at::empty({(int64_t)bytes + 256}, at::TensorOptions().dtype(at::kByte).device(dev));
    *ptr = (void *)(((uintptr_t)ws.data_ptr() + 255) & ~(uintptr_t)255);
    return ws;
}

struct GraphArgs {           // the graph + state-network part shared by loop_forward and state_step
    gnn_loop_args_t a{};
    at::Tensor arcs_c;
};

void fill_graph(gnn_loop_args_t &a, const at::Tensor &nodes, const at::Tensor &arcs, const OptTensorList &adjacency,
                at::IntArrayRef adjacency_dims, const OptTensorList &arcnode, at::IntArrayRef arcnode_dims, const OptTensorList &hub,
                at::IntArrayRef hub_dims, const at::Device &dev) {
    TORCH_CHECK(nodes.is_cuda(), "nodes: expected a tensor on a HIP device (the message-passing loop has no CPU path)");
    TORCH_CHECK(nodes.dim() == 2 && arcs.dim() == 2 && arcs.size(1) >= 2, "nodes must be [N, L] and arcs [E, 2 + A]");
    a.abi_version = GNN_ABI_VERSION;
    a.n_nodes = (int32_t)nodes.size(0); a.n_arcs = (int32_t)arcs.size(0);
    a.dim_node_label = (int32_t)nodes.size(1); a.dim_arc_label = (int32_t)arcs.size(1) - 2;
    a.nodes = f32(nodes, "nodes", dev); a.ld_nodes = a.dim_node_label;
    a.arc_labels = f32(arcs, "arcs", dev) + 2; a.ld_arcs = (int32_t)arcs.size(1);
    a.adjacency = csr_of(adjacency, adjacency_dims, "adjacency", dev);
    a.arcnode = csr_of(arcnode, arcnode_dims, "arcnode", dev);
    if (hub.size() > 0) {    // hub rows: {light rowptr, src, w?, row_scale?, seg_beg, seg_end} + {n_dst, n_src, nnz, n_segments}
        TORCH_CHECK(hub.size() == 6 && hub_dims.size() == 4, "hub: {rowptr, src, w?, row_scale?, seg_beg, seg_end} + {n_dst, n_src, nnz, n_segments}");
        OptTensorList light;
        for (int i = 0; i < 4; ++i) light.push_back(hub.get(i));
        a.adjacency_light = csr_of(light, hub_dims.slice(0, 3), "hub (light operator)", dev);
        a.heavy_seg_beg = i32(hub.get(4), "hub seg_beg", dev); a.heavy_seg_end = i32(hub.get(5), "hub seg_end", dev);
        a.n_heavy_segments = (int32_t)hub_dims[3];
    }
}

std::tuple<at::Tensor, at::Tensor, at::Tensor> loop_forward(
    const at::Tensor &nodes, const at::Tensor &arcs, const OptTensorList &adjacency, at::IntArrayRef adjacency_dims,
    const OptTensorList &arcnode, at::IntArrayRef arcnode_dims, const OptTensorList &nodegraph, at::IntArrayRef nodegraph_dims,
    at::TensorList net_state_weights, at::IntArrayRef net_state_spec, at::TensorList net_output_weights, at::IntArrayRef net_output_spec,
    double bn_eps, const std::optional<at::Tensor> &state0, const at::Tensor &out_index, const std::optional<at::Tensor> &arc_src,
    const std::optional<at::Tensor> &arc_dst, int64_t state_dim, int64_t max_iteration, double state_threshold, int64_t focus, int64_t flags,
    const OptTensorList &hub, at::IntArrayRef hub_dims, const std::optional<at::Tensor> &type_nodes, at::IntArrayRef type_offsets,
    at::IntArrayRef type_dim_label, const OptTensorList &composite_adjacency, at::IntArrayRef composite_dims, at::IntArrayRef loop_events,
    at::IntArrayRef group_node_begin, at::IntArrayRef group_set_begin) {
    const at::Device dev = nodes.device();
    gnn_loop_args_t a{};
    fill_graph(a, nodes, arcs, adjacency, adjacency_dims, arcnode, arcnode_dims, hub, hub_dims, dev);
    const int T = (int)type_dim_label.size();
    a.composite = T > 0;
    a.n_types = T > 0 ? T : 1;
    TORCH_CHECK(a.n_types <= GNN_MAX_TYPES, "at most ", GNN_MAX_TYPES, " node types");
    size_t wpos = 0, spos = 0;
    for (int t = 0; t < a.n_types; ++t) mlp_of(a.net_state[t], net_state_weights, net_state_spec, &wpos, &spos, bn_eps, "net_state", dev);
    TORCH_CHECK(wpos == net_state_weights.size() && spos == net_state_spec.size(), "net_state: ", a.n_types, " network(s) expected, surplus weights / spec entries");
    wpos = spos = 0;
    mlp_of(a.net_output, net_output_weights, net_output_spec, &wpos, &spos, bn_eps, "net_output", dev);
    if (a.composite) {
        TORCH_CHECK((int)type_offsets.size() == T + 1 && (int)composite_adjacency.size() == 4 * T && (int)composite_dims.size() == 3 * T,
                    "composite: type_offsets[T + 1], composite_adjacency[4 T], composite_dims[3 T] expected for T = ", T);
        a.type_nodes = i32(type_nodes, "type_nodes", dev);
        for (int t = 0; t < T; ++t) {
            a.type_dim_label[t] = (int32_t)type_dim_label[t];
            a.type_offsets[t] = (int32_t)type_offsets[t];
            OptTensorList one;
            for (int i = 0; i < 4; ++i) one.push_back(composite_adjacency.get(4 * t + i));
            a.composite_adjacency[t] = csr_of(one, composite_dims.slice(3 * t, 3), "composite_adjacency", dev);
        }
        a.type_offsets[T] = (int32_t)type_offsets[T];
    }
    TORCH_CHECK(state_dim >= 0 && max_iteration >= 0 && state_threshold >= 0, "state_dim, max_iteration, state_threshold must be >= 0");
    a.state_dim = (int32_t)state_dim; a.max_iteration = (int32_t)max_iteration; a.state_threshold = (float)state_threshold;
    const int S = state_dim > 0 ? (int)state_dim : a.dim_node_label;
    if (state_dim > 0) {
        TORCH_CHECK(state0.has_value() && state0->defined(), "state0 is required when state_dim > 0");
        TORCH_CHECK(state0->dim() == 2 && state0->size(0) == a.n_nodes && state0->size(1) == state_dim, "state0 must be [n_nodes, state_dim] = [",
                    a.n_nodes, ", ", state_dim, "], got ", state0->sizes());
        a.state0 = f32(*state0, "state0", dev);
    }
    TORCH_CHECK(focus >= GNN_FOCUS_NODE && focus <= GNN_FOCUS_GRAPH, "focus must be 0 (node), 1 (arc) or 2 (graph)");
    a.focus = (int32_t)focus;
    a.out_index = i32(out_index, "out_index", dev);
    a.n_out = (int32_t)out_index.numel();
    if (focus == GNN_FOCUS_ARC) {
        a.arc_src = i32(arc_src, "arc_src", dev); a.arc_dst = i32(arc_dst, "arc_dst", dev);
        TORCH_CHECK(a.n_arcs == 0 || (a.arc_src && a.arc_dst && arc_src->numel() == a.n_arcs && arc_dst->numel() == a.n_arcs),
                    "arc focus needs arc_src / arc_dst of n_arcs entries");
    }
    if (focus == GNN_FOCUS_GRAPH) a.nodegraph = csr_of(nodegraph, nodegraph_dims, "nodegraph", dev);
    a.flags = (int32_t)flags;
    if (loop_events.size() == 2) { a.ev_loop_begin = (void *)(uintptr_t)loop_events[0]; a.ev_loop_end = (void *)(uintptr_t)loop_events[1]; }
    a.stream = current_stream(dev);
    // independent convergence groups: batches merged into one call, each stopping on its own (k becomes [n_groups])
    std::vector<int32_t> groups(group_node_begin.begin(), group_node_begin.end());
    if (!groups.empty()) {
        TORCH_CHECK(groups.size() >= 2 && groups.size() <= (size_t)GNN_MAX_GROUPS_RESIDENT + 1, "group_node_begin: between 2 and ", GNN_MAX_GROUPS_RESIDENT + 1, " entries");
        a.group_node_begin = groups.data(); a.n_groups = (int32_t)groups.size() - 1;
    }
    // group sets: parts of one batch that share the loop's condition (k of a set's groups is the same)
    std::vector<int32_t> sets(group_set_begin.begin(), group_set_begin.end());
    if (!sets.empty()) {
        TORCH_CHECK(!groups.empty() && sets.size() >= 2, "group_set_begin needs group_node_begin and at least one set");
        a.group_set_begin = sets.data(); a.n_group_sets = (int32_t)sets.size() - 1;
    }

    const auto opts = at::TensorOptions().dtype(at::kFloat).device(dev);
    const int64_t rows_out = focus == GNN_FOCUS_GRAPH ? a.nodegraph.n_dst : a.n_out;
    at::Tensor k = groups.empty() ? at::empty({}, opts) : at::empty({(int64_t)a.n_groups}, opts);
    at::Tensor state = at::empty({a.n_nodes, S}, opts);
    at::Tensor out = at::empty({rows_out, a.net_output.units[a.net_output.n_layers - 1]}, opts);
    a.k_out = k.data_ptr<float>(); a.state_out = state.data_ptr<float>(); a.out = out.data_ptr<float>();
    const size_t bytes = gnn_loop_workspace_bytes(&a);
    TORCH_CHECK(bytes != 0, "libgnnloop: ", gnn_last_error());
    at::Tensor ws = workspace(bytes, dev, &a.workspace);
    a.workspace_bytes = bytes;
    check_rc(gnn_loop_forward(&a));
    return {k, state, out};
}

// network shape only (no weights): what gnn_loop_groups_supported reads
void mlp_shape_of(gnn_mlp_t &m, at::IntArrayRef spec, const char *name) {
    m = gnn_mlp_t{};
    TORCH_CHECK(spec.size() >= 3, name, ": spec is {in_dim, has_bn, n_layers, units..., activations...}");
    m.in_dim = (int32_t)spec[0]; m.has_bn = (int32_t)spec[1]; m.n_layers = (int32_t)spec[2];
    TORCH_CHECK(m.n_layers >= 1 && m.n_layers <= GNN_MAX_LAYERS && spec.size() == 3 + 2 * (size_t)m.n_layers, name, ": malformed spec");
    for (int l = 0; l < m.n_layers; ++l) { m.units[l] = (int32_t)spec[3 + l]; m.activation[l] = (int32_t)spec[3 + m.n_layers + l]; }
}

// May `loop_forward(..., group_node_begin)` run these batches as independent loops of one call?  (shapes only, no tensors)
int64_t loop_groups_supported(int64_t n_nodes, int64_t dim_node_label, int64_t dim_arc_label, at::IntArrayRef net_state_spec,
                           at::IntArrayRef net_output_spec, int64_t state_dim, int64_t max_iteration, int64_t focus, int64_t flags,
                           int64_t n_out, at::IntArrayRef group_node_begin, at::IntArrayRef group_set_begin) {
    gnn_loop_args_t a{};
    a.abi_version = GNN_ABI_VERSION;
    a.n_nodes = (int32_t)n_nodes; a.dim_node_label = (int32_t)dim_node_label; a.dim_arc_label = (int32_t)dim_arc_label;
    a.n_types = 1;
    mlp_shape_of(a.net_state[0], net_state_spec, "net_state");
    mlp_shape_of(a.net_output, net_output_spec, "net_output");
    a.state_dim = (int32_t)state_dim; a.max_iteration = (int32_t)max_iteration; a.focus = (int32_t)focus; a.flags = (int32_t)flags;
    a.n_out = (int32_t)n_out;
    std::vector<int32_t> groups(group_node_begin.begin(), group_node_begin.end());
    if (groups.size() < 2 || groups.size() > (size_t)GNN_MAX_GROUPS_RESIDENT + 1) return 0;
    a.group_node_begin = groups.data(); a.n_groups = (int32_t)groups.size() - 1;
    std::vector<int32_t> sets(group_set_begin.begin(), group_set_begin.end());
    if (sets.size() >= 2) { a.group_set_begin = sets.data(); a.n_group_sets = (int32_t)sets.size() - 1; }
    return gnn_loop_groups_supported(&a);
}

at::Tensor aggregate(const OptTensorList &csr, at::IntArrayRef dims, const at::Tensor &X) {
    const at::Device dev = X.device();
    TORCH_CHECK(X.is_cuda(), "X: expected a tensor on a HIP device");
    const gnn_csr_t c = csr_of(csr, dims, "csr", dev);
    TORCH_CHECK(X.dim() == 2 && X.size(0) == c.n_src, "X must be [n_src, F] = [", c.n_src, ", F], got ", X.sizes());
    const float *x = f32(X, "X", dev);
    at::Tensor out = at::empty({c.n_dst, X.size(1)}, X.options());
    check_rc(gnn_aggregate(&c, x, (int32_t)X.size(1), (int32_t)X.size(1), out.data_ptr<float>(), (int32_t)X.size(1), current_stream(dev)));
    return out;
}

at::Tensor converged(const at::Tensor &state, const std::optional<at::Tensor> &state_old, double threshold) {
    const at::Device dev = state.device();
    TORCH_CHECK(state.is_cuda(), "state: expected a tensor on a HIP device");
    TORCH_CHECK(state.dim() == 2 && state.size(1) >= 1, "state must be [N, S], S >= 1");
    const float *s = f32(state, "state", dev), *so = f32(state_old, "state_old", dev);
    TORCH_CHECK(!so || state_old->sizes() == state.sizes(), "state_old must have the shape of state");
    at::Tensor flag = at::empty({1}, at::TensorOptions().dtype(at::kInt).device(dev));
    check_rc(gnn_converged(s, so, (int32_t)state.size(0), (int32_t)state.size(1), (int32_t)state.size(1), (float)threshold,
                           flag.data_ptr<int32_t>(), current_stream(dev)));
    return flag;
}

std::tuple<at::Tensor, at::Tensor> state_step(const at::Tensor &nodes, const at::Tensor &arcs, const OptTensorList &adjacency,
                                              at::IntArrayRef adjacency_dims, const OptTensorList &arcnode, at::IntArrayRef arcnode_dims,
                                              at::TensorList net_state_weights, at::IntArrayRef net_state_spec, double bn_eps,
                                              const at::Tensor &state, int64_t state_dim, double state_threshold, int64_t flags,
                                              const OptTensorList &hub, at::IntArrayRef hub_dims, const std::optional<at::Tensor> &type_nodes,
                                              at::IntArrayRef type_offsets, at::IntArrayRef type_dim_label,
                                              const OptTensorList &composite_adjacency, at::IntArrayRef composite_dims,
                                              const std::optional<at::Tensor> &aggregated_nodes, const std::optional<at::Tensor> &aggregated_arcs) {
    const at::Device dev = nodes.device();
    gnn_loop_args_t a{};
    fill_graph(a, nodes, arcs, adjacency, adjacency_dims, arcnode, arcnode_dims, hub, hub_dims, dev);
    // the reference's own argument list (GNN.py:217): the aggregates formed once by Loop are handed in; arcs / arcnode / the composite
    // adjacencies are then not read (arcs may be [0, 2 + A])
    const bool given = (aggregated_nodes.has_value() && aggregated_nodes->defined()) || (aggregated_arcs.has_value() && aggregated_arcs->defined());
    const int T = (int)type_dim_label.size();
    a.composite = T > 0;
    a.n_types = T > 0 ? T : 1;
    TORCH_CHECK(a.n_types <= GNN_MAX_TYPES, "at most ", GNN_MAX_TYPES, " node types");
    size_t wpos = 0, spos = 0;
    for (int t = 0; t < a.n_types; ++t) mlp_of(a.net_state[t], net_state_weights, net_state_spec, &wpos, &spos, bn_eps, "net_state", dev);
    TORCH_CHECK(wpos == net_state_weights.size() && spos == net_state_spec.size(), "net_state: ", a.n_types, " network(s) expected, surplus weights / spec entries");
    if (a.composite) {       // CompositeGNNnodeBased.convergence (CompositeGNN.py:215-234)
        TORCH_CHECK((int)type_offsets.size() == T + 1 && (given || ((int)composite_adjacency.size() == 4 * T && (int)composite_dims.size() == 3 * T)),
                    "composite: type_offsets[T + 1], composite_adjacency[4 T], composite_dims[3 T] expected for T = ", T);
        a.type_nodes = i32(type_nodes, "type_nodes", dev);
        for (int t = 0; t < T; ++t) {
            a.type_dim_label[t] = (int32_t)type_dim_label[t];
            a.type_offsets[t] = (int32_t)type_offsets[t];
            if (given) continue;
            OptTensorList one;
            for (int i = 0; i < 4; ++i) one.push_back(composite_adjacency.get(4 * t + i));
            a.composite_adjacency[t] = csr_of(one, composite_dims.slice(3 * t, 3), "composite_adjacency", dev);
        }
        a.type_offsets[T] = (int32_t)type_offsets[T];
    }
    // gnn_state_step validates the state network(s) only; the output network of the args is a placeholder of the right input width
    a.net_output = gnn_mlp_t{};
    a.net_output.in_dim = a.composite ? (state_dim > 0 ? (int32_t)state_dim : a.dim_node_label)
                                      : (state_dim > 0 ? (int32_t)state_dim + a.dim_node_label : a.dim_node_label);
    a.net_output.n_layers = 1; a.net_output.units[0] = 1;
    a.state_dim = (int32_t)state_dim; a.max_iteration = 1; a.state_threshold = (float)state_threshold;
    a.focus = GNN_FOCUS_NODE; a.flags = (int32_t)flags;
    const int S = state_dim > 0 ? (int)state_dim : a.dim_node_label;
    TORCH_CHECK(state.dim() == 2 && state.size(0) == a.n_nodes && state.size(1) == S, "state must be [n_nodes, S] = [", a.n_nodes, ", ", S, "]");
    const float *s = f32(state, "state", dev);
    a.state0 = s;
    a.stream = current_stream(dev);
    at::Tensor out = at::empty_like(state), flag = at::empty({1}, at::TensorOptions().dtype(at::kInt).device(dev));
    const size_t bytes = gnn_loop_workspace_bytes(&a);
    TORCH_CHECK(bytes != 0, "libgnnloop: ", gnn_last_error());
    at::Tensor ws = workspace(bytes, dev, &a.workspace);
    a.workspace_bytes = bytes;
    if (given) {
        // [n_nodes, >= needed columns] float32 matrices whose rows may be strided (column blocks of `aggregated_component`)
        auto view = [&](const std::optional<at::Tensor> &t, const char *name, int32_t *ld) -> const float * {
            if (!(t.has_value() && t->defined()) || t->numel() == 0) { *ld = 0; return nullptr; }
            TORCH_CHECK(t->device() == dev && t->scalar_type() == at::kFloat, name, ": expected a float32 tensor on ", dev);
            TORCH_CHECK(t->dim() == 2 && t->size(0) == a.n_nodes && t->stride(1) == 1, name, ": expected [n_nodes, columns] with unit column stride");
            *ld = (int32_t)std::max<int64_t>(t->stride(0), t->size(1));
            return t->const_data_ptr<float>();
        };
        int32_t ld_n = 0, ld_a = 0;
        const float *an = view(aggregated_nodes, "aggregated_nodes", &ld_n), *aa = view(aggregated_arcs, "aggregated_arcs", &ld_a);
        int need_n = 0;
        if (a.composite) for (int t = 0; t < T; ++t) need_n += a.type_dim_label[t];
        else need_n = state_dim > 0 ? a.dim_node_label : 0;
        TORCH_CHECK(a.n_nodes == 0 || need_n == 0 || (an && aggregated_nodes->size(1) == need_n), "aggregated_nodes must be [n_nodes, ", need_n, "]");
        TORCH_CHECK(a.n_nodes == 0 || a.dim_arc_label == 0 || (aa && aggregated_arcs->size(1) == a.dim_arc_label), "aggregated_arcs must be [n_nodes, ",
                    a.dim_arc_label, "]");
        check_rc(gnn_state_step_agg(&a, s, an, ld_n, aa, ld_a, out.data_ptr<float>(), flag.data_ptr<int32_t>()));
    } else {
        check_rc(gnn_state_step(&a, s, out.data_ptr<float>(), flag.data_ptr<int32_t>()));
    }
    return {out, flag};
}

at::Tensor mlp_forward(at::TensorList weights, at::IntArrayRef spec, double bn_eps, const at::Tensor &X) {
    const at::Device dev = X.device();
    TORCH_CHECK(X.is_cuda(), "X: expected a tensor on a HIP device");
    gnn_mlp_t m;
    size_t wpos = 0, spos = 0;
    mlp_of(m, weights, spec, &wpos, &spos, bn_eps, "mlp", dev);
    TORCH_CHECK(X.dim() == 2 && X.size(1) == m.in_dim, "X must be [M, in_dim] = [M, ", m.in_dim, "], got ", X.sizes());
    const float *x = f32(X, "X", dev);
    const int32_t M = (int32_t)X.size(0);
    at::Tensor Y = at::empty({M, m.units[m.n_layers - 1]}, X.options());
    const size_t bytes = gnn_mlp_workspace_bytes(&m, M);
    void *wsp;
    at::Tensor ws = workspace(bytes, dev, &wsp);
    check_rc(gnn_mlp_forward(&m, x, m.in_dim, M, Y.data_ptr<float>(), (int32_t)Y.size(1), wsp, bytes, current_stream(dev)));
    return Y;
}

}  // namespace

TORCH_LIBRARY(gnnkeras, m) {
    m.def("loop_forward(Tensor nodes, Tensor arcs, Tensor?[] adjacency, int[] adjacency_dims, Tensor?[] arcnode, int[] arcnode_dims, "
          "Tensor?[] nodegraph, int[] nodegraph_dims, Tensor[] net_state_weights, int[] net_state_spec, Tensor[] net_output_weights, "
          "int[] net_output_spec, float bn_eps, Tensor? state0, Tensor out_index, Tensor? arc_src, Tensor? arc_dst, int state_dim, "
          "int max_iteration, float state_threshold, int focus, int flags, Tensor?[] hub, int[] hub_dims, Tensor? type_nodes, "
          "int[] type_offsets, int[] type_dim_label, Tensor?[] composite_adjacency, int[] composite_dims, int[] loop_events, "
          "int[] group_node_begin=[], int[] group_set_begin=[]) -> (Tensor k, Tensor state, Tensor out)");
    m.def("loop_groups_supported(int n_nodes, int dim_node_label, int dim_arc_label, int[] net_state_spec, int[] net_output_spec, "
          "int state_dim, int max_iteration, int focus, int flags, int n_out, int[] group_node_begin, int[] group_set_begin=[]) -> int", &loop_groups_supported);
    m.def("aggregate(Tensor?[] csr, int[] dims, Tensor X) -> Tensor");
    m.def("pool(Tensor?[] nodegraph, int[] dims, Tensor out_nodes) -> Tensor");
    m.def("converged(Tensor state, Tensor? state_old, float threshold) -> Tensor");
    m.def("state_step(Tensor nodes, Tensor arcs, Tensor?[] adjacency, int[] adjacency_dims, Tensor?[] arcnode, int[] arcnode_dims, "
          "Tensor[] net_state_weights, int[] net_state_spec, float bn_eps, Tensor state, int state_dim, float state_threshold, int flags, "
          "Tensor?[] hub, int[] hub_dims, Tensor? type_nodes, int[] type_offsets, int[] type_dim_label, "
          "Tensor?[] composite_adjacency, int[] composite_dims, Tensor? aggregated_nodes=None, Tensor? aggregated_arcs=None) -> (Tensor state_new, Tensor moving)");
    m.def("mlp_forward(Tensor[] weights, int[] spec, float bn_eps, Tensor X) -> Tensor");
}

TORCH_LIBRARY_IMPL(gnnkeras, CUDA, m) {      // dispatch key CUDA = HIP devices on ROCm builds of PyTorch
    m.impl("loop_forward", &loop_forward);
    m.impl("aggregate", &aggregate);
    m.impl("pool", &aggregate);
    m.impl("converged", &converged);
    m.impl("state_step", &state_step);
    m.impl("mlp_forward", &mlp_forward);
}
