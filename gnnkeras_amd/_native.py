"""ctypes binding of libgnnloop.so (C ABI: include/gnnloop.h) — the only way the Python host code reaches the GPU
arithmetic of the loop.  There is deliberately no CPU fallback: if the shared library is missing or the tensors are
not on a HIP device, the calls raise."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
# GNNKERAS_AMD_LIB: load another build of the same sources (tests: the debug build whose in-launch waits expire at once)
LIB_PATH = os.environ.get('GNNKERAS_AMD_LIB') or os.path.join(CSRC, 'libgnnloop.so')

GNN_ABI_VERSION = 9
GNN_MAX_LAYERS = 8
GNN_MAX_TYPES = 8

ACTIVATIONS = {'linear': 0, None: 0, 'relu': 1, 'selu': 2, 'tanh': 3, 'sigmoid': 4, 'elu': 5, 'softplus': 6,
               'softmax': 7}
FOCUS = {'n': 0, 'a': 1, 'g': 2}
FLAG_UNFUSED = 1
FLAG_NO_EARLY_EXIT = 2
FLAG_FUSED_GEN_MASK = 7 << 4
FLAG_FUSED_GEN2, FLAG_FUSED_GEN4, FLAG_FUSED_GEN5, FLAG_FUSED_GEN6, FLAG_FUSED_GEN7 = 2 << 4, 4 << 4, 5 << 4, 6 << 4, 7 << 4     # pin the fused-kernel generation (tests, tuning)

EXPORTS = ['gnn_last_error', 'gnn_last_kernel_name', 'gnn_abi_version', 'gnn_struct_size', 'gnn_loop_workspace_bytes', 'gnn_loop_forward', 'gnn_loop_groups_supported', 'gnn_aggregate',
           'gnn_mlp_workspace_bytes', 'gnn_mlp_forward', 'gnn_converged', 'gnn_state_step', 'gnn_state_step_agg', 'gnn_state_ld', 'gnn_debug_occupy', 'gnn_debug_occupy_until', 'gnn_debug_expiry_beacon', 'gnn_debug_host_flag',
           'gnn_device_malloc', 'gnn_device_free', 'gnn_ipc_export', 'gnn_ipc_open', 'gnn_ipc_close', 'gnn_shard_iteration_peers', 'gnn_peer_wait', 'gnn_peer_publish', 'gnn_shard_iteration_split_rows',
           'gnn_shard_setup', 'gnn_shard_iteration', 'gnn_shard_output', 'gnn_gather_rows',
           'gnn_shard_can_split', 'gnn_shard_partial', 'gnn_shard_iteration_split',
           'gnn_dense', 'gnn_fold_bn', 'gnn_dense_grad_workspace_bytes', 'gnn_dense_grad', 'gnn_act_grad',
           'gnn_colstats_workspace_bytes', 'gnn_colstats', 'gnn_first_layer_param_grads', 'gnn_bn_input_grad',
           'gnn_scatter_add_rows', 'gnn_axpby', 'gnn_loss_grad', 'gnn_dropout', 'gnn_adam_step', 'gnn_adam_multi', 'gnn_sgd_step',
           'gnn_converged_gated', 'gnn_aggregate_gated', 'gnn_train_workspace_bytes', 'gnn_train_step', 'gnn_ragged_copy',
           'gnn_comm_unique_id', 'gnn_comm_create', 'gnn_comm_destroy', 'gnn_shard_loop']
GNN_MAX_SEGMENTS = 6
LOSSES = {'categorical_crossentropy': 0, 'cce': 0, 'binary_crossentropy': 1, 'bce': 1, 'mse': 2,
          'mean_squared_error': 2, 'mae': 3, 'mean_absolute_error': 3}

_f32p = C.POINTER(C.c_float)
_i32p = C.POINTER(C.c_int32)


class CSR(C.Structure):
    _fields_ = [('n_dst', C.c_int32), ('n_src', C.c_int32), ('nnz', C.c_int32),
                ('rowptr', C.c_void_p), ('src', C.c_void_p), ('w', C.c_void_p), ('row_scale', C.c_void_p)]


class MLP(C.Structure):
    _fields_ = [('in_dim', C.c_int32), ('n_layers', C.c_int32),
                ('units', C.c_int32 * GNN_MAX_LAYERS), ('activation', C.c_int32 * GNN_MAX_LAYERS),
                ('kernel', C.c_void_p * GNN_MAX_LAYERS), ('bias', C.c_void_p * GNN_MAX_LAYERS),
                ('has_bn', C.c_int32), ('bn_eps', C.c_float),
                ('bn_gamma', C.c_void_p), ('bn_beta', C.c_void_p), ('bn_mean', C.c_void_p), ('bn_var', C.c_void_p)]


class LoopArgs(C.Structure):
    _fields_ = [('abi_version', C.c_int32), ('composite', C.c_int32),
                ('n_nodes', C.c_int32), ('n_arcs', C.c_int32), ('dim_node_label', C.c_int32),
                ('dim_arc_label', C.c_int32),
                ('nodes', C.c_void_p), ('ld_nodes', C.c_int32),
                ('arc_labels', C.c_void_p), ('ld_arcs', C.c_int32),
                ('adjacency', CSR), ('arcnode', CSR),
                ('n_types', C.c_int32), ('type_dim_label', C.c_int32 * GNN_MAX_TYPES),
                ('type_nodes', C.c_void_p), ('type_offsets', C.c_int32 * (GNN_MAX_TYPES + 1)),
                ('composite_adjacency', CSR * GNN_MAX_TYPES),
                ('net_state', MLP * GNN_MAX_TYPES), ('net_output', MLP),
                ('state_dim', C.c_int32), ('max_iteration', C.c_int32), ('state_threshold', C.c_float),
                ('state0', C.c_void_p),
                ('focus', C.c_int32), ('n_out', C.c_int32), ('out_index', C.c_void_p),
                ('arc_src', C.c_void_p), ('arc_dst', C.c_void_p), ('nodegraph', CSR),
                ('k_out', C.c_void_p), ('state_out', C.c_void_p), ('out', C.c_void_p),
                ('workspace', C.c_void_p), ('workspace_bytes', C.c_size_t), ('stream', C.c_void_p),
                ('flags', C.c_int32), ('nodes_src', C.c_void_p), ('ld_nodes_src', C.c_int32),
                ('adjacency_light', CSR), ('heavy_seg_beg', C.c_void_p), ('heavy_seg_end', C.c_void_p),
                ('n_heavy_segments', C.c_int32),
                ('ev_loop_begin', C.c_void_p), ('ev_loop_end', C.c_void_p),
                ('group_node_begin', C.c_void_p), ('n_groups', C.c_int32),
                ('group_set_begin', C.c_void_p), ('n_group_sets', C.c_int32)]


class DenseArgs(C.Structure):
    _fields_ = [('M', C.c_int32), ('H', C.c_int32), ('n_segments', C.c_int32),
                ('seg_ptr', C.c_void_p * GNN_MAX_SEGMENTS), ('seg_rowidx', C.c_void_p * GNN_MAX_SEGMENTS),
                ('seg_ld', C.c_int32 * GNN_MAX_SEGMENTS), ('seg_width', C.c_int32 * GNN_MAX_SEGMENTS),
                ('seg_wrow', C.c_int32 * GNN_MAX_SEGMENTS),
                ('W', C.c_void_p), ('ldw', C.c_int32), ('bias', C.c_void_p),
                ('addend', C.c_void_p), ('ld_addend', C.c_int32), ('addend_rowidx', C.c_void_p),
                ('activation', C.c_int32), ('Y', C.c_void_p), ('ldy', C.c_int32), ('out_rowidx', C.c_void_p),
                ('gate', C.c_void_p), ('stream', C.c_void_p), ('in_center', C.c_void_p)]


class MLPGrads(C.Structure):
    _fields_ = [('dgamma', C.c_void_p), ('dbeta', C.c_void_p),
                ('dkernel', C.c_void_p * GNN_MAX_LAYERS), ('dbias', C.c_void_p * GNN_MAX_LAYERS)]


GNN_MAX_DROPOUT = 8


class DropoutSpec(C.Structure):         # gnn_dropout_spec_t (ABI 8)
    _fields_ = [('n', C.c_int32), ('alpha', C.c_int32), ('net_id', C.c_int32), ('pos', C.c_int32 * GNN_MAX_DROPOUT),
                ('index', C.c_int32 * GNN_MAX_DROPOUT), ('rate', C.c_float * GNN_MAX_DROPOUT)]


GNN_MAX_PEERS = 7


class PeerSet(C.Structure):             # gnn_peer_set_t (ABI 8): the peers' full state buffer an iteration writes + their arrival arrays
    _fields_ = [('n_peers', C.c_int32), ('state_out_full', C.c_void_p * GNN_MAX_PEERS), ('arrive', C.c_void_p * GNN_MAX_PEERS)]


class TrainArgs(C.Structure):
    _fields_ = [('loop', LoopArgs), ('adjacency_by_source', CSR), ('nodegraph_by_source', CSR),
                ('targets', C.c_void_p), ('sample_weight', C.c_void_p), ('loss_kind', C.c_int32),
                ('average_st_grads', C.c_int32), ('bn_momentum', C.c_float),
                ('grad_state', MLPGrads), ('grad_output', MLPGrads),
                ('y_pred', C.c_void_p), ('state', C.c_void_p), ('loss', C.c_void_p), ('k_host', C.POINTER(C.c_int32)),
                ('tape', C.c_void_p), ('tape_bytes', C.c_size_t), ('tile_node_begin', C.c_void_p), ('n_tiles', C.c_int32),
                ('grad_state_types', MLPGrads * GNN_MAX_TYPES),          # ABI 6: one gradient holder per node type (composite models)
                # ABI 7: the device word that says whether the step's gradients are valid (the optimizers' gate) and the previous step's, read for free
                ('grads_ok_dev', C.POINTER(C.c_void_p)), ('prev_grads_ok_host', C.POINTER(C.c_int32)),
                # ABI 8: the networks' Dropout layers (positions >= 1) and the step's mask seed
                ('drop_state', DropoutSpec * GNN_MAX_TYPES), ('drop_output', DropoutSpec), ('drop_seed', C.c_uint32),
                # ABI 9: the training-mode forward alone (no loss, no gradients)
                ('forward_only', C.c_int32)]


class ShardLoopArgs(C.Structure):       # gnn_shard_loop_args_t (ABI 7): the sharded loop driven from native code (csrc/shard_loop.hpp)
    _fields_ = [('loop', C.POINTER(LoopArgs)), ('adjacency_own', C.POINTER(CSR)), ('adjacency_halo', C.POINTER(CSR)),
                ('agg_partial', C.c_void_p), ('buf', C.c_void_p * 2),
                ('row_base', C.c_int32), ('rows_per_slice', C.c_int32), ('chunk', C.c_int32),
                ('world_size', C.c_int32), ('rank', C.c_int32), ('SP', C.c_int32),
                ('first_iteration', C.c_int32), ('n_iterations', C.c_int32), ('transport', C.c_int32), ('n_chunks', C.c_int32),
                ('chunk_begin', C.POINTER(C.c_int32)), ('node_iota', C.c_void_p), ('emulated', C.c_int32), ('comm', C.c_void_p)]


class RaggedDesc(C.Structure):
    _fields_ = [('src', C.c_void_p), ('dst', C.c_void_p), ('count', C.c_int64), ('kind', C.c_int32), ('iadd', C.c_int32),
                ('fval', C.c_float), ('width', C.c_int32)]


RC_COPY_F32, RC_COPY_I32_ADD, RC_COPY_ROWS_ADD2, RC_FILL_F32, RC_FILL_I32, RC_IOTA_I32, RC_COPY_U8 = range(7)
RC_CHUNK = 2048


class NativeError(RuntimeError):
    pass


def build(verbose: bool = False) -> str:
    """Compile libgnnloop.so for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    res = subprocess.run(['make', '-j3', '-C', CSRC, 'all'], capture_output=True, text=True)      # the sanitizer build of the host side is tests/test_abi.py's own (`make asan`)
    if verbose or res.returncode:
        print(res.stdout, res.stderr)
    if res.returncode:
        raise NativeError('building libgnnloop.so failed:\n' + res.stderr[-4000:])
    return LIB_PATH


def source_hash() -> str:
    """Identity of the library's SOURCES (every file of csrc/ that make compiles + the C header), 16 hex digits of a SHA-256 over
    (name, content) in name order.  Computable where there is no .git (the GPU box gets a snapshot without it): the PMC traffic records
    of profiles/hbm_traffic.json carry it, and bench.py reports a record's traffic only when it was taken on THESE sources."""
    import hashlib
    h = hashlib.sha256()
    files = sorted(f for f in os.listdir(CSRC) if f.endswith(('.hip', '.hpp', '.cpp')) or f == 'Makefile')
    paths = [os.path.join(CSRC, f) for f in files] + [os.path.join(os.path.dirname(HERE), 'include', 'gnnloop.h')]
    for path in paths:
        h.update(os.path.basename(path).encode() + b'\0')
        with open(path, 'rb') as fh: h.update(fh.read())
        h.update(b'\0')
    return h.hexdigest()[:16]


_lib = None


def lib():
    """The loaded library; raises if it has not been built (no silent fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise NativeError(f'{LIB_PATH} is missing: run `python -c "import __graft_entry__ as g; g.build()"` '
                              f'or `make -C {CSRC}`. There is no CPU fallback for the HIP path.')
        l = C.CDLL(LIB_PATH)
        l.gnn_last_error.restype = C.c_char_p
        l.gnn_last_kernel_name.restype = C.c_char_p
        l.gnn_abi_version.restype = C.c_int
        l.gnn_loop_workspace_bytes.restype = C.c_size_t
        l.gnn_loop_workspace_bytes.argtypes = [C.POINTER(LoopArgs)]
        l.gnn_loop_forward.restype = C.c_int
        l.gnn_loop_forward.argtypes = [C.POINTER(LoopArgs)]
        l.gnn_aggregate.restype = C.c_int
        l.gnn_aggregate.argtypes = [C.POINTER(CSR), C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p]
        l.gnn_mlp_workspace_bytes.restype = C.c_size_t
        l.gnn_mlp_workspace_bytes.argtypes = [C.POINTER(MLP), C.c_int32]
        l.gnn_mlp_forward.restype = C.c_int
        l.gnn_mlp_forward.argtypes = [C.POINTER(MLP), C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_int32,
                                      C.c_void_p, C.c_size_t, C.c_void_p]
        l.gnn_converged.restype = C.c_int
        l.gnn_converged.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_void_p,
                                    C.c_void_p]
        l.gnn_state_step.restype = C.c_int
        l.gnn_state_step.argtypes = [C.POINTER(LoopArgs), C.c_void_p, C.c_void_p, C.c_void_p]
        l.gnn_debug_occupy.restype = C.c_int
        l.gnn_debug_occupy.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.c_void_p]
        l.gnn_debug_occupy_until.restype = C.c_int
        l.gnn_debug_occupy_until.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]
        l.gnn_device_malloc.restype = C.c_int
        l.gnn_device_malloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
        l.gnn_device_free.restype = C.c_int
        l.gnn_device_free.argtypes = [C.c_void_p]
        for fn_ in (l.gnn_ipc_export, l.gnn_ipc_close): fn_.restype = C.c_int
        l.gnn_ipc_export.argtypes = [C.c_void_p, C.c_void_p]
        l.gnn_ipc_close.argtypes = [C.c_void_p]
        l.gnn_ipc_open.restype = C.c_int
        l.gnn_ipc_open.argtypes = [C.c_void_p, C.POINTER(C.c_void_p)]
        l.gnn_shard_iteration_peers.restype = C.c_int
        l.gnn_shard_iteration_peers.argtypes = [C.POINTER(LoopArgs), C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_int32,
                                                C.POINTER(PeerSet)]
        l.gnn_peer_wait.restype = C.c_int
        l.gnn_peer_wait.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]
        l.gnn_peer_publish.restype = C.c_int
        l.gnn_peer_publish.argtypes = [C.POINTER(PeerSet), C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]
        l.gnn_debug_host_flag.restype = C.c_int
        l.gnn_debug_host_flag.argtypes = [C.POINTER(C.POINTER(C.c_int32)), C.POINTER(C.c_void_p)]
        l.gnn_debug_expiry_beacon.restype = C.c_int
        l.gnn_debug_expiry_beacon.argtypes = [C.POINTER(C.c_void_p), C.c_int32, C.c_void_p]
        l.gnn_state_step_agg.restype = C.c_int
        l.gnn_state_step_agg.argtypes = [C.POINTER(LoopArgs), C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]
        l.gnn_state_ld.restype = C.c_int32
        l.gnn_state_ld.argtypes = [C.c_int32]
        l.gnn_shard_setup.restype = C.c_int
        l.gnn_shard_setup.argtypes = [C.POINTER(LoopArgs)]
        l.gnn_shard_iteration.restype = C.c_int
        l.gnn_shard_iteration.argtypes = [C.POINTER(LoopArgs), C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32,
                                          C.c_int32, C.c_void_p, C.c_int32]
        l.gnn_shard_output.restype = C.c_int
        l.gnn_shard_output.argtypes = [C.POINTER(LoopArgs), C.c_void_p, C.c_void_p, C.c_int32]
        vp, i32, f32, sz = C.c_void_p, C.c_int32, C.c_float, C.c_size_t
        protos = {
            'gnn_shard_can_split': (C.c_int, [C.POINTER(LoopArgs)]),
            'gnn_shard_partial': (C.c_int, [C.POINTER(LoopArgs), C.POINTER(CSR), vp, vp]),
            'gnn_shard_iteration_split': (C.c_int, [C.POINTER(LoopArgs), C.POINTER(CSR), vp, vp, vp, i32, vp, i32, i32, vp, i32]),
            'gnn_shard_iteration_split_rows': (C.c_int, [C.POINTER(LoopArgs), C.POINTER(CSR), vp, vp, vp, i32, vp, i32, i32, vp, i32, vp, i32, i32]),
            'gnn_dense': (C.c_int, [C.POINTER(DenseArgs)]),
            'gnn_gather_rows': (C.c_int, [vp, i32, vp, i32, i32, vp, i32, vp]),
            'gnn_fold_bn': (C.c_int, [vp, vp, i32, i32, vp, vp, vp, vp, f32, vp, vp, i32, vp]),
            'gnn_dense_grad_workspace_bytes': (sz, [i32, i32, i32]),
            'gnn_dense_grad': (C.c_int, [vp, i32, vp, i32, vp, i32, i32, i32, vp, vp, i32, vp, vp, sz, vp]),
            'gnn_act_grad': (C.c_int, [vp, i32, vp, i32, vp, i32, i32, i32, i32, vp]),
            'gnn_colstats_workspace_bytes': (sz, [i32, i32]),
            'gnn_colstats': (C.c_int, [vp, i32, vp, i32, i32, vp, vp, vp, vp, f32, vp, vp, sz, vp]),
            'gnn_first_layer_param_grads': (C.c_int, [vp, vp, vp, i32, i32, vp, vp, vp, vp, f32, i32, vp, vp, vp, vp, vp, vp, i32, i32, vp]),
            'gnn_bn_input_grad': (C.c_int, [vp, i32, vp, i32, vp, i32, i32, i32, vp, vp, vp, f32, vp, vp, vp, i32, vp]),
            'gnn_scatter_add_rows': (C.c_int, [vp, i32, vp, i32, i32, vp, i32, vp]),
            'gnn_axpby': (C.c_int, [f32, vp, f32, vp, vp, sz, vp]),
            'gnn_loss_grad': (C.c_int, [i32, vp, vp, vp, i32, i32, vp, vp, vp]),
            'gnn_dropout': (C.c_int, [vp, i32, vp, i32, i32, i32, f32, C.c_uint32, i32, i32, vp]),
            'gnn_adam_step': (C.c_int, [vp, vp, vp, vp, sz, f32, f32, f32, f32, i32, vp, vp]),
            'gnn_adam_multi': (C.c_int, [vp, vp, vp, vp, vp, i32, f32, f32, f32, f32, i32, vp, vp]),
            'gnn_sgd_step': (C.c_int, [vp, vp, vp, sz, f32, f32, vp, vp]),
            'gnn_converged_gated': (C.c_int, [vp, vp, i32, i32, i32, f32, vp, vp, vp, f32, vp]),
            'gnn_aggregate_gated': (C.c_int, [C.POINTER(CSR), vp, i32, i32, vp, i32, vp, vp]),
            'gnn_train_workspace_bytes': (sz, [C.POINTER(TrainArgs)]),
            'gnn_train_step': (C.c_int, [C.POINTER(TrainArgs)]),
            'gnn_ragged_copy': (C.c_int, [vp, i32, vp, i32, vp]),
            'gnn_comm_unique_id': (C.c_int, [vp]),
            'gnn_comm_create': (C.c_int, [i32, i32, vp, C.POINTER(vp)]),
            'gnn_comm_destroy': (C.c_int, [vp]),
            'gnn_shard_loop': (C.c_int, [C.POINTER(ShardLoopArgs)]),
        }
        for name, (res, args) in protos.items():
            fn = getattr(l, name)
            fn.restype, fn.argtypes = res, args
        l.gnn_struct_size.restype = C.c_size_t
        l.gnn_struct_size.argtypes = [C.c_int]
        if l.gnn_abi_version() != GNN_ABI_VERSION:
            raise NativeError('libgnnloop.so ABI version mismatch: rebuild it')
        if (l.gnn_struct_size(0), l.gnn_struct_size(1), l.gnn_struct_size(2), l.gnn_struct_size(3)) != \
                (C.sizeof(CSR), C.sizeof(MLP), C.sizeof(LoopArgs), LoopArgs.flags.offset) or \
                (l.gnn_struct_size(4), l.gnn_struct_size(5), l.gnn_struct_size(6), l.gnn_struct_size(7)) != \
                (C.sizeof(TrainArgs), TrainArgs.tape.offset, C.sizeof(RaggedDesc), C.sizeof(ShardLoopArgs)):
            raise NativeError('ctypes struct layout does not match libgnnloop.so: rebuild it')
        _lib = l
    return _lib


def check(rc: int):
    if rc != 0:
        raise NativeError(lib().gnn_last_error().decode())


def require_device(t: torch.Tensor, name: str):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise NativeError(f'{name} must be a tensor on a HIP device: the message-passing loop has no CPU path')


def ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def current_stream(device) -> C.c_void_p:
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def make_csr(d: dict | None) -> CSR:
    """ctypes view of `SparseMatrix.device_csr()`; tensors must be kept alive by the caller."""
    c = CSR()
    if d is None:
        return c
    c.n_dst, c.n_src, c.nnz = d['n_dst'], d['n_src'], d['nnz']
    c.rowptr, c.src, c.w, c.row_scale = ptr(d['rowptr']), ptr(d['src']), ptr(d['w']), ptr(d['row_scale'])
    return c
