"""The C-ABI library loads and exports every symbol include/gnnloop.h declares; struct layouts of the ctypes binding
match the compiled ones; argument validation fails with a status + message and never touches the GPU."""
import ctypes as C
import os
import re

import pytest

from gnnkeras_amd import _native as nat

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def lib():
    if not os.path.exists(nat.LIB_PATH):
        nat.build()
    return nat.lib()


def declared_functions():
    text = open(os.path.join(ROOT, 'include', 'gnnloop.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(gnn_[a-z_0-9]+)\s*\(', text)))


def test_every_declared_symbol_is_exported(lib):
    names = declared_functions()
    assert set(names) == set(nat.EXPORTS), (names, nat.EXPORTS)
    for n in names:
        assert getattr(lib, n) is not None


def test_struct_layout_matches(lib):
    assert lib.gnn_abi_version() == nat.GNN_ABI_VERSION
    assert lib.gnn_struct_size(0) == C.sizeof(nat.CSR)
    assert lib.gnn_struct_size(1) == C.sizeof(nat.MLP)
    assert lib.gnn_struct_size(2) == C.sizeof(nat.LoopArgs)
    assert lib.gnn_struct_size(3) == nat.LoopArgs.flags.offset


def _args():
    a = nat.LoopArgs()
    a.abi_version = nat.GNN_ABI_VERSION
    a.n_nodes, a.n_arcs, a.dim_node_label, a.dim_arc_label = 10, 20, 3, 2
    a.state_dim, a.max_iteration, a.state_threshold = 4, 5, 0.01
    a.n_types = 1
    m = a.net_state[0]
    m.in_dim, m.n_layers = 2 * 4 + 2 * 3 + 2, 1
    m.units[0], m.activation[0] = 4, 2
    o = a.net_output
    o.in_dim, o.n_layers = 4 + 3, 1
    o.units[0], o.activation[0] = 2, 7
    a.n_out = 10
    return a


def test_workspace_size_and_validation_without_gpu(lib):
    a = _args()
    n = lib.gnn_loop_workspace_bytes(C.byref(a))
    assert n > 0 and n % 256 == 0
    a.max_iteration = 500
    assert lib.gnn_loop_workspace_bytes(C.byref(a)) >= n            # grows with the flag array only
    bad = _args(); bad.abi_version = 99
    assert lib.gnn_loop_workspace_bytes(C.byref(bad)) == 0 and b'abi_version' in lib.gnn_last_error()
    bad = _args(); bad.net_state[0].in_dim = 7
    assert lib.gnn_loop_workspace_bytes(C.byref(bad)) == 0 and b'in_dim' in lib.gnn_last_error()
    bad = _args(); bad.net_state[0].units[0] = 9                     # state net must map back to the state width
    assert lib.gnn_loop_workspace_bytes(C.byref(bad)) == 0 and b'output width' in lib.gnn_last_error()
    bad = _args(); bad.state_threshold = -1.0
    assert lib.gnn_loop_workspace_bytes(C.byref(bad)) == 0
    bad = _args(); bad.focus = 5
    assert lib.gnn_loop_workspace_bytes(C.byref(bad)) == 0 and b'focus' in lib.gnn_last_error()
    bad = _args(); bad.composite = 1; bad.n_types = 1; bad.max_iteration = 0
    assert lib.gnn_loop_workspace_bytes(C.byref(bad)) == 0 and b'max_iteration' in lib.gnn_last_error()
    # forward with NULL pointers is rejected before any HIP call
    assert lib.gnn_loop_forward(None) != 0
    a = _args()
    assert lib.gnn_loop_forward(C.byref(a)) != 0 and len(lib.gnn_last_error()) > 0
    assert lib.gnn_aggregate(None, None, 0, 0, None, 0, None) != 0
    assert lib.gnn_mlp_forward(None, None, 0, 0, None, 0, None, 0, None) != 0
    assert lib.gnn_converged(None, None, 4, 0, 0, 0.0, None, None) != 0


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    monkeypatch.setattr(nat, '_lib', None)
    monkeypatch.setattr(nat, 'LIB_PATH', str(tmp_path / 'nope.so'))
    with pytest.raises(nat.NativeError):
        nat.lib()
