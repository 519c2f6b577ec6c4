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


def test_group_and_set_tables_are_validated_without_gpu(lib):
    """Host arrays of the convergence groups / group sets (ABI 4): every malformed table is refused by the plan (sizes only, no
    GPU); gnn_loop_groups_supported answers 0 instead of reading past them."""
    def with_groups(begin, sets=None):
        a = _args()
        gb = (C.c_int32 * len(begin))(*begin)
        a.group_node_begin, a.n_groups = C.cast(gb, C.c_void_p), len(begin) - 1
        keep = [gb]
        if sets is not None:
            sb = (C.c_int32 * len(sets))(*sets)
            a.group_set_begin, a.n_group_sets = C.cast(sb, C.c_void_p), len(sets) - 1
            keep.append(sb)
        return a, keep
    a, keep = with_groups([0, 4, 10])
    assert lib.gnn_loop_workspace_bytes(C.byref(a)) > 0
    a, keep = with_groups([0, 4, 10], [0, 2])                    # both groups one set
    assert lib.gnn_loop_workspace_bytes(C.byref(a)) > 0
    for begin, sets, word in (([0, 4, 9], None, b'span'), ([0, 0, 10], None, b'empty'), ([1, 4, 10], None, b'span'),
                              ([0, 4, 10], [0, 1], b'group_set_begin'), ([0, 4, 10], [0, 0, 2], b'empty'), ([0, 4, 10], [1, 2], b'group_set_begin')):
        a, keep = with_groups(begin, sets)
        assert lib.gnn_loop_workspace_bytes(C.byref(a)) == 0 and word in lib.gnn_last_error(), (begin, sets, lib.gnn_last_error())
        assert lib.gnn_loop_groups_supported(C.byref(a)) == 0


def test_validation_paths_under_address_and_ub_sanitizers():
    """The host side of the library (argument validation, plan / workspace carving, table handling) built with
    -fsanitize=address,undefined (`make asan`: host code only, the device code is compiled as usual) and driven by the tests of
    this file in a child process - no GPU needed, SURVEY 5.  GPU AddressSanitizer is not available on this pool."""
    import subprocess, sys, glob
    asan_lib = os.path.join(nat.CSRC, 'libgnnloop_asan.so')
    # (`make` rebuilds it only when a source is newer: build() no longer does, and a stale sanitizer build lacks new exports)
    res = subprocess.run(['make', '-C', nat.CSRC, 'asan'], capture_output=True, text=True, timeout=1500)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-2000:]
    rt = glob.glob('/opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so')
    if not rt: pytest.skip('no clang AddressSanitizer runtime in this image')
    env = dict(os.environ, GNNKERAS_AMD_LIB=asan_lib, LD_PRELOAD=rt[0], ASAN_OPTIONS='detect_leaks=0:verify_asan_link_order=0:halt_on_error=1',
               UBSAN_OPTIONS='halt_on_error=1:print_stacktrace=1')
    res = subprocess.run([sys.executable, '-m', 'pytest', os.path.abspath(__file__), '-q', '-x', '-k', 'not sanitizers and not missing_library'],
                         capture_output=True, text=True, env=env, cwd=ROOT, timeout=900)
    assert res.returncode == 0 and ' passed' in res.stdout, res.stdout[-3000:] + res.stderr[-3000:]
    assert 'ERROR: AddressSanitizer' not in res.stderr and 'runtime error' not in res.stderr, res.stderr[-3000:]


def test_kernels_with_hand_counted_waits_use_no_scratch(tmp_path):
    """k_train_wgrad_b6 / k_train_wgrad_dx_b6 wait for their LDS-DMA loads with `s_waitcnt vmcnt(N)` counts written by hand (N = the loads of
    one ring slot: csrc/kernels_train_big.hpp).  A register spill is a scratch load / store on the same counter and would make those counts
    wrong without any test on a small input noticing: every instantiation in the built library must have no scratch segment and no spill."""
    import re, shutil, subprocess
    llvm = '/opt/rocm/lib/llvm/bin'
    if not all(os.path.exists(os.path.join(llvm, t)) for t in ('llvm-objcopy', 'clang-offload-bundler', 'llvm-readelf')):
        pytest.skip('ROCm llvm tools not found')
    so = nat.LIB_PATH
    fat, co = str(tmp_path / 'fat.bin'), str(tmp_path / 'dev.co')
    subprocess.run([os.path.join(llvm, 'llvm-objcopy'), '--dump-section', f'.hip_fatbin={fat}', so, str(tmp_path / 'unused.so')], check=True, capture_output=True)
    subprocess.run([os.path.join(llvm, 'clang-offload-bundler'), '--unbundle', '--type=o', f'--input={fat}', '--targets=hipv4-amdgcn-amd-amdhsa--gfx950',
                    f'--output={co}'], check=True, capture_output=True)
    notes = subprocess.run([os.path.join(llvm, 'llvm-readelf'), '--notes', co], check=True, capture_output=True, text=True).stdout
    seen = 0
    for m in re.finditer(r'\.name:\s+(\S*(?:k_train_wgrad_b6|k_train_wgrad_dx_b6)\S*)\n(.*?)\.wavefront_size', notes, re.S):
        blk = m.group(2)
        scratch = int(re.search(r'\.private_segment_fixed_size:\s+(\d+)', blk).group(1))
        spills = int(re.search(r'\.vgpr_spill_count:\s+(\d+)', blk).group(1)) + int(re.search(r'\.sgpr_spill_count:\s+(\d+)', blk).group(1))
        dyn = re.search(r'\.uses_dynamic_stack:\s+(\w+)', blk).group(1)
        assert scratch == 0 and spills == 0 and dyn == 'false', (m.group(1), scratch, spills, dyn)
        seen += 1
    assert seen >= 16, seen          # 2 widths x 7 activations + the one-pass kernel's 2 widths
