"""Phase totals of k_state_xwide_b3 (library built with -DXW_PROFILE -DXB_EXPERIMENT, given as GNNKERAS_AMD_LIB): python scripts/dev/xb_prof.py [d] [dbg,...]"""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, bench
from gnnkeras_amd import _native as nat
from gnnkeras_amd.synth import er_device_batch
from gnnkeras_amd.Models.GNN import GNNnodeBased
dev = torch.device('cuda:0')
N, E, d, K = 300_000, 3_000_000, int(sys.argv[1]) if len(sys.argv) > 1 else 200, 20
x = er_device_batch(N, E, dev, aggregation_mode='average', seed=77)
ns, no = bench.starter_nets(d, dev)
gen = torch.Generator(device=dev); gen.manual_seed(2)
s0 = torch.randn((N, d), generator=gen, device=dev) * 0.1
gnn = GNNnodeBased(ns, no, d, K, 0.0)
inputs = gnn.process_inputs(x)
lib = nat.lib()
out = (ctypes.c_ulonglong * 8)()
for dbg in (sys.argv[2].split(',') if len(sys.argv) > 2 else ['0']):
    os.environ['GNN_XB_DBG'] = dbg
    k, st, o = gnn.Loop(*inputs, state0=s0); torch.cuda.synchronize()
    lib.gnn_xw_profile(out, 1)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); k, st, o = gnn.Loop(*inputs, state0=s0); e1.record(); torch.cuda.synchronize()
    lib.gnn_xw_profile(out, 1)
    v = list(out); wg = 256 * K; tiles = max(v[3], 1)
    print(f'dbg {dbg}: loop {e0.elapsed_time(e1) / K * 1e3:.0f} us per iteration (instrumented); matrix wave 0, s_memtime ticks per tile: wait for the rows {v[0] / tiles:.0f}; '
          f'K loop {v[1] / tiles:.0f}; constant + epilogue {v[2] / tiles:.0f}; tiles per workgroup and launch {tiles / wg:.1f}; per launch and workgroup: matrix wave 0 {v[6] / wg:.0f}, '
          f'first gather wave {v[5] / wg:.0f} of which waiting for a free half-slot {v[4] / wg:.0f}', flush=True)
