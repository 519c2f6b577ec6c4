"""k_state_xwide_b3 (widths 129 .. 256, kernel_state_xwide.hpp) against the un-fused path on small shapes, then time per iteration launch on the
bench's d = 200 point and two more widths.  (Round 6 ran it with the f32-matrix-instruction form of the kernel next to it - GNN_XWIDE_B3=0, since
deleted: profiles/r06_xwide_ablations.txt.)  usage: python scripts/dev/xwide_b3_check.py [quick]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import bench
from gnnkeras_amd import _native as nat
from gnnkeras_amd.synth import er_device_batch
from gnnkeras_amd.Models.GNN import GNNnodeBased

dev = torch.device('cuda:0')
def loop(N, E, d, K, mode='average', thr=0.0, seed=7):
    x = er_device_batch(N, E, dev, aggregation_mode=mode, seed=seed)
    ns, no = bench.starter_nets(d, dev)
    gen = torch.Generator(device=dev); gen.manual_seed(2)
    s0 = torch.randn((N, d), generator=gen, device=dev) * 0.1
    gnn = GNNnodeBased(ns, no, d, K, thr)
    return gnn, gnn.process_inputs(x), s0, ns

def rel(a, b): return float((a - b).abs().max() / b.abs().max())

for (N, E, d) in [(3_000, 30_000, 200), (20_011, 160_000, 160), (1_777, 40_000, 256), (33, 66, 192), (9_000, 180_000, 136), (5_000, 50_000, 129)]:
    gnn, inputs, s0, ns = loop(N, E, d, 6)
    out = {}
    for name, env, flags in (('b3', {}, 0), ('unfused', {}, nat.FLAG_UNFUSED)):
        os.environ.update(env)
        gnn.native_flags = flags
        k, st, o = gnn.Loop(*inputs, state0=s0); torch.cuda.synchronize()
        out[name] = (float(k), st.clone(), nat.lib().gnn_last_kernel_name().decode())
    print(f"N={N} E={E} d={d}: k {[v[0] for v in out.values()]}  fused vs un-fused {rel(out['b3'][1], out['unfused'][1]):.2e}  [{out['b3'][2]}]", flush=True)

if len(sys.argv) > 1 and sys.argv[1] == 'quick': sys.exit(0)
for (N, E, d) in [(300_000, 3_000_000, 200), (1_000_000, 10_000_000, 160), (200_000, 2_000_000, 256)]:
    gnn, inputs, s0, ns = loop(N, E, d, 20)
    b_iter = bench.algorithmic_bytes_per_iteration(N, E, d, ns.units[0], False)
    for env in ({},):
        os.environ.update(env)
        el, k, t_iter = bench.measure_loop(gnn, inputs, s0, steps=3, warmup=1)
        print(f"N={N} d={d} {env}: {1e6 * t_iter:.1f} us per iteration, {b_iter / t_iter / 8e12:.3f} of 8 TB/s  [{nat.lib().gnn_last_kernel_name().decode()}]", flush=True)
