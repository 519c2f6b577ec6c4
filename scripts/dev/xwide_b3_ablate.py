"""Ablations of k_state_xwide_b3 on the bench's d = 200 point (an -DXB_EXPERIMENT build: GNNKERAS_AMD_LIB=.../libgnnloop_xbexp.so).
GNN_XB_DBG bits: 1 no matrix instructions, 2 no weight loads, 4 no split at deposit, 8 matrix waves only hand the rows back."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, bench
from gnnkeras_amd import _native as nat
from gnnkeras_amd.synth import er_device_batch
from gnnkeras_amd.Models.GNN import GNNnodeBased
dev = torch.device('cuda:0')
N, E, d = 300_000, 3_000_000, int(sys.argv[1]) if len(sys.argv) > 1 else 200
x = er_device_batch(N, E, dev, aggregation_mode='average', seed=77)
ns, no = bench.starter_nets(d, dev)
gen = torch.Generator(device=dev); gen.manual_seed(2)
s0 = torch.randn((N, d), generator=gen, device=dev) * 0.1
gnn = GNNnodeBased(ns, no, d, 20, 0.0)
inputs = gnn.process_inputs(x)
b_iter = bench.algorithmic_bytes_per_iteration(N, E, d, ns.units[0], False)
for mw in sys.argv[2].split(',') if len(sys.argv) > 2 else ['0']:
    os.environ['GNN_XWIDE_MW'] = mw
    for dbg in [int(v) for v in (sys.argv[4].split(",") if len(sys.argv) > 4 else "0,1,2,3,4,8,12".split(","))]:
        os.environ['GNN_XB_DBG'] = str(dbg)
        el, k, t_iter = bench.measure_loop(gnn, inputs, s0, steps=2, warmup=1)
        print(f"d={d} mw={mw} dbg={dbg:2d}: {1e6 * t_iter:.1f} us per iteration, {b_iter / t_iter / 8e12:.3f}  [{nat.lib().gnn_last_kernel_name().decode()}]", flush=True)
