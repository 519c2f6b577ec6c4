"""Per-iteration time of the state transition against graph size, one line per size: automatic choice and every pinned
kernel (2 = phase-alternating, 4 = wave-specialised, 5 = small whole-loop, 6 = mid-size whole-loop), d and arcs/node fixed."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gnnkeras_amd import _native as nat
from gnnkeras_amd.synth import er_graph
from gnnkeras_amd.Models.MLP import MLP, get_inout_dims
from gnnkeras_amd.Models.GNN import GNNnodeBased
from gnnkeras_amd.Sequencers.GraphSequencers import MultiGraphSequencer

d = int(sys.argv[1]) if len(sys.argv) > 1 else 64
sizes = [int(float(s)) for s in sys.argv[2].split(',')] if len(sys.argv) > 2 else [2000, 8000, 16000, 30000, 60000, 100000, 200000, 400000]
inp, lay = get_inout_dims('state', 14, 3, 2, 'n', d)
ns = MLP(inp[0], lay, 'selu', 'lecun_normal', 'lecun_normal', rng=0)
inp, lay = get_inout_dims('output', 14, 3, 2, 'n', d)
no = MLP(inp[0], lay, 'softmax', 'glorot_normal', 'glorot_normal', rng=1)
PINS = [(0, 'auto'), (nat.FLAG_FUSED_GEN2, 'gen2'), (nat.FLAG_FUSED_GEN4, 'gen4'), (nat.FLAG_FUSED_GEN5, 'small'), (nat.FLAG_FUSED_GEN6, 'mid')]
for N in sizes:
    E = 10 * N
    g = er_graph(N, E, aggregation_mode='average')
    x = MultiGraphSequencer([g], 'n', 'average', 1, shuffle=False)[0][0]
    s0 = torch.from_numpy(np.random.default_rng(1).normal(0, .1, (N, d)).astype(np.float32)).cuda()
    B_iter = E * (4 + 4 * d) + N * (4 + 12 * d)
    line = f'N={N:7d} E={E:8d} d={d}:'
    ref = None
    for flags, nm in PINS:
        def run(iters, reps=5):
            gnn = GNNnodeBased(ns, no, d, iters, 0.0); gnn.native_flags = flags
            inputs = gnn.process_inputs(x)
            ts = []
            for r in range(reps + 1):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); k, st, o = gnn.Loop(*inputs, state0=s0); e1.record(); torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1))
            return min(ts[1:]), float(k), st, nat.lib().gnn_last_kernel_name().decode()
        t10, _, _, _ = run(10); t50, k, st, kn = run(50)
        it = (t50 - t10) / 40
        if ref is None: ref = st
        err = float((st - ref).abs().max() / ref.abs().max())
        line += f'  {nm} {it*1e3:6.1f} us ({B_iter/it/8e9*100/1e3*1e3:4.1f}%, {kn.split("<")[0][8:]}, k={k:.0f}, d={err:.0e})'
    print(line, flush=True)
