"""One C3 variant of the wave-specialised kernel (library built with -DGNN_F4_EXPERIMENT: `make -C gnnkeras_amd/csrc c3exp`, loaded through
GNNKERAS_AMD_LIB; GNN_F4_VARIANT selects it): the launch time from HIP events over 50 iterations x 5 forwards, a checksum of the state (every variant
sums a row's neighbours in ascending-source order: the bits must not move), and the per-workgroup time line of one launch."""
import sys, os, ctypes, zlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ['GNN_FUSED_KERNEL'] = '4'
import numpy as np, torch
from gnnkeras_amd import _native as nat
from gnnkeras_amd.synth import er_graph
from gnnkeras_amd.Models.MLP import MLP, get_inout_dims
from gnnkeras_amd.Models.GNN import GNNnodeBased
from gnnkeras_amd.Sequencers.GraphSequencers import MultiGraphSequencer
N = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100000
E, d, K = 10 * N, 64, 50
g = er_graph(N, E, aggregation_mode='average'); x = MultiGraphSequencer([g], 'n', 'average', 1, shuffle=False)[0][0]
inp, lay = get_inout_dims('state', 14, 3, 2, 'n', d); ns = MLP(inp[0], lay, 'selu', 'lecun_normal', 'lecun_normal', rng=0)
inp, lay = get_inout_dims('output', 14, 3, 2, 'n', d); no = MLP(inp[0], lay, 'softmax', 'glorot_normal', 'glorot_normal', rng=1)
s0 = torch.from_numpy(np.random.default_rng(1).normal(0, .1, (N, d)).astype(np.float32)).cuda()
gnn = GNNnodeBased(ns, no, d, K, 0.0); inputs = gnn.process_inputs(x)
ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
for e in ev: e.record()
gnn.loop_events = ev
ts = []
for rep in range(7):
    k, st, out = gnn.Loop(*inputs, state0=s0); torch.cuda.synchronize()
    if rep >= 2: ts.append(1e3 * ev[0].elapsed_time(ev[1]) / K)
name = nat.lib().gnn_last_kernel_name().decode()
crc = zlib.crc32(st.cpu().numpy().tobytes())
L = nat.lib()
line = f'variant {os.environ.get("GNN_F4_VARIANT", "0")}: {name}: {np.median(ts):.2f} us per launch (min {min(ts):.2f}, max {max(ts):.2f}; 5 forwards x {K} launches), k = {float(k):g}, state crc32 {crc:08x}'
try:
    L.gnn_f4_wg_times.argtypes = [ctypes.c_void_p]
    buf = (ctypes.c_ulonglong * 4096)()
    assert L.gnn_f4_wg_times(buf) == 0
    t = np.array(list(buf), dtype=np.float64).reshape(1024, 4); t = t[t[:, 3] > 0]
    rel = (t - t[:, 0].min()) * 0.01
    ent, fil, dep, ext = (rel[:, i] for i in range(4))
    line += (f'; {len(t)} workgroups: fill done p50 {np.median(fil):.1f} us, first deposit p50 {np.median(dep):.1f} / p90 {np.percentile(dep, 90):.1f}, '
             f'exit p50 {np.median(ext):.1f} / max {ext.max():.1f}')
except Exception as e:
    line += f' (no time line: {e})'
print(line, flush=True)
