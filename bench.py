#!/usr/bin/env python3
"""bench.py — the headline measurement of BASELINE.json on MI355X.

    python bench.py --gpus N --steps K --warmup W            (N > 1: launched by torch.distributed.run, one rank per GPU)

Workload (config.workload): BASELINE config C4 — synthetic directed Erdős–Rényi graph, 1 000 000 nodes / 10 000 000
arcs, state_dim 64, node-focused, 'average' aggregation, max_iteration 50, threshold 0 (fixed work, k = 50), starter
networks (BN + Dense(159->64, selu) state net, BN + Dense(78->2, softmax) output net), weights default_rng(0/1),
state0 = default_rng(1).normal(0, 0.1).  One *step* = one whole forward pass `Loop(...)` of that graph: setup
aggregates + 50 fused iterations + output network, inputs already resident in HBM.

metric  = node-state updates/s = arcs x iterations / wall seconds (whole job, all ranks), float32 arithmetic.
roofline = the fused iteration kernel: algorithmic bytes per launch / its average duration,
           measured live with HIP events recorded on the launch stream around the 50 iteration launches.
cpu_baseline = the torch-CPU restatement of the reference's un-fused op sequence (oracle/torch_cpu.py, all host
           cores), timed on rank 0 at N = 1 on a bounded sample (a few iterations of the same graph).
Also reported (extra keys): the MUTAG batch-32 forward (BASELINE config C2) in ms/graph next to its CPU baseline.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# predict() / evaluate() overlap independent batches on up to 8 HIP streams; by default HIP multiplexes all streams of
# a process onto 4 hardware queues.  Must be set before the HIP runtime starts (importing torch does that).
os.environ.setdefault('GPU_MAX_HW_QUEUES', '16')

import numpy as np
import torch

HBM_PEAK = 8.0e12          # B/s, MI355X_MICROARCH.md chip-level parameters (spec)


def starter_nets(d, device, focus='n'):
    from gnnkeras_amd.Models.MLP import MLP, get_inout_dims
    inp, lay = get_inout_dims('state', 14, 3, 2, focus, d)
    ns = MLP(inp[0], lay, 'selu', 'lecun_normal', 'lecun_normal', rng=0, device=device)
    inp, lay = get_inout_dims('output', 14, 3, 2, focus, d)
    no = MLP(inp[0], lay, 'softmax', 'glorot_normal', 'glorot_normal', rng=1, device=device)
    return ns, no


def algorithmic_bytes_per_iteration(n_nodes, n_arcs, d, h1, per_arc_weights):
    """E(4 + 4d [+4]) + N(4 + 4d + 4d + 4 H1): arc source id + gathered neighbour row (+ weight);
    per node row pointer, own state read, new state write, constant-term read."""
    return n_arcs * (4 + 4 * d + (4 if per_arc_weights else 0)) + n_nodes * (4 + 8 * d + 4 * h1)


def mutag_section(device, cpu: bool):
    """BASELINE C2: all 136 MUTAG batches of 32 graphs, d = 32, max_iteration = 50, threshold 0.01."""
    from gnnkeras_amd.load_MUTAG import load_graphs
    from gnnkeras_amd.Models.GNN import GNNgraphBased
    from gnnkeras_amd.Sequencers.GraphSequencers import MultiGraphSequencer
    graphs = load_graphs()
    seq = MultiGraphSequencer(graphs, 'g', 'average', 32, shuffle=False, device=device)
    ns, no = starter_nets(32, device, 'g')
    gnn = GNNgraphBased(ns, no, 32, 50, 0.01)
    items = [seq[i][0] for i in range(len(seq))]
    inputs = [gnn.process_inputs(x) for x in items]
    rng = np.random.default_rng(1)
    s0s = [torch.from_numpy(rng.normal(0, 0.1, (x[0].shape[0], 32)).astype(np.float32)).to(device) for x in items]
    for inp, s0 in zip(inputs, s0s): gnn.Loop(*inp, state0=s0)              # warm-up: one untimed pass
    torch.cuda.synchronize()
    ks = []
    t0 = time.perf_counter()
    for inp, s0 in zip(inputs, s0s):
        k, st, out = gnn.Loop(*inp, state0=s0)
        ks.append(k)
    torch.cuda.synchronize()
    t_one = time.perf_counter() - t0                                    # one stream: a batch's latency, 136 times
    # the way predict() / evaluate() walk a sequencer: independent batches side by side on a few HIP streams (a batch
    # keeps ~16 of the 256 CUs busy)
    width = gnn._round_width(seq, device)
    run = lambda i: gnn.Loop(*inputs[i], state0=s0s[i])
    for _ in gnn._batches_concurrently(len(inputs), run, device, width): pass
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ks = [r[0] for _, r in gnn._batches_concurrently(len(inputs), run, device, width)]
    torch.cuda.synchronize()
    t_gpu = time.perf_counter() - t0
    ks = [float(k) for k in ks]
    n_graphs = len(graphs)
    arcs_iters = sum(x[1].shape[0] * k for x, k in zip(items, ks))
    res = {'workload': 'MUTAG (TU Mutagenicity) 4337 graphs as 136 batches of 32, state_dim=32, max_iteration=50, '
                       'threshold=0.01, graph-focused forward',
           'fwd_ms_per_graph': 1e3 * t_gpu / n_graphs, 'fwd_ms_per_batch': 1e3 * t_gpu / len(items),
           'concurrent_batches': width,
           'one_stream_fwd_ms_per_graph': 1e3 * t_one / n_graphs, 'one_stream_fwd_ms_per_batch': 1e3 * t_one / len(items),
           'us_per_iteration': 1e6 * t_one / max(sum(ks), 1), 'mean_k': float(np.mean(ks)),
           'updates_per_s': arcs_iters / t_gpu}
    if cpu:
        from oracle import torch_cpu
        from oracle.harness import _np, _triple
        sample = list(range(0, len(items), 17))[:8]
        t_cpu = 0.0
        for i in sample:
            x = items[i]
            t1 = time.perf_counter()
            torch_cpu.loop(_np(x[0]), _np(x[1]), _triple(x[5]), _triple(x[6]), _triple(x[7]),
                           np.ones(x[0].shape[0], bool), net_state=ns.spec(), net_output=no.spec(), state_vect_dim=32,
                           max_iteration=50, state_threshold=0.01, focus='g', state0=_np(s0s[i]))
            t_cpu += time.perf_counter() - t1
        cpu_ms_graph = 1e3 * t_cpu / (32 * len(sample))
        res['cpu_fwd_ms_per_graph'] = cpu_ms_graph
        res['speedup_vs_cpu'] = cpu_ms_graph / res['fwd_ms_per_graph']
        res['cpu_sample'] = f'{len(sample)} of 136 batches, torch CPU {torch.get_num_threads()} threads'
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--workload', choices=['c4', 'c3', 'c5'], default='c4',
                    help='c4: ER 1M/10M (headline); c3: ER 100k/1M; c5: composite ER 500k/5M, 3 node types')
    ap.add_argument('--nodes', type=float, default=None)
    ap.add_argument('--arcs', type=float, default=None)
    ap.add_argument('--state-dim', type=int, default=64)
    ap.add_argument('--max-iteration', type=int, default=50)
    ap.add_argument('--aggregation', default='average')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-mutag', action='store_true')
    ap.add_argument('--unfused', action='store_true')
    ap.add_argument('--force-sharded', action='store_true', help='run the N>1 code path even with one rank (smoke test)')
    ap.add_argument('--exchange', choices=['auto', 'allgather', 'halo'], default='auto',
                    help='N>1 state exchange: whole slices (all-gather) or compacted halos (all-to-all)')
    args = ap.parse_args()

    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world != args.gpus:
        raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}')
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X: the message-passing loop has no CPU path')
    torch.cuda.set_device(local_rank)
    device = torch.device('cuda', local_rank)
    sharded = world > 1 or args.force_sharded
    if sharded:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29511')
        dist.init_process_group('nccl', rank=rank, world_size=world, device_id=device)

    from gnnkeras_amd import _native as nat
    from gnnkeras_amd.synth import er_graph, er_composite_graph
    from gnnkeras_amd.Models.GNN import GNNnodeBased
    from gnnkeras_amd.Models.CompositeGNN import CompositeGNNnodeBased
    from gnnkeras_amd.Models.MLP import MLP, get_inout_dims
    from gnnkeras_amd.Sequencers.GraphSequencers import MultiGraphSequencer, CompositeMultiGraphSequencer

    sizes = {'c4': (1e6, 1e7), 'c3': (1e5, 1e6), 'c5': (5e5, 5e6)}[args.workload]
    N, E = int(args.nodes or sizes[0]), int(args.arcs or sizes[1])
    d, K_it = args.state_dim, args.max_iteration
    s0_host = np.random.default_rng(1).normal(0, 0.1, (N, d)).astype(np.float32)
    composite = args.workload == 'c5'
    if composite:
        dims = (14, 8, 4)
        graph = er_composite_graph(N, E, dim_node_label=dims, aggregation_mode=args.aggregation, seed=1234)
        inp, lay = get_inout_dims('state', dims, 3, 2, 'n', d)
        nets_s = [MLP(i, lay, 'selu', 'lecun_normal', 'lecun_normal', rng=t, device=device) for t, i in enumerate(inp)]
        inp, lay = get_inout_dims('output', dims, 3, 2, 'n', d)
        no = MLP(inp[0], lay, 'softmax', 'glorot_normal', 'glorot_normal', rng=9, device=device)
        ns = nets_s[0]
        gnn = CompositeGNNnodeBased(nets_s, no, d, K_it, 0.0)
        Sequencer = CompositeMultiGraphSequencer
    else:
        graph = er_graph(N, E, aggregation_mode=args.aggregation, seed=1234)
        ns, no = starter_nets(d, device)
        gnn = GNNnodeBased(ns, no, d, K_it, 0.0)
        Sequencer = MultiGraphSequencer
    if args.unfused: gnn.native_flags = nat.FLAG_UNFUSED

    if not sharded:
        seq = Sequencer([graph], 'n', args.aggregation, 1, shuffle=False, device=device)
        x = seq[0][0]
        inputs = gnn.process_inputs(x)
        s0 = torch.from_numpy(s0_host).to(device)
        ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        for e in ev: e.record()
        gnn.loop_events = ev
        step = lambda: gnn.Loop(*inputs, state0=s0)
        per_arc_w = inputs[7 if composite else 5].csr().w is not None
        sync_all = torch.cuda.synchronize
    else:
        import torch.distributed as dist
        from gnnkeras_amd.distributed import make_sharded_loop
        sl = make_sharded_loop(gnn, graph, rank=rank, world_size=world, device=device, exchange=args.exchange)
        s0 = torch.from_numpy(s0_host).to(device)
        step = lambda: sl.forward(s0)
        per_arc_w = sl.per_arc_weights

        def sync_all():
            torch.cuda.synchronize()
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup): step()
    loop_ms, ks = [], []
    sync_all()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        k, state, out = step()
    sync_all()
    elapsed = time.perf_counter() - t0
    if sharded:
        import torch.distributed as dist
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t)
    k_val = float(k)
    ms_per_step = 1e3 * elapsed / args.steps
    value = E * k_val * args.steps / elapsed

    # dominant kernel: fused iteration; duration from the HIP events the library records around the 50 launches
    if not sharded:
        t_loop_ms = []
        for _ in range(5):
            step(); torch.cuda.synchronize()
            t_loop_ms.append(ev[0].elapsed_time(ev[1]))
        t_iter = 1e-3 * float(np.median(t_loop_ms)) / max(k_val, 1)
    else:
        t_iter = sl.kernel_seconds_per_iteration(s0)
    h1 = ns.units[0]
    n_local = N if not sharded else sl.n_local
    e_local = E if not sharded else sl.e_local
    b_iter = algorithmic_bytes_per_iteration(n_local, e_local, d, h1, per_arc_w)
    achieved = b_iter / t_iter
    roofline = {'bound': 'hbm', 'achieved': achieved / 1e9, 'peak': HBM_PEAK / 1e9, 'unit': 'GB/s',
                'frac': achieved / HBM_PEAK, 'traffic': None,
                'kernel': ('k_state_fused4<64,false,4,4> (wave-specialised; GNN_FUSED_KERNEL=%s)' % os.environ.get('GNN_FUSED_KERNEL', 'auto'))
                          if not args.unfused else 'k_aggregate+k_segdense+k_converge',
                'algorithmic_bytes_per_launch': b_iter, 'avg_launch_us': 1e6 * t_iter}
    traffic_file = os.path.join(ROOT, 'profiles', 'hbm_traffic.json')
    if os.path.exists(traffic_file) and not sharded and not args.unfused:
        try:
            tr = json.load(open(traffic_file))
            if tr.get('workload_nodes') == N and tr.get('workload_arcs') == E and not composite:
                roofline['traffic'] = tr['hbm_bytes_per_launch']
        except Exception:
            pass

    result = {
        'metric': 'node-state updates/s (arcs x iterations / s), forward Loop',
        'value': value, 'unit': 'arc-updates/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': ms_per_step, 'higher_is_better': True, 'scaling': 'strong', 'vs_baseline': None,
        'dtype': 'f32', 'data': 'synthetic',
        'config': {'workload': f'{args.workload.upper()} Erdos-Renyi {N} nodes / {E} arcs, state_dim={d}, max_iteration={K_it}, '
                               f'threshold=0 (k={k_val:g}), node-focused, {args.aggregation} aggregation, '
                               + ('3 node types with per-type ' if composite else '') +
                               f'BN+Dense({ns.input_dim}->{h1},selu) state net',
                   'sharding': 'single GPU' if not sharded else f'node-range shards over {world} GPUs, '
                                                               f'RCCL {"all-to-all of compacted halos" if type(sl).__name__ == "HaloShardedLoop" else "all-gather of state slices"} per iteration'},
        'roofline': roofline,
        'loop_only_updates_per_s': E / t_iter if not sharded else None,
        'fwd_ms_per_graph': ms_per_step,
    }

    if rank == 0 and not sharded and not args.no_cpu_baseline and not composite:
        from oracle import torch_cpu
        from oracle.harness import _np, _triple
        it_cpu = 3
        tm = {}
        torch_cpu.loop(_np(x[0]), _np(x[1]), _triple(x[5]), _triple(x[6]), _triple(x[7]), np.ones(N, bool),
                       net_state=ns.spec(), net_output=no.spec(), state_vect_dim=d, max_iteration=it_cpu,
                       state_threshold=0.0, state0=s0_host, timings=tm)
        result['cpu_baseline'] = {'value': E * it_cpu / tm['loop_s'], 'unit': 'arc-updates/s',
                                  'cores': torch.get_num_threads(), 'kind': 'port',
                                  'sample': f'{it_cpu} iterations of the same C4 graph, loop only, torch-CPU '
                                            f'restatement of the TF op sequence (not TensorFlow), '
                                            f'{os.cpu_count()} host cpus'}
        result['speedup_vs_cpu_loop'] = (E / t_iter) / result['cpu_baseline']['value']
    if rank == 0 and not sharded and not args.no_mutag and args.workload == 'c4':
        result['mutag'] = mutag_section(device, cpu=not args.no_cpu_baseline)

    if rank == 0:
        print(json.dumps(result))
    if sharded:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
