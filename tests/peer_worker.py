"""One rank of tests/test_gpu_peer.py: two PROCESSES share the one GPU, each owns half of the node range and runs the sharded loop with the
exchange in the iteration kernel's epilogue (peer stores into the other process' IPC-mapped state buffers, gnn_shard_iteration_peers).  Rank 0
also runs both ranks' shards in its own process with the exchange as plain slice copies (the harness of tests/test_gpu_round5.py) and compares:
the same k, the same state bits, the same output bits.  The process group is gloo: only IPC handles and results travel over it."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.distributed as dist


def main():
    rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.cuda.set_device(0)
    from gnnkeras_amd import _native as nat
    from gnnkeras_amd.distributed import ShardedLoop, partition
    from gnnkeras_amd.synth import er_graph_slice
    from gnnkeras_amd.Models.GNN import GNNnodeBased
    from gnnkeras_amd.Models.MLP import MLP, get_inout_dims
    N, E, d, K = 40_009, 320_000, int(os.environ.get('PEER_D', '64')), 6
    thr = float(os.environ.get('PEER_THR', '0.0'))
    inp, lay = get_inout_dims('state', 14, 3, 2, 'n', d)
    ns = MLP(inp[0], lay, 'selu', 'lecun_normal', 'lecun_normal', rng=0)
    ns.set_weights([a * 0.25 if a.ndim == 2 else a for a in ns.get_weights()])
    inp, lay = get_inout_dims('output', 14, 3, 2, 'n', d)
    no = MLP(inp[0], lay, 'softmax', 'glorot_normal', 'glorot_normal', rng=1)
    model = GNNnodeBased(ns, no, d, K, thr)
    s0 = torch.from_numpy(np.random.default_rng(1).normal(0, 0.1, (N, d)).astype(np.float32)).cuda()
    ranges = partition(N, world)[1]
    mine = er_graph_slice(N, E, *ranges[rank], aggregation_mode='average', seed=23)
    sl = ShardedLoop(model, mine, rank, world, 'cuda', overlap=False)
    sl.enable_peer_exchange()
    outs = []
    for rep in range(3):                      # (the arrival words keep growing across forwards: nothing is reset)
        k, st, o = sl.forward(s0)
        torch.cuda.synchronize()
        outs.append((float(k), st.cpu().numpy().copy(), o.cpu().numpy().copy()))
    name = nat.lib().gnn_last_kernel_name().decode()
    gathered = [None] * world
    dist.all_gather_object(gathered, (outs, name))
    sl.close()
    ok = True
    if rank == 0:
        # the reference: every rank's shard in THIS process, the exchange as slice copies
        shards = [ShardedLoop(model, er_graph_slice(N, E, *ranges[r], aggregation_mode='average', seed=23), r, world, 'cuda', overlap=False) for r in range(world)]
        for s_ in shards:
            s_._load_state0(s0); s_._setup(); s_._initial_flags()
        n = shards[0].plan.rows_per_slice * shards[0].SP
        for it in range(K):
            for s_ in shards: s_._iteration(it)
            for r, src in enumerate(shards):
                piece = src.buf[(it + 1) & 1].view(-1)[r * n:(r + 1) * n]
                for dst in shards:
                    if dst is not src: dst.buf[(it + 1) & 1].view(-1)[r * n:(r + 1) * n].copy_(piece)
        ref = [s_._output() for s_ in shards]
        torch.cuda.synchronize()
        for r in range(world):
            k_ref, st_ref, o_ref = float(ref[r][0]), ref[r][1].cpu().numpy(), ref[r][2].cpu().numpy()
            outs_r, name_r = gathered[r]
            assert 'peer stores' in name_r, name_r
            for rep, (k, st, o) in enumerate(outs_r):
                assert k == k_ref and k > 0, (r, rep, k, k_ref)
                assert np.array_equal(st, st_ref) and np.array_equal(o, o_ref), (r, rep, float(np.abs(st - st_ref).max()))
        print(f'PEER_OK world={world} d={d} thr={thr} k={float(ref[0][0]):g} kernel={gathered[0][1]}', flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
