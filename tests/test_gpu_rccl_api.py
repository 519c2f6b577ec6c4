"""The two collectives of the sharded loop, called exactly the way gnnkeras_amd/distributed.py calls them, on the real
backend ("nccl" = RCCL) with the only world size one GPU allows (1): an in-place `all_gather_into_tensor` whose input is
the rank's slice of the output, and an `all_to_all_single` with explicit split sizes into a slice of the state buffer.
A wrong dtype / aliasing / split convention is rejected by the backend here, without needing a second GPU; the data
movement between ranks is covered by the gloo tests (tests/test_distributed.py) and the emulated shards
(tests/test_gpu_parity.py)."""
import os
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = textwrap.dedent('''
    import os, torch, torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
    SP, rows = 64, 1001                       # a slice of 1000 state rows + its flag row
    buf = torch.arange(rows * SP, dtype=torch.float32, device='cuda').view(rows, SP).clone()
    want = buf.clone()
    flat = buf.view(-1)
    n = rows * SP
    dist.all_gather_into_tensor(flat, flat[0 * n:1 * n])                 # ShardedLoop._exchange
    torch.cuda.synchronize(); assert torch.equal(buf, want)
    send = torch.randn(300, SP, device='cuda')
    recv = buf[100:400].view(-1)                                         # HaloShardedLoop._exchange: into a slice of buf
    dist.all_to_all_single(recv, send.view(-1), output_split_sizes=[300 * SP], input_split_sizes=[300 * SP])
    torch.cuda.synchronize(); assert torch.equal(buf[100:400], send) and torch.equal(buf[:100], want[:100])
    t = torch.tensor([1.5], dtype=torch.float64, device='cuda'); dist.all_reduce(t, op=dist.ReduceOp.MAX)   # bench.py timing
    dist.barrier(); torch.cuda.synchronize()
    dist.destroy_process_group()
    print('rccl api ok')
''')


@pytest.mark.gpu
def test_rccl_accepts_the_sharded_loops_collectives(tmp_path):
    script = tmp_path / 'rccl_api.py'
    script.write_text(SCRIPT)
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT='29533', HSA_ENABLE_IPC_MODE_LEGACY='0')
    res = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=600, env=env)
    assert res.returncode == 0 and 'rccl api ok' in res.stdout, res.stdout[-2000:] + res.stderr[-3000:]
