"""N>1 path on CPU: world_size-2 gloo processes exercise the shard / broadcast / gather logic bench.py and
inference.py use on RCCL (no GPU, no HIP call)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_total, out_q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from mspi_amd import sharding as S
    from mspi_amd.backbones.resnet import ResNet
    torch.manual_seed(rank)                     # ranks start with DIFFERENT weights
    m = ResNet().eval()
    n = S.broadcast_weights(m, 0)
    ck = float(sum(p.double().sum() for p in m.parameters()))
    lo, hi = S.shard_bounds(n_total, rank, world)
    clips = torch.arange(n_total, dtype=torch.float32)[lo:hi]
    maps = clips.view(-1, 1, 1) * torch.ones(hi - lo, 3, 4)        # "map" of clip i is filled with i
    g = S.gather_maps(maps, n_total, 0)
    if rank == 0:
        out_q.put((n, ck, g))
    else:
        out_q.put((n, ck, None))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_total", [8, 7])
def test_two_rank_shard_broadcast_gather(n_total):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_total, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(2)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert res[0][0] == res[1][0] > 11e6                     # every float parameter + buffer went over the wire
    assert abs(res[0][1] - res[1][1]) < 1e-6                 # identical weights after the broadcast
    g = [r[2] for r in res if r[2] is not None][0]
    assert tuple(g.shape) == (n_total, 3, 4)
    assert torch.equal(g[:, 0, 0], torch.arange(n_total, dtype=torch.float32))   # global clip order restored


def test_shard_bounds_cover_everything():
    from mspi_amd.sharding import shard_bounds
    for n in (1, 7, 8, 64, 65):
        for w in (1, 2, 4, 8):
            b = [shard_bounds(n, r, w) for r in range(w)]
            assert b[0][0] == 0 and b[-1][1] == n and all(b[i][1] == b[i + 1][0] for i in range(w - 1))
            assert max(h - l for l, h in b) - min(h - l for l, h in b) <= 1
