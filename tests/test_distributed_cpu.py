"""N>1 path on CPU: two gloo ranks exchange cell buffers exactly as bench.py's ranks
exchange them over RCCL (one gather per frame to rank 0, then a scatter of cells into
the row-major image).  No rendering happens here (no GPU): each rank fabricates the
buffer it WOULD have rendered from a known image, so the test pins the partition /
padding / gather / de-interleave logic bit for bit."""
import os
import socket
import numpy as np
import pytest

import helpers  # noqa: F401


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, w, h, q):
    import torch
    import torch.distributed as dist
    from raylib_amd import tiling
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rng = np.random.RandomState(123)
    image = rng.rand(h, w, 4).astype(np.float32)           # every rank knows the truth
    pad = tiling.padded_cells(w, h, world) * 64
    mine = np.zeros((pad, 4), np.float32)
    cells = tiling.extract_cells(image, rank, world)
    mine[: len(cells)] = cells
    t = torch.from_numpy(mine)
    gathered = [torch.zeros_like(t) for _ in range(world)] if rank == 0 else None
    dist.gather(t, gathered, dst=0)
    ok = True
    if rank == 0:
        stacked = torch.stack(gathered).reshape(-1, 4)
        src, dst = tiling.torch_scatter_plan(w, h, world, "cpu")
        out = torch.zeros(h * w, 4)
        out[dst] = stacked[src]
        ok = bool(np.array_equal(out.numpy().reshape(h, w, 4), image))
        ok = ok and bool(np.array_equal(tiling.assemble(w, h, world, [g.numpy() for g in gathered]), image))
    # max-over-ranks timing reduction used by bench.py
    tt = torch.tensor([float(rank + 1)])
    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    ok = ok and tt.item() == float(world)
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, ok))


@pytest.mark.parametrize("shape", [(64, 64), (40, 28), (1920, 1080)])
def test_two_rank_gather_reassembles_image(shape):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, shape[0], shape[1], q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert all(ok for _, ok in res), res


def test_partition_covers_every_pixel_once():
    from raylib_amd import tiling
    for (w, h) in ((64, 64), (40, 28), (1920, 1080), (3840, 2160), (9, 7)):
        for world in (1, 2, 4, 8):
            seen = np.zeros(w * h, np.int32)
            for r in range(world):
                idx = tiling.pixel_index_map(w, h, r, world)
                np.add.at(seen, idx[idx >= 0], 1)
            assert (seen == 1).all()


@pytest.mark.parametrize("world", [3, 4, 8])
def test_scatter_plan_reassembles_for_the_world_sizes_the_driver_runs(world):
    """bench.py's rank-0 assembly (stack the gathered, padded buffers, one indexed copy) for 4 and 8 ranks and an odd count, without
    processes: the buffers are what the ranks would have sent."""
    import torch
    from raylib_amd import tiling
    for (w, h) in ((1920, 1080), (40, 28), (9, 7)):
        image = np.random.RandomState(world).rand(h, w, 4).astype(np.float32)
        pad = tiling.padded_cells(w, h, world) * 64
        bufs = []
        for r in range(world):
            mine = np.full((pad, 4), -1.0, np.float32)              # padding must never reach the frame
            cells = tiling.extract_cells(image, r, world)
            mine[: len(cells)] = cells
            bufs.append(torch.from_numpy(mine))
        src, dst = tiling.torch_scatter_plan(w, h, world, "cpu")
        out = torch.full((h * w, 4), -2.0)
        out[dst] = torch.stack(bufs).reshape(-1, 4)[src]
        assert np.array_equal(out.numpy().reshape(h, w, 4), image)
