"""N > 1 path on CPU: frame partition + the gather of decoded blocks, world_size 2 over gloo."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, total, width, q_out):
    sys.path.insert(0, ROOT)
    import importlib.util
    spec = importlib.util.spec_from_file_location("qshard", os.path.join(ROOT, "qcrypto-ldpc_amd", "shard.py"))
    sh = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(sh)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = sh.frame_range(total, world, rank)
    # a "decoded block" that is a pure function of the global frame id
    local = (torch.arange(lo, hi, dtype=torch.int32)[:, None] * 7 + torch.arange(width, dtype=torch.int32)[None, :])
    out = sh.gather_blocks(local, total, dst=0)
    # the bench's form: a receive buffer allocated once, the gather lands in it (a view of it when the shards are equal)
    buf = sh.gather_buffer(local, total, dst=0)
    out2 = sh.gather_blocks(local, total, dst=0, out=buf)
    out3 = sh.gather_blocks(local + 1, total, dst=0, out=buf)
    if rank == 0:
        exp = torch.arange(total, dtype=torch.int32)[:, None] * 7 + torch.arange(width, dtype=torch.int32)[None, :]
        ok = bool((out == exp).all()) and out.shape == exp.shape and bool((out3 == exp + 1).all()) and out3.shape == exp.shape
        if total % world == 0:
            ok = ok and out3.data_ptr() == buf.data_ptr()          # no allocation, no copy
        q_out.put(ok and out2.shape == exp.shape)
    else:
        assert out is None and buf is None and out2 is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("total", [8, 9, 3])
def test_gather_blocks_world2(total):
    ctx = mp.get_context("spawn")
    q_out = ctx.Queue()
    port = 29500 + (os.getpid() + total) % 500
    procs = [ctx.Process(target=_worker, args=(r, 2, port, total, 5, q_out)) for r in range(2)]
    for p in procs:
        p.start()
    ok = q_out.get(timeout=120)
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert ok


def test_frame_range_partitions_exactly():
    import importlib.util
    spec = importlib.util.spec_from_file_location("qshard", os.path.join(ROOT, "qcrypto-ldpc_amd", "shard.py"))
    sh = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(sh)
    for total in (0, 1, 7, 4096, 262144, 262145):
        for world in (1, 2, 4, 8):
            r = [sh.frame_range(total, world, k) for k in range(world)]
            assert r[0][0] == 0 and r[-1][1] == total
            assert all(r[k][1] == r[k + 1][0] for k in range(world - 1))
            assert max(h - l for l, h in r) - min(h - l for l, h in r) <= 1
