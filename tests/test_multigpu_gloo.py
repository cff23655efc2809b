"""N>1 path on CPU: world_size-2 (and 3) gloo processes exercise the band partition + colour gather plumbing
that bench.py runs over RCCL.  Each rank fills its band from a known function of (x, y); rank 0 must
reassemble the exact full frame."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from softwarerenderer_amd import multigpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _frame(height, width):
    y, x = np.mgrid[0:height, 0:width]
    return np.stack([x * 1.0, y * 1.0, x * 1000.0 + y, np.ones_like(x) * 1.0], axis=-1).astype(np.float32)


def _worker(rank, world, port, height, width, ok):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    full = _frame(height, width)
    band = multigpu.band_partition(height, world)[rank]
    y0, rows = multigpu.band_pixel_rows(height, band)
    local = torch.zeros((multigpu.max_band_rows(height, world), width, 4), dtype=torch.float32)
    local[:rows] = torch.from_numpy(full[y0:y0 + rows])
    out = multigpu.gather_bands(local, height, width, rank, world, dst=0)
    if rank == 0:
        ok[0] = int(np.array_equal(out.numpy(), full))
    else:
        assert out is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,height,width", [(2, 100, 37), (2, 64, 16), (3, 333, 20), (2, 128, 24), (4, 256, 8)])
def test_band_gather_reassembles_the_frame(world, height, width):
    ok = mp.get_context("spawn").Array("i", [0])
    mp.spawn(_worker, args=(world, _free_port(), height, width, ok), nprocs=world, join=True)
    assert ok[0] == 1


def test_assemble_numpy_equals_concatenation():
    full = _frame(50, 8)
    bands = []
    for b in multigpu.band_partition(50, 3):
        y0, rows = multigpu.band_pixel_rows(50, b)
        pad = np.zeros((multigpu.max_band_rows(50, 3), 8, 4), dtype=np.float32)
        pad[:rows] = full[y0:y0 + rows]
        bands.append(pad)
    assert np.array_equal(multigpu.assemble_numpy(bands, 50, 3), full)


# ---- interleaved stripes (swr_set_band_interleaved): partition + reassembly, on the CPU ----
@pytest.mark.parametrize("height,world,k", [(100, 2, 1), (333, 3, 2), (256, 4, 4), (64, 8, 1), (40, 3, 5)])
def test_stripe_partition_covers_every_row_once(height, world, k):
    rows = multigpu.stripe_rows(height, world, k)
    allr = np.concatenate(rows)
    assert sorted(allr.tolist()) == list(range(height))
    for r, idx in enumerate(rows):
        assert all(((int(y) // 16) // k) % world == r for y in idx) and np.all(np.diff(idx) > 0)
    # with one stripe per rank of ceil(rows / world) tile rows the stripes ARE the contiguous bands (when they divide evenly)
    if multigpu.tile_rows(height) % world == 0:
        per = multigpu.tile_rows(height) // world
        bands = multigpu.band_partition(height, world)
        for r, idx in enumerate(multigpu.stripe_rows(height, world, per)):
            y0, n = multigpu.band_pixel_rows(height, bands[r])
            assert idx.tolist() == list(range(y0, y0 + n))


def _stripe_worker(rank, world, port, height, width, k, ok):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    full = _frame(height, width)
    rows = multigpu.stripe_rows(height, world, k)
    cap = max(len(r) for r in rows)
    local = torch.zeros((cap, width, 4), dtype=torch.float32)
    local[:len(rows[rank])] = torch.from_numpy(full[rows[rank]])
    bufs = [torch.empty_like(local) for _ in range(world)] if rank == 0 else None
    dist.gather(local, gather_list=bufs, dst=0)
    if rank == 0:
        out = multigpu.assemble_stripes(bufs, height, world, k)
        ok[0] = int(np.array_equal(out.numpy(), full))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,height,width,k", [(2, 100, 16, 1), (3, 333, 8, 2)])
def test_interleaved_stripes_gather_reassembles_the_frame(world, height, width, k):
    ok = mp.get_context("spawn").Array("i", [0])
    mp.spawn(_stripe_worker, args=(world, _free_port(), height, width, k, ok), nprocs=world, join=True)
    assert ok[0] == 1
