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
