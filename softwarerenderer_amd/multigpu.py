"""Multi-GPU frames: contiguous bands of 16-pixel tile rows, one process per GPU, colour gather.

The reference already parallelises a triangle over tile rows (Rasterizer.cs:462); pixels are
independent given the ordered triangle list, so a frame shards by screen-space tile rows with NO
data-path collective while rendering: every rank holds the (small) geometry, runs vertex + setup
for all triangles and bins / rasterises only the tiles of its band -- a triangle straddling a band
edge is rasterised on both sides with identical per-tile arithmetic, so the union of the bands is
bit-identical to the single-GPU frame.  The one exchange step is the end-of-frame gather of the
colour bands to rank 0 (RCCL send/recv over xGMI: 7 links into the root concurrently; a ring
all-gather would be per-link bound).

torch / torch.distributed are plumbing here (device memory for the bands, streams, RCCL).
"""
from __future__ import annotations

from typing import List, Tuple

import numpy as np

TILE = 16


def tile_rows(height: int) -> int:
    return (max(height, 0) + TILE - 1) // TILE


def band_partition(height: int, world_size: int) -> List[Tuple[int, int]]:
    """Split the frame's tile rows into `world_size` contiguous bands (first_tile_row, n_tile_rows),
    sizes differing by at most one row; ranks beyond the number of tile rows get empty bands."""
    n = tile_rows(height)
    q, r = divmod(n, world_size)
    out, start = [], 0
    for i in range(world_size):
        cnt = q + (1 if i < r else 0)
        out.append((start, cnt))
        start += cnt
    return out


def stripe_rows(height: int, world_size: int, stripe_tile_rows: int) -> List[np.ndarray]:
    """Interleaved variant (swr_set_band_interleaved): the tile rows are cut into stripes of `stripe_tile_rows` rows, stripe s
    belongs to rank s % world_size.  Returns, per rank, the frame's pixel rows that rank stores, in storage order -- i.e.
    `frame[rows[r]] = band_r[:len(rows[r])]` reassembles the frame (assemble_stripes)."""
    n = tile_rows(height)
    out = [[] for _ in range(world_size)]
    for ty in range(n):
        r = (ty // stripe_tile_rows) % world_size
        out[r].extend(range(ty * TILE, min(height, (ty + 1) * TILE)))
    return [np.asarray(o, dtype=np.int64) for o in out]


def assemble_stripes(bands, height: int, world_size: int, stripe_tile_rows: int, frame=None):
    """Scatter per-rank stripe buffers (numpy arrays or torch tensors, rows beyond a rank's own count are padding) into the
    (height, ...) frame: one indexed row copy per rank."""
    rows = stripe_rows(height, world_size, stripe_tile_rows)
    is_torch = not isinstance(bands[0], np.ndarray)
    if frame is None:
        shape = (height,) + tuple(bands[0].shape[1:])
        if is_torch:
            import torch
            frame = torch.empty(shape, dtype=bands[0].dtype, device=bands[0].device)
        else:
            frame = np.empty(shape, dtype=bands[0].dtype)
    for r, idx in enumerate(rows):
        if len(idx) == 0:
            continue
        if is_torch:
            import torch
            frame.index_copy_(0, torch.as_tensor(idx, device=frame.device), bands[r][:len(idx)])
        else:
            frame[idx] = bands[r][:len(idx)]
    return frame


def band_pixel_rows(height: int, band: Tuple[int, int]) -> Tuple[int, int]:
    """(first pixel row, number of pixel rows) stored by a band."""
    y0 = band[0] * TILE
    y1 = min(height, (band[0] + band[1]) * TILE)
    return y0, max(0, y1 - y0)


def max_band_rows(height: int, world_size: int) -> int:
    return max(band_pixel_rows(height, b)[1] for b in band_partition(height, world_size))


def gather_bands(local_band, height: int, width: int, rank: int, world_size: int, dst: int = 0, group=None, frame=None, dist=None,
                 async_op: bool = False, post_stream=None, force_collective: bool = False):
    """Gather the per-rank colour bands (torch tensors of shape (max_band_rows, width, C), rows beyond a
    band's own count are padding) to `dst`; returns the assembled (height, width, C) frame there, else None.
    `frame` (optional, dst only) is a preallocated output.  When every band has the same number of pixel rows
    the receive buffers ARE row blocks of the frame (no assembly copy).  Works on any backend
    (nccl == RCCL on ROCm; gloo on CPU for the tests).
    async_op=True returns (frame, work) without making the calling stream wait for the collective: the caller keeps
    rendering and calls work.wait() only before it reuses `local_band` (RCCL: a stream-level wait, the host never blocks).
    Bands of unequal size need a copy after the collective; in async mode it runs on `post_stream` behind the collective."""
    import torch
    if dist is None:
        import torch.distributed as dist
    if world_size == 1 and not force_collective:         # (force_collective: bench.py --force-dist runs the RCCL leg with one rank)
        out = local_band[:band_pixel_rows(height, (0, tile_rows(height)))[1]]
        return (out, None) if async_op else out
    bands = band_partition(height, world_size)
    rows = [band_pixel_rows(height, b) for b in bands]
    bufs = None
    uniform = all(r[1] == local_band.shape[0] for r in rows)
    if rank == dst:
        if frame is None:
            frame = torch.empty((height, width) + tuple(local_band.shape[2:]), dtype=local_band.dtype, device=local_band.device)
        if uniform:
            bufs = [frame[y0:y0 + n] for (y0, n) in rows]          # contiguous row blocks: receive in place
        else:
            bufs = [torch.empty_like(local_band) for _ in range(world_size)]
    work = dist.gather(local_band, gather_list=bufs, dst=dst, group=group, async_op=True) if async_op else \
        dist.gather(local_band, gather_list=bufs, dst=dst, group=group)

    def assemble():
        for r, (y0, n) in enumerate(rows):
            if n:
                frame[y0:y0 + n] = bufs[r][:n]

    if rank == dst and not uniform:
        if async_op and local_band.is_cuda:
            stream = post_stream if post_stream is not None else torch.cuda.current_stream()
            with torch.cuda.stream(stream):
                if work is not None:
                    work.wait()
                assemble()
            # the receive buffers were allocated on the calling stream and are read on `stream`: tell the caching allocator, or it
            # may hand their memory to the next allocation of the render stream while the assembly copy is still pending
            for b in bufs:
                b.record_stream(stream)
        else:
            if async_op and work is not None:
                work.wait()
            assemble()
    if async_op:
        return (frame if rank == dst else None), work
    return frame if rank == dst else None


def assemble_numpy(bands: List[np.ndarray], height: int, world_size: int) -> np.ndarray:
    """Host-side equivalent of gather_bands for arrays already on one process (tests)."""
    parts = []
    for r, band in enumerate(band_partition(height, world_size)):
        _, rows = band_pixel_rows(height, band)
        parts.append(bands[r][:rows])
    return np.concatenate(parts, axis=0)
