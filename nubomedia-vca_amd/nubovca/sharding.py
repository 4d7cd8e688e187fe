"""Multi-GPU frontend plumbing: media streams are the independent units (each has private temporal
state -- Faces list, MHI, previous gray), so they are sharded statically over ranks and never migrate;
the only collective is the per-tick gather of a fixed-size box table.
Works with any torch.distributed backend ("nccl" == RCCL on the GPU box, "gloo" in CPU tests)."""
import numpy as np
import torch
import torch.distributed as dist


def streams_of_rank(n_streams, world, rank):
    """static assignment stream_id % world == rank"""
    return [s for s in range(n_streams) if s % world == rank]


RANK_SHIFT = 16          # column 0 of a row: box count in the low 16 bits, the producing rank above them


def pack_boxes(results, max_boxes, rank=0):
    """results: list of (boxes[n,4], ids[n]) per local stream -> int32 [n_local, 1 + 4*max_boxes]; every row is stamped with
    the rank that produced it (the gathered table then shows which ranks took part)"""
    tab = np.zeros((len(results), 1 + 4 * max_boxes), np.int32)
    for i, (b, _) in enumerate(results):
        n = min(len(b), max_boxes)
        tab[i, 0] = n | (rank << RANK_SHIFT)
        tab[i, 1:1 + 4 * n] = np.asarray(b[:n], np.int32).reshape(-1)
    return tab


def pack_box_arrays(boxes, counts, out=None, rank=0):
    """the same table from the batched call's raw result arrays (boxes int32 [n, cap, 4], counts int32 [n]) without
    per-stream python work; entries past a stream's count are zeroed like pack_boxes leaves them"""
    n, cap = boxes.shape[0], boxes.shape[1]
    tab = out if out is not None else np.empty((n, 1 + 4 * cap), np.int32)
    k = np.minimum(counts, cap)
    tab[:, 0] = k | (rank << RANK_SHIFT)
    live = (np.arange(cap, dtype=np.int32)[None, :] < k[:, None])
    np.multiply(boxes, live[:, :, None], out=tab[:, 1:].reshape(n, cap, 4))
    return tab


def unpack_boxes(tab):
    return [np.asarray(row[1:1 + 4 * (int(row[0]) & 0xffff)], np.int32).reshape(-1, 4) for row in np.asarray(tab)]


def table_ranks(gathered):
    """the rank stamp of every row of a gathered [world, n_local, cols] table"""
    return np.asarray(gathered)[..., 0] >> RANK_SHIFT


def gather_tables(local_tab, device=None):
    """all_gather of equally shaped tables; returns [world, n_local, cols] (numpy) on every rank"""
    world = dist.get_world_size() if dist.is_initialized() else 1
    t = torch.from_numpy(np.ascontiguousarray(local_tab))
    if device is not None:
        t = t.to(device)
    if world == 1:
        return t.cpu().numpy()[None]
    out = torch.empty((world * t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
    dist.all_gather_into_tensor(out, t)
    return out.cpu().numpy().reshape((world,) + tuple(t.shape))


class TableGather:
    """Per-tick result gather that stays off the critical path: the all_gather of tick k is started asynchronously and
    only waited for when tick k+1 hands in its table (or at `finish`), so the collective overlaps the next tick's
    kernels.  `last()` returns the most recently completed [world, n_local, cols] table."""

    def __init__(self, device=None):
        self.device, self.pending, self.done = device, None, None
        self.outs, self.flip = [None, None], 0

    def submit(self, local_tab):
        self._complete()
        world = dist.get_world_size() if dist.is_initialized() else 1
        t = torch.from_numpy(np.ascontiguousarray(local_tab))
        if self.device is not None:
            t = t.to(self.device)
        if world == 1:
            self.done = (t, (1,) + tuple(t.shape))
            return
        shape = (world * t.shape[0],) + tuple(t.shape[1:])
        self.flip ^= 1                     # two output tensors alternate: the previous one may still be read by last()
        out = self.outs[self.flip]
        if out is None or tuple(out.shape) != shape or out.dtype != t.dtype or out.device != t.device:
            out = self.outs[self.flip] = torch.empty(shape, dtype=t.dtype, device=t.device)
        work = dist.all_gather_into_tensor(out, t, async_op=True)
        self.pending = (work, out, t, (world,) + tuple(t.shape))

    def _complete(self):
        if self.pending is not None:
            work, out, _t, shape = self.pending
            work.wait()
            self.done, self.pending = (out, shape), None

    def finish(self):
        self._complete()

    def last(self):
        self._complete()
        if self.done is None:
            return None
        out, shape = self.done
        return out.cpu().numpy().reshape(shape)


def merge_by_stream(gathered, n_streams, world):
    """gathered[r][j] belongs to stream streams_of_rank(...)[j]; returns per-stream list of boxes"""
    res = [None] * n_streams
    for r in range(world):
        for j, s in enumerate(streams_of_rank(n_streams, world, r)):
            res[s] = unpack_boxes(gathered[r][j:j + 1])[0]
    return res
