"""Multi-GPU batch sharding for the inference engine.

Utterances are independent (no cross-utterance term anywhere: every kernel of the reference
indexes by batch_offset, nv_wavenet_singleblock.cuh:70, nv_wavenet_persistent.cuh:110), so the
batch shards with no data-path collective: rank g of G generates utterances
[g*B/G, (g+1)*B/G) on its own GPU with replicated weights, and ONE collective at the end gathers
the [B/G][N] int32 sample blocks, which are contiguous row ranges of the [B][N] output
(SURVEY.md 8e).  The reference itself is single-GPU; this is the added multi-GPU path of
BASELINE.json configs[4] (batch 64 = 8 x 8).  Backend "nccl" is RCCL on ROCm; "gloo" on CPU.
"""
import numpy as np
import torch
import torch.distributed as dist


def shard_range(total_batch, world_size, rank):
    """Contiguous, balanced split: the first (total % world) ranks get one extra utterance."""
    base, extra = divmod(total_batch, world_size)
    start = rank * base + min(rank, extra)
    return start, base + (1 if rank < extra else 0)


def shard_inputs(Lh, sel, world_size, rank):
    """Slice conditioning [N][L][B][2R] and selectors [N][B] along the batch axis for this rank.
    Works on numpy arrays and torch tensors; returns contiguous copies."""
    B = Lh.shape[2]
    s, n = shard_range(B, world_size, rank)
    if isinstance(Lh, np.ndarray):
        return np.ascontiguousarray(Lh[:, :, s:s + n]), np.ascontiguousarray(sel[:, s:s + n])
    return Lh[:, :, s:s + n].contiguous(), sel[:, s:s + n].contiguous()


def shard_features(x, world_size, rank):
    """The same for the conditioning's SOURCE (round 5: the conditioning is computed in the generation kernel): mel frames or upsampled
    features [B][n_cond][frames | samples] are batch-major, so a rank's share is a contiguous row range -- no copy for torch tensors
    (a view; WavenetEngine.setMel / setFeatures take any strides), a contiguous slice for numpy arrays."""
    s, n = shard_range(x.shape[0], world_size, rank)
    return np.ascontiguousarray(x[s:s + n]) if isinstance(x, np.ndarray) else x[s:s + n]


def gather_samples(y_local, total_batch, group=None, async_op=False):
    """All ranks contribute their [b_local][N] int32 block; every rank receives the [B][N] result
    (one all_gather over RCCL/xGMI: 4*B*N bytes in total). Ragged shards are padded to the
    largest shard for the collective and trimmed afterwards.
    Returns (y_full, None), or (None, finish) with async_op=True: call finish() for the result."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    N = y_local.shape[1]
    sizes = [shard_range(total_batch, world, r)[1] for r in range(world)]
    assert y_local.shape[0] == sizes[rank]
    mx = max(sizes)
    if y_local.shape[0] != mx:
        pad = torch.zeros(mx - y_local.shape[0], N, dtype=y_local.dtype, device=y_local.device)
        y_local = torch.cat([y_local, pad], 0)
    dev = y_local.device
    if dist.get_backend(group) == "gloo" and dev.type != "cpu":
        y_local = y_local.cpu()            # gloo (CPU tests / smoke runs): stage through host memory
    out = torch.empty(world * mx, N, dtype=y_local.dtype, device=y_local.device)
    work = dist.all_gather_into_tensor(out, y_local.contiguous(), group=group, async_op=async_op)

    def finish():
        if work is not None:
            work.wait()
        res = out if all(s == mx for s in sizes) else torch.cat([out[r * mx:r * mx + sizes[r]] for r in range(world)], 0)
        return res.to(dev) if res.device != dev else res

    return (None, finish) if async_op else (finish(), None)


class ChunkGatherer:
    """Streaming form of gather_samples for run_chunks ("with run_chunks, gather per chunk"): every finished chunk
    of `count` samples is gathered on its own -- columns [first, first + count) of every rank's [b_local][N] block -- while the
    next chunk is generated, so the full [B][N] result is complete one collective after the last chunk instead of one whole-
    utterance collective at the end.  Use as the consumer callback of run_chunks:

        g = ChunkGatherer(total_batch, num_samples, y_local)      # y_local: the rank's [b_local][N] output buffer (host or device)
        engine.run_chunks(chunk, g, num_samples, b_local, y_local)
        y_full = g.finish()                                       # [total_batch][num_samples] on every rank

    At most `depth` collectives are in flight (the oldest is completed before a new one starts)."""

    def __init__(self, total_batch, num_samples, y_local, group=None, depth=2):
        self.total, self.N, self.y_local, self.group, self.depth = total_batch, num_samples, y_local, group, depth
        self.full = None
        self.pending = []          # (first, count, finish)

    def _land(self):
        first, count, fin = self.pending.pop(0)
        blk = fin()
        if self.full is None:
            self.full = torch.zeros(self.total, self.N, dtype=blk.dtype, device=blk.device)
        self.full[:, first:first + count] = blk

    def __call__(self, y_out, first, count):
        y = self.y_local if self.y_local is not None else y_out
        y = torch.as_tensor(y)
        while len(self.pending) >= self.depth:
            self._land()
        _, fin = gather_samples(y[:, first:first + count].contiguous(), self.total, self.group, async_op=True)
        self.pending.append((first, count, fin))

    def finish(self):
        while self.pending:
            self._land()
        return self.full
