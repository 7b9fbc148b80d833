"""Sequence sharding across GPUs: the path shards by independent event sequences -- one
process per GPU, contiguous blocks of sequences per rank, no data-path collective; the only communication is an
all-gather that collates the reconstructed frames (RCCL over xGMI on GPUs, gloo in the CPU tests)."""
import torch
import torch.distributed as dist


def shard_range(n_sequences, rank, world):
    """Contiguous block of sequences owned by `rank` (the first n % world ranks get one extra)."""
    if n_sequences < 0 or world < 1 or not (0 <= rank < world):
        raise ValueError("bad shard arguments")
    base, extra = divmod(n_sequences, world)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def collate_frames(local_frames, n_sequences=None, group=None, force=False):
    """all-gather per-rank frames [B_local, 1, H, W] -> [sum B_local, 1, H, W] in rank order.

    Equal shards use one all_gather_into_tensor; ragged shards (n_sequences % world != 0) pad to the largest
    shard and trim.  Outputs are not needed by the recurrence, so callers may issue this on a side stream."""
    if not (dist.is_available() and dist.is_initialized()):
        return local_frames
    world = dist.get_world_size(group)
    if world == 1 and not force:      # force: run the collective even for one rank (backend smoke test)
        return local_frames
    B = local_frames.shape[0]
    if n_sequences is None or n_sequences % world == 0:
        out = local_frames.new_empty((world * B,) + tuple(local_frames.shape[1:]))
        dist.all_gather_into_tensor(out, local_frames.contiguous(), group=group)
        return out
    sizes = [shard_range(n_sequences, r, world) for r in range(world)]
    bmax = max(e - s for s, e in sizes)
    pad = local_frames.new_zeros((bmax,) + tuple(local_frames.shape[1:]))
    pad[:B] = local_frames
    out = local_frames.new_empty((world * bmax,) + tuple(local_frames.shape[1:]))
    dist.all_gather_into_tensor(out, pad, group=group)
    return torch.cat([out[r * bmax: r * bmax + (e - s)] for r, (s, e) in enumerate(sizes)], 0)
