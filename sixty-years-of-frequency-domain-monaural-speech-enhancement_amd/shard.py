"""Utterance sharding across ranks (one rank = one GPU = one engine replica) and the gather of enhanced waveforms.

The reference decodes `for file_id in file_list` on one pinned device (e.g. DCCRN/dccrn_decode_vb.py:24); utterances
are independent, so the path shards with no data-path collective.  The only exchange is the final gather of enhanced
waveforms to rank 0 (RCCL over xGMI with backend "nccl", gloo in the CPU tests) - 256 kB per 4 s utterance.
"""
import torch
import torch.distributed as dist


def shard_range(n_items, rank, world):
    """Contiguous shard [lo, hi) of n_items for `rank`; sizes differ by at most one."""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_counts(n_items, world):
    return [shard_range(n_items, r, world)[1] - shard_range(n_items, r, world)[0] for r in range(world)]


def gather_waveforms(local, n_items, dst=0, group=None):
    """local: [n_local, L] tensor of this rank's enhanced waveforms (n_local = its shard size).
    Returns the [n_items, L] tensor in original order on rank `dst`, None elsewhere.  Uneven shards are padded to
    the largest shard for the collective and trimmed on arrival."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return local
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    counts = shard_counts(n_items, world)
    assert local.shape[0] == counts[rank], (local.shape, counts, rank)
    cmax = max(counts)
    send = local
    if local.shape[0] < cmax:
        send = torch.zeros((cmax,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        send[:local.shape[0]] = local
    send = send.contiguous()
    bufs = [torch.empty_like(send) for _ in range(world)] if rank == dst else None
    dist.gather(send, bufs, dst=dst, group=group)
    if rank != dst:
        return None
    return torch.cat([b[:c] for b, c in zip(bufs, counts)], dim=0)


class GatherPipe:
    """Steady-state gather for a decode loop: preallocated receive rows on rank `dst`, asynchronous collective, two send
    slots - so the gather of batch k travels over xGMI while batch k+1 is being enhanced (equal shards only).

        pipe = GatherPipe(n_local, n_samples, device)
        for k in ...:
            out = pipe.slot()                  # [n_local, n_samples] buffer to enhance into (waits for its last gather)
            engine.enhance_batch(wav, out)
            pipe.submit()                      # async gather of that slot
        rows = pipe.finish()                   # rank dst: list of `world` tensors of the LAST batch, else None
    """

    def __init__(self, n_local, n_samples, device, dst=0, group=None, dtype=torch.float32, single_rank=False):
        # single_rank: run the collective even in a one-rank group (exercises the RCCL path on a 1-GPU box)
        self.group, self.dst = group, dst
        self.on = dist.is_available() and dist.is_initialized() and (dist.get_world_size(group) > 1 or single_rank)
        self.world = dist.get_world_size(group) if self.on else 1
        self.rank = dist.get_rank(group) if self.on else 0
        self.send = [torch.empty((n_local, n_samples), dtype=dtype, device=device) for _ in range(2)]
        # gloo has no device-side gather: the rows are staged through pinned host memory (CPU tests, or a GPU box
        # whose RCCL cannot be used); nccl (= RCCL over xGMI) gathers device to device
        self.host = self.on and dist.get_backend(group) == 'gloo' and torch.device(device).type != 'cpu'
        rdev = 'cpu' if self.host else device
        self.stage = [torch.empty((n_local, n_samples), dtype=dtype).pin_memory() for _ in range(2)] if self.host else None
        self.recv = [[torch.empty((n_local, n_samples), dtype=dtype, device=rdev) for _ in range(self.world)]
                     for _ in range(2)] if (self.on and self.rank == dst) else [None, None]
        self.work = [None, None]
        self.k = 0

    def slot(self):
        i = self.k & 1
        if self.work[i] is not None:
            self.work[i].wait()
            self.work[i] = None
        return self.send[i]

    def submit(self):
        i = self.k & 1
        if self.on:
            src = self.send[i]
            if self.host:
                self.stage[i].copy_(src)          # synchronous D2H on the current stream
                src = self.stage[i]
            self.work[i] = dist.gather(src, self.recv[i], dst=self.dst, group=self.group, async_op=True)
        self.k += 1

    def finish(self):
        for i in (0, 1):
            if self.work[i] is not None:
                self.work[i].wait()
                self.work[i] = None
        last = (self.k - 1) & 1
        if not self.on:
            return [self.send[last]]
        return self.recv[last] if self.rank == self.dst else None


def run_sharded(enhance_fn, wav, dst=0, group=None):
    """wav: [N, L] batch of equal-length clips, identical on every rank (or only meaningful rows of the local
    shard).  Each rank enhances its shard with `enhance_fn([n_local, L]) -> [n_local, L_out]`; rank dst gets all N."""
    world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
    rank = dist.get_rank(group) if world > 1 else 0
    lo, hi = shard_range(wav.shape[0], rank, world)
    out = enhance_fn(wav[lo:hi])
    return gather_waveforms(out, wav.shape[0], dst, group)
