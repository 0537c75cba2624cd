"""Scenario-batch data parallelism.

Every (scenario, ego) solve is independent (evaluate.py:469-558 solves the agents of a
timestep against the same predictions), so a batch shards into contiguous blocks, one
per GPU / process, with NO collective on the data path.  The only exchange is one
all-gather of the first-step controls u*[:, :, 0] so every rank holds the full action
vector (256 KiB per rank at 32 768 scenarios): torch.distributed all_gather_into_tensor,
which is RCCL over xGMI with the "nccl" backend and plain TCP with "gloo" (CPU tests)."""
import torch
import torch.distributed as dist


def shard_range(B_total, rank, world):
    """Contiguous block [lo, hi) of rank `rank`; the first B_total % world ranks get one more."""
    q, r = divmod(B_total, world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def first_controls(u):
    """u[B,2,N] -> u[:, :, 0] contiguous ([B,2]: the (a, df) each agent applies, evaluate.py:492)."""
    return u[:, :, 0].contiguous()


def allgather_controls(u0_local, B_total=None, group=None):
    """All-gather of per-shard first-step controls.  u0_local[b_r, 2] on this rank's device
    (CUDA tensor with nccl/RCCL, CPU tensor with gloo) -> [B_total, 2] on every rank.
    Equal shards use one all_gather_into_tensor; ragged shards pad to the largest shard."""
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return u0_local
    world = dist.get_world_size(group)
    n_local = torch.tensor([u0_local.shape[0]], device=u0_local.device, dtype=torch.int64)
    if B_total is not None and B_total % world == 0:
        out = torch.empty((B_total, 2), dtype=u0_local.dtype, device=u0_local.device)
        dist.all_gather_into_tensor(out, u0_local.contiguous(), group=group)
        return out
    sizes = [torch.zeros_like(n_local) for _ in range(world)]
    dist.all_gather(sizes, n_local, group=group)
    sizes = [int(s.item()) for s in sizes]
    m = max(sizes)
    pad = torch.zeros((m, 2), dtype=u0_local.dtype, device=u0_local.device)
    pad[:u0_local.shape[0]] = u0_local
    out = torch.empty((world * m, 2), dtype=u0_local.dtype, device=u0_local.device)
    dist.all_gather_into_tensor(out, pad, group=group)
    return torch.cat([out[r * m:r * m + sizes[r]] for r in range(world)], dim=0)


def _words(t):
    """Bit patterns of a float tensor as int64 words, hashed to 31 bits each (NaN-safe: no float comparison)."""
    t = t.contiguous().reshape(-1)
    w = t.view(torch.int64) if t.element_size() == 8 else t.view(torch.int32).to(torch.int64)
    return (w ^ (w >> 32) ^ (w >> 17)) & 0x7FFFFFFF


def _checksums(t, first_index):
    """(plain, position-weighted) integer checksums of tensor `t`, whose first element has global index `first_index`.
    Sized so that nothing overflows int64 for up to 2^24 elements: 31-bit words summed; 16-bit words x 16-bit weights."""
    h = _words(t)
    pos = torch.arange(first_index, first_index + h.numel(), device=h.device, dtype=torch.int64)
    return torch.stack([h.sum(), ((h & 0xFFFF) * (pos % 65521 + 1)).sum()])


def verify_gathered(u0_local, gathered, group=None):
    """Untimed self-check of the exchange: does the all-gathered action vector `gathered` [B_total, 2] hold,
    on THIS rank, every rank's shard u0 [b_r, 2] in rank order?  Every rank reduces two integer checksums of the bit
    patterns of its own shard (one plain, one weighted by the element's global position) with an all-reduce SUM, and
    compares the totals with the same checksums of its gathered vector; its own block is compared word for word.
    -> dict(ok, ranks_seen, per_rank_B_local, B_total).  Collective: every rank must call it."""
    dev = u0_local.device
    if dist.is_available() and dist.is_initialized():
        world, rank = dist.get_world_size(group), dist.get_rank(group)
    else:
        world, rank = 1, 0
    n_local = torch.tensor([u0_local.shape[0]], device=dev, dtype=torch.int64)
    if world > 1:
        sizes = [torch.zeros_like(n_local) for _ in range(world)]
        dist.all_gather(sizes, n_local, group=group)
        sizes = [int(s.item()) for s in sizes]
    else:
        sizes = [int(n_local.item())]
    lo = sum(sizes[:rank])
    per_row = int(u0_local[0].numel()) if u0_local.shape[0] else 2
    total = _checksums(u0_local, lo * per_row)
    seen = torch.ones(1, device=dev, dtype=torch.int64)
    if world > 1:
        dist.all_reduce(total, op=dist.ReduceOp.SUM, group=group)
        dist.all_reduce(seen, op=dist.ReduceOp.SUM, group=group)
    ok = tuple(gathered.shape) == (sum(sizes), per_row) and gathered.dtype == u0_local.dtype
    if ok:
        ok = bool(torch.equal(_checksums(gathered, 0), total))
        ok = ok and bool(torch.equal(_words(gathered[lo:lo + sizes[rank]]), _words(u0_local)))
    flag = torch.tensor([1 if ok else 0], device=dev, dtype=torch.int64)
    if world > 1:
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)       # one rank's failure fails the line everywhere
    return dict(ok=bool(flag.item() == 1), ranks_seen=int(seen.item()), per_rank_B_local=sizes, B_total=sum(sizes))
