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
