"""Host-side multi-GPU logic (one process per GPU, torch.distributed; backend "nccl" is RCCL on ROCm, "gloo" in the CPU tests).

Independent files and segment-aligned shards need NO data-path collective: ranks only agree on who owns
what (a pure function of the sizes) and reduce their timings / counters at the end.
"""
import torch


def shard_by_bytes(sizes, rank, world):
    """Indices of the files rank `rank` owns: contiguous split by cumulative byte balance (deterministic on every rank)."""
    total = sum(sizes)
    lo, hi = total * rank / world, total * (rank + 1) / world
    out, acc = [], 0
    for i, s in enumerate(sizes):
        mid = acc + s / 2.0
        if lo <= mid < hi or (rank == world - 1 and mid >= hi):
            out.append(i)
        acc += s
    return out


def _device(dist):
    return torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")


def all_reduce_max(value, dist):
    if dist is None:
        return float(value)
    t = torch.tensor([value], dtype=torch.float64, device=_device(dist))
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def all_reduce_sum(value, dist):
    if dist is None:
        return float(value)
    t = torch.tensor([value], dtype=torch.float64, device=_device(dist))
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def make_gather(dist, device):
    """gather(name, [local_value]) -> list over all ranks.  Integers and tuples of integers (the carries of opus_amd/shard.py: context
    function, token count + tail kinds, net depth, cut, length, status) travel as ONE fixed-size int64 tensor all-gather — no pickling,
    no host staging under RCCL; 1-D uint8 tensors of different lengths (head bytes, prefix halo, definition headers) as their sizes
    (the same integer gather) and one padded all-gather.  Works with backend "nccl" (RCCL) and "gloo"."""
    world = dist.get_world_size()
    on_cpu = dist.get_backend() != "nccl"
    dev = torch.device("cpu") if on_cpu else device

    def gather_ints(vals):
        t = torch.tensor(vals, dtype=torch.int64, device=dev)
        bufs = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(bufs, t)
        return [b.tolist() for b in bufs]

    def gather(name, values):
        v = values[0]
        if isinstance(v, bool) or v is None:
            out = [None] * world
            dist.all_gather_object(out, v)
            return out
        if isinstance(v, int):
            return [r[0] for r in gather_ints([v])]
        if isinstance(v, (tuple, list)) and all(isinstance(x, int) and not isinstance(x, bool) for x in v):
            return [type(v)(r) for r in gather_ints(list(v))]
        if not torch.is_tensor(v):
            out = [None] * world
            dist.all_gather_object(out, v)
            return out
        sizes = [r[0] for r in gather_ints([int(v.numel())])]
        cap = max(max(sizes), 1)
        pad = torch.zeros(cap, dtype=torch.uint8, device=dev)
        pad[:v.numel()] = v.cpu() if on_cpu else v
        bufs = [torch.empty(cap, dtype=torch.uint8, device=dev) for _ in range(world)]
        dist.all_gather(bufs, pad)
        return [bufs[q][:sizes[q]].to(device) for q in range(world)]

    return gather
