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


HEAD_CAP = 1 << 16      # first message of a variable-size point-to-point transfer: 8 bytes of length + this many payload bytes


class DistComm:
    """One rank per process over torch.distributed (backend "nccl" = RCCL over xGMI, or "gloo" in rehearsals / CPU tests).

    gather(name, [value])              one fixed-size int64 all-gather (the schema of a name is fixed: every rank contributes the same
                                       number of integers); values are coerced with int() and must fit 63 bits in magnitude (u64 words
                                       travel as their two's complement)
    gather_sized(name, [tensor], sizes) one padded uint8 all-gather of byte strings whose lengths every rank already knows
    p2p(name, sends, recvs, ranks)     point-to-point byte transfers (neighbour halo copies): sends = [[(dst, tensor)]], recvs =
                                       [[(src, nbytes or None)]]; a transfer of unknown size starts with a fixed-size message
                                       (length + the first HEAD_CAP bytes), the rest follows only if there is a rest
    Counters `collectives` / `p2p_messages` let the tests pin the number of exchange steps."""

    def __init__(self, dist, device):
        self.dist, self.device = dist, device
        self.world, self.rank = dist.get_world_size(), dist.get_rank()
        self.on_cpu = dist.get_backend() != "nccl"
        self.dev = torch.device("cpu") if self.on_cpu else device
        self.collectives = self.p2p_messages = 0

    # -- collectives
    def gather(self, name, values):
        v = values[0]
        ints = [int(x) for x in v] if isinstance(v, (tuple, list)) else [int(v)]
        for x in ints:
            if not -(1 << 63) <= x < (1 << 64):
                raise ValueError(f"gather({name}): {x} does not fit 64 bits")
        ints = [x - (1 << 64) if x >= (1 << 63) else x for x in ints]
        t = torch.tensor(ints, dtype=torch.int64, device=self.dev)
        bufs = [torch.empty_like(t) for _ in range(self.world)]
        self.dist.all_gather(bufs, t)
        self.collectives += 1
        rows = torch.stack(bufs).cpu().tolist()          # one host read-back for the whole exchange
        if isinstance(v, (tuple, list)):
            return [type(v)(r) for r in rows]
        return [r[0] for r in rows]

    def gather_sized(self, name, values, sizes):
        v = values[0]
        assert int(v.numel()) == int(sizes[self.rank]), (name, int(v.numel()), sizes[self.rank])
        cap = max(max(int(x) for x in sizes), 1)
        pad = torch.zeros(cap, dtype=torch.uint8, device=self.dev)
        pad[:v.numel()] = v.cpu() if self.on_cpu else v
        bufs = [torch.empty(cap, dtype=torch.uint8, device=self.dev) for _ in range(self.world)]
        self.dist.all_gather(bufs, pad)
        self.collectives += 1
        return [bufs[q][:int(sizes[q])].to(self.device) for q in range(self.world)]

    # -- point to point
    def _run(self, ops):
        if ops:
            for w in self.dist.batch_isend_irecv(ops):
                w.wait()

    def p2p(self, name, sends, recvs, ranks):
        dist = self.dist
        out_msgs, in_specs = sends[0], recvs[0]
        stage = lambda t: (t.cpu() if self.on_cpu else t).contiguous()
        ops, first_out, first_in, known_in = [], [], [], []
        for dst, t in out_msgs:       # the peer's matching recv entry decides the protocol: a transfer is "unknown size" iff the name says so
            if name == "heads":
                n = int(t.numel())
                m = torch.zeros(8 + HEAD_CAP, dtype=torch.uint8, device=self.dev)
                m[:8] = torch.tensor(list(n.to_bytes(8, "little")), dtype=torch.uint8, device=self.dev)
                k = min(n, HEAD_CAP)
                m[8:8 + k] = stage(t[:k])
                first_out.append((dst, t, m))
                ops.append(dist.P2POp(dist.isend, m, dst))
            else:
                ops.append(dist.P2POp(dist.isend, stage(t), dst))
            self.p2p_messages += 1
        for src, n in in_specs:
            if n is None:
                m = torch.empty(8 + HEAD_CAP, dtype=torch.uint8, device=self.dev)
                first_in.append((src, m))
                ops.append(dist.P2POp(dist.irecv, m, src))
            else:
                buf = torch.empty(int(n), dtype=torch.uint8, device=self.dev)
                known_in.append((src, buf))
                ops.append(dist.P2POp(dist.irecv, buf, src))
        self._run(ops)
        # the rest of a transfer that did not fit its first message (a first line longer than HEAD_CAP: rare)
        ops, rest_in = [], {}
        for dst, t, m in first_out:
            if int(t.numel()) > HEAD_CAP:
                ops.append(dist.P2POp(dist.isend, stage(t[HEAD_CAP:]), dst))
                self.p2p_messages += 1
        heads = {}
        for src, m in first_in:
            n = int.from_bytes(bytes(m[:8].cpu().tolist()), "little")
            heads[src] = (n, m)
            if n > HEAD_CAP:
                rest_in[src] = torch.empty(n - HEAD_CAP, dtype=torch.uint8, device=self.dev)
                ops.append(dist.P2POp(dist.irecv, rest_in[src], src))
        self._run(ops)
        got = []
        known = {src: buf for src, buf in known_in}
        for src, n in in_specs:
            if n is None:
                total, m = heads[src]
                t = m[8:8 + min(total, HEAD_CAP)]
                if src in rest_in:
                    t = torch.cat([t, rest_in[src]])
                got.append(t.to(self.device))
            else:
                got.append(known[src].to(self.device))
        return [got]
