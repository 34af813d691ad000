"""One huge source sharded by byte range over several GPUs (BASELINE config 4).

Every rank holds a contiguous byte range of the source in HBM, cut at ARBITRARY positions.  The sequential lexer state of the
reference (parse.rs:364-391: string/comment context, `ignore_dents`, indentation) is carried across ranks by a chain of tiny
all-gathers; each rank then owns whole top-level segments, i.e. a standalone well-formed source:

  1. heads      cuts move to the next '\\n' (a line never splits a run token); the skipped head bytes go to the left neighbour
  2. ctx        composed context-DFA transition function of the rank's bytes            (4 B / rank)   -> start state
  3. tails      number of raw tokens + kinds of the last 8 tokens                       (16 B / rank)  -> `record {` look-back
  4. depth      net change of the bracket counter                                       (8 B / rank)   -> absolute depth
  5. cuts       first top-level segment start provable inside the shard, and its length (16 B / rank)  -> ownership
  6. halo       the bytes before a rank's cut go to its left neighbour                  (bulk, one neighbour)
  7. headers    definition headers (functions as `extern` signatures, types, globals)   (bulk, variable) -> global tables for IR

Steps 1-5 are latency-bound (a few bytes); 6 is one point-to-point sized transfer; 7 is the variable-size all-gather of the
definition table.  The `Comm` classes hide torch.distributed so that the same code runs W simulated ranks in one process
(tests) or one rank per GPU (bench.py).
"""
import ctypes as C

import torch

from . import frontend as F

CS_N = 0
CTX_IDENTITY = 0 | (1 << 3) | (2 << 6) | (3 << 9) | (4 << 12) | (5 << 15)
TK_NO_PREV = 0x3B3B3B3B3B3B3B3B


def ctx_apply(fn, s):
    return (fn >> (3 * s)) & 7


def ctx_compose(f, g):
    """first f, then g (lex.h ctx_compose)"""
    r = 0
    for s in range(6):
        r |= ctx_apply(g, ctx_apply(f, s)) << (3 * s)
    return r


def fold_tail(prev_kinds, ntok, tail_kinds):
    """kinds of the last 8 tokens after appending a shard with `ntok` tokens whose own tail is `tail_kinds`"""
    own = [(tail_kinds >> (8 * j)) & 0xFF for j in range(min(ntok, 8))]
    old = [(prev_kinds >> (8 * j)) & 0xFF for j in range(8)]
    seq = (own + old)[:8]
    out = 0
    for j, k in enumerate(seq):
        out |= k << (8 * j)
    return out


_HIP = None


def _hip():
    """the HIP runtime, loaded once (the headers of a unit are copied device to device into a torch tensor)"""
    global _HIP
    if _HIP is None:
        _HIP = C.cdll.LoadLibrary("libamdhip64.so")
        _HIP.hipMemcpy.restype = C.c_int
    return _HIP


class LocalComm:
    """W simulated ranks in one process: all_gather returns the list the driver assembled."""
    def __init__(self, world):
        self.world = world


def _first_newline(t):
    """index of the first '\\n' in a uint8 device tensor, or -1"""
    n = t.numel()
    step = 1 << 16
    for lo in range(0, n, step):
        chunk = t[lo:lo + step]
        hit = (chunk == 10).nonzero()
        if hit.numel():
            return lo + int(hit[0].item())
    return -1


class RankState:
    """What one rank holds while the carries are exchanged."""
    def __init__(self, ctx, nominal, rank, world):
        self.ctx, self.nominal, self.rank, self.world = ctx, nominal, rank, world
        self.shard = None

    # -- step 1
    def head(self):
        if self.rank == 0:
            self.h = 0
        else:
            i = _first_newline(self.nominal)
            if i < 0:
                raise RuntimeError("shard without a newline: choose larger shards")
            self.h = i + 1
        return self.nominal[:self.h]

    def set_next_head(self, next_head):
        parts = [self.nominal[self.h:]]
        if next_head is not None and next_head.numel():
            parts.append(next_head)
        self.eff = torch.cat(parts) if len(parts) > 1 else parts[0].clone()
        pad = torch.zeros(64, dtype=torch.uint8, device=self.eff.device)
        self.eff_padded = torch.cat([self.eff, pad])          # kernels may read a few bytes past the end
        self.n = self.eff.numel()
        torch.cuda.synchronize()       # the library runs on its own stream: torch's writes must have landed

    # -- step 2
    def ctx_fn(self):
        return self.ctx.shard_ctx(self.eff_padded.data_ptr(), self.n)

    # -- step 3
    def lex(self, start_state):
        self.shard, self.ntok, self.tail = self.ctx.shard_lex(self.eff_padded.data_ptr(), self.n, start_state)
        return self.ntok, self.tail

    # -- step 4
    def net_depth(self, prev_kinds):
        return self.ctx.shard_net_depth(self.shard, prev_kinds)

    # -- step 5
    def cut(self, depth_in):
        if self.rank == 0:
            self.c = 0
        else:
            pos = self.ctx.shard_boundary(self.shard, depth_in)
            self.c = self.n if pos is None else pos           # no provable segment start: this rank owns nothing
        self.ctx.shard_free(self.shard)
        self.shard = None
        return self.c

    def prefix(self):
        """bytes before the cut: they belong to the nearest owning rank on the left"""
        return self.eff[:self.c]

    def set_own(self, right_prefixes):
        parts = [self.eff[self.c:]] + [p for p in right_prefixes if p.numel()]
        self.own = torch.cat(parts) if len(parts) > 1 else parts[0].clone()


def plan(states, gather):
    """Runs steps 1-6 for the ranks in `states` (all ranks when simulated, one rank under torch.distributed).
    `gather(name, local_values)` -> list over ALL ranks of the value each rank contributed (python objects or tensors)."""
    world = states[0].world
    heads = gather("heads", [s.head() for s in states])
    for s in states:
        s.set_next_head(heads[s.rank + 1] if s.rank + 1 < world else None)
    fns = gather("ctx", [s.ctx_fn() for s in states])
    for s in states:
        f = CTX_IDENTITY
        for q in range(s.rank):
            f = ctx_compose(f, fns[q])
        s.start_state = ctx_apply(f, CS_N)
    tails = gather("tails", [s.lex(s.start_state) for s in states])
    for s in states:
        prev = TK_NO_PREV
        for q in range(s.rank):
            prev = fold_tail(prev, tails[q][0], tails[q][1])
        s.prev_kinds = prev
    nets = gather("depth", [s.net_depth(s.prev_kinds) for s in states])
    cl = gather("cuts", [(s.cut(sum(nets[:s.rank])), s.n) for s in states])      # (cut, length): one exchange
    cuts, lens = [int(x[0]) for x in cl], [int(x[1]) for x in cl]
    prefixes = gather("prefixes", [s.prefix() for s in states])
    for s in states:
        right = []
        q = s.rank + 1
        while q < world:
            right.append(prefixes[q])
            if cuts[q] < lens[q]:          # rank q owns something: its prefix ends our range
                break
            q += 1
        if s.c >= s.n and s.rank > 0:
            s.own = s.eff[:0]
        else:
            s.set_own(right)
    return states


def compile_sharded(states, gather, lower=True):
    """Steps 6-8: parse the owned bytes, exchange definition headers, lower the own definitions between the others' headers.
    Returns per state a dict(status, unit, own_counts) — `unit` is the joined unit (headers of the ranks before, own
    definitions, headers of the ranks after)."""
    world = states[0].world
    results = []
    parsed = []
    for s in states:
        buf = torch.cat([s.own, torch.zeros(64, dtype=torch.uint8, device=s.own.device)])
        torch.cuda.synchronize()
        u = s.ctx.get_definitions_device(buf.data_ptr(), s.own.numel())
        s._own_buf = buf
        parsed.append(u)
    # a shard that needed a resync pass (swallowed lexical error) changes the lexer context of everything behind
    # it, which the carries exchanged in plan() did not know: the sharded path reports such a source as UNSUPPORTED
    statuses = gather("status", [F.UNSUPPORTED if (u.status == F.OK and u.resync_passes() > 0) else u.status for u in parsed])
    if any(st != F.OK for st in statuses):
        first = next(r for r, st in enumerate(statuses) if st != F.OK)
        return [dict(status=statuses[first], unit=u, ir=None, first_bad_rank=first) for u in parsed]
    hdrs = []
    for s, u in zip(states, parsed):
        p, n = u.headers_device()
        t = torch.empty(n, dtype=torch.uint8, device=s.own.device)
        if n:
            rc = _hip().hipMemcpy(C.c_void_p(t.data_ptr()), C.c_void_p(p), C.c_size_t(n), 3)   # device to device
            if rc != 0:
                raise RuntimeError(f"hipMemcpy of the definition headers failed: hip error {rc}")
        hdrs.append(t)
    all_hdrs = gather("headers", hdrs)
    for s, u in zip(states, parsed):
        own_counts = u.counts()
        before = [all_hdrs[q] for q in range(s.rank)]
        after = [all_hdrs[q] for q in range(s.rank + 1, world)]
        # The other ranks' headers are parsed as a unit of their own and JOINED with the unit of the owned bytes on the GPU
        # (opusfe_unit_join: arrays concatenated and re-based in definition order): the owned gigabyte is parsed once.
        split_pos = sum(int(t.numel()) for t in before)
        htext = torch.cat(before + after + [torch.zeros(64, dtype=torch.uint8, device=s.own.device)])
        n = htext.numel() - 64
        torch.cuda.synchronize()
        import time
        t0 = time.perf_counter()
        uh = s.ctx.get_definitions_device(htext.data_ptr(), n)
        s._hdr_text = htext
        rc = uh.status
        u2 = u
        if rc == F.OK:
            u2 = u.join(uh, split_pos)
        hdr_s = time.perf_counter() - t0       # the price of lowering against whole-program tables: the others' headers, parsed and joined
        if rc == F.OK and lower:
            rc = u2.compile_source()
        results.append(dict(status=rc, unit=u2, own_counts=own_counts, first_bad_rank=None, header_bytes=n, header_reparse_s=hdr_s))
        uh.close()
        if u2 is not u:
            u.close()
    return results
