"""One huge source sharded by byte range over several GPUs (BASELINE config 4).

Every rank holds a contiguous byte range of the source in HBM, cut at ARBITRARY positions.  The sequential lexer state of the
reference (parse.rs:364-391: string/comment context, `ignore_dents`, indentation) is carried across ranks by ONE small all-gather;
each rank then owns whole top-level segments, i.e. a standalone well-formed source:

  1. heads    p2p         cuts move to the next '\n' (a line never splits a run token): the skipped head bytes go to the left neighbour
  2. carry    all-gather  64 bytes per rank (opusfe_shard_carry): composed context-DFA function, token count, kinds of the last / first
                          tokens, net bracket depth, first top-level segment start, length, look-back flag.  It is computed SPECULATIVELY
                          for the entry a shard behind a '\n' almost always has (code context, depth 0); from the gathered carries every
                          rank verifies every rank's speculation, so all reach the same verdict with no further exchange.  Only if it
                          fails (a cut inside a string / inside parentheses, a brace right behind a cut) do they run the dependent chain
                          tails -> depth -> cuts (three more 8..16-byte all-gathers).
  3. prefix   p2p         the bytes before a rank's cut go to the nearest owning rank on its left (sizes known from step 2)
  4. status   all-gather  (status, header length) per rank after parsing
  5. headers  all-gather  definition headers (functions as `extern` signatures, types, globals), variable size -> global tables for IR

The `comm` object hides torch.distributed so that the same code runs W simulated ranks in one process (LocalComm, tests) or one
rank per GPU (sharding.DistComm, bench.py).
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
    """W simulated ranks in one process (tests): a collective returns the list the driver assembled, a point-to-point exchange is a
    dictionary.  Same interface as sharding.DistComm (one rank per process over torch.distributed)."""
    def __init__(self, world):
        self.world = world
        self.collectives = self.p2p_messages = 0

    def gather(self, name, values):
        self.collectives += 1
        return values            # all ranks live in this process

    def gather_sized(self, name, values, sizes):
        self.collectives += 1
        return values

    def p2p(self, name, sends, recvs, ranks):
        """sends[i] = [(dst, tensor)], recvs[i] = [(src, nbytes or None)] for the i-th local rank (ranks[i]) -> received tensors"""
        box = {}
        for r, out in zip(ranks, sends):
            for dst, t in out:
                box[(r, dst)] = t
                self.p2p_messages += 1
        return [[box[(src, r)] for src, _ in inc] for r, inc in zip(ranks, recvs)]


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

    # -- step 2: the whole fixed-size carry, computed for the common entry (context N, depth 0)
    def carry(self):
        self.shard, c = self.ctx.shard_carry(self.eff_padded.data_ptr(), self.n)
        self.spec_state = CS_N
        self.ntok, self.tail = c[1], c[2]
        self._spec_tail = (c[1], c[2])
        return c

    def release_shard(self):
        if self.shard is not None:
            self.ctx.shard_free(self.shard)
            self.shard = None

    def ctx_fn(self):
        return self.ctx.shard_ctx(self.eff_padded.data_ptr(), self.n)

    # -- step 3
    def lex(self, start_state):
        if self.shard is not None and getattr(self, "spec_state", None) == start_state and hasattr(self, "_spec_tail"):
            return self._spec_tail           # the speculative lex was right for this rank
        self.release_shard()
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


def speculation_holds(carries):
    """Every rank checks EVERY rank's speculative carry (opusfe_shard_carry: entry context N, entry depth 0, no brace whose depth delta
    looks back over the cut) from the gathered carries alone, so all ranks reach the same verdict without another exchange."""
    f, depth = CTX_IDENTITY, 0
    for q, c in enumerate(carries):
        if q > 0 and (ctx_apply(f, CS_N) != CS_N or depth != 0 or (c[7] & 1)):
            return False
        f = ctx_compose(f, c[0])
        depth += _i64(c[4])
    return True


def _i64(v):
    return v - (1 << 64) if v >= (1 << 63) else v


NO_CUT = (1 << 64) - 1


def plan(states, comm):
    """Steps 1-6 for the ranks in `states` (all ranks when simulated, one rank under torch.distributed).

      heads   p2p      the bytes up to a rank's first '\n' go to its left neighbour (a line never splits a token)
      carry   ONE all-gather of 8 words per rank (opusfe_shard_carry), verified by every rank for every rank
      prefix  p2p      the bytes before a rank's cut go to the nearest owning rank on its left (sizes known from the carries)

    Only when the speculation of some rank fails (a cut inside a string or inside parentheses, a brace right behind a cut) do the
    ranks run the dependent chain: tails -> depth -> cuts, three more small all-gathers."""
    if callable(comm) and not hasattr(comm, "gather"):       # (a bare gather function: the simulated ranks of older callers)
        g = comm
        comm = LocalComm(states[0].world)
        comm.gather = g
    world = states[0].world
    ranks = [s.rank for s in states]
    heads = [s.head() for s in states]
    got = comm.p2p("heads", [[(s.rank - 1, h)] if s.rank > 0 else [] for s, h in zip(states, heads)],
                   [[(s.rank + 1, None)] if s.rank + 1 < world else [] for s in states], ranks)
    for s, g in zip(states, got):
        s.set_next_head(g[0] if g else None)
    carries = comm.gather("carry", [tuple(s.carry()) for s in states])
    carries = [[int(x) & ((1 << 64) - 1) for x in c] for c in carries]
    fast = speculation_holds(carries)
    for s in states:
        s.speculation_held = fast
    if fast:
        cuts = [0] + [c[6] if c[5] == NO_CUT else c[5] for c in carries[1:]]
        lens = [c[6] for c in carries]
        for s in states:
            s.c = cuts[s.rank]
            s.release_shard()
    else:
        fns = [c[0] for c in carries]
        for s in states:
            f = CTX_IDENTITY
            for q in range(s.rank):
                f = ctx_compose(f, fns[q])
            s.start_state = ctx_apply(f, CS_N)
        tails = comm.gather("tails", [s.lex(s.start_state) for s in states])
        for s in states:
            prev = TK_NO_PREV
            for q in range(s.rank):
                prev = fold_tail(prev, tails[q][0], tails[q][1])
            s.prev_kinds = prev
        nets = comm.gather("depth", [s.net_depth(s.prev_kinds) for s in states])
        cl = comm.gather("cuts", [(s.cut(sum(int(x) for x in nets[:s.rank])), s.n) for s in states])
        cuts, lens = [int(x[0]) for x in cl], [int(x[1]) for x in cl]
    # the bytes before a rank's cut belong to the nearest owning rank on its left: point to point, sizes known to both sides
    owns = [q == 0 or cuts[q] < lens[q] for q in range(world)]

    def owner_left(q):
        p = q - 1
        while p > 0 and not owns[p]:
            p -= 1
        return p
    sends, recvs = [], []
    for s in states:
        sends.append([(owner_left(s.rank), s.prefix())] if s.rank > 0 and cuts[s.rank] > 0 else [])
        inc = []
        if owns[s.rank]:
            q = s.rank + 1
            while q < world:
                if cuts[q] > 0:
                    inc.append((q, cuts[q]))
                if owns[q]:
                    break
                q += 1
        recvs.append(inc)
    got = comm.p2p("prefixes", sends, recvs, ranks)
    for s, right in zip(states, got):
        if not owns[s.rank]:
            s.own = s.eff[:0]
        else:
            s.set_own(right)
    return states


def _effective_status(u, own_len):
    """Status of a rank's unit as part of the whole source.  A unit that needed resync passes (swallowed lexical errors)
    was lexed with contexts the carries did not know; it composes with the ranks behind it exactly when the reference's lexer is back
    in code context at bracket depth 0 at its end (opusfe_unit_end_state), and its syntax error stands only if it does not touch the
    unit's end (there the whole source continues where this unit stops)."""
    st = u.status
    if st not in (F.OK, F.SYNTAX) or u.resync_passes() == 0:
        return st
    if st == F.OK:
        state, depth = u.end_state()
        return F.OK if state in (0, 5) and depth == 0 else F.UNSUPPORTED
    e = u.error()
    if e is None or int(e.low) + int(e.len) >= own_len or int(e.tok_kind) == 59 or int(e.kind) == 1:      # EndOfFile token / UnexpectedEndOfFile
        return F.UNSUPPORTED
    return F.SYNTAX


def compile_sharded(states, comm, lower=True):
    """Steps 6-8: parse the owned bytes, exchange definition headers, lower the own definitions between the others' headers.
    Returns per state a dict(status, unit, own_counts) — `unit` is the joined unit (headers of the ranks before, own
    definitions, headers of the ranks after).  Two collectives: (status, header length) of every rank, then the header text."""
    if callable(comm) and not hasattr(comm, "gather"):
        g = comm
        comm = LocalComm(states[0].world)
        comm.gather = g
        comm.gather_sized = lambda name, values, sizes: g(name, values)
    world = states[0].world
    parsed, hdrs, info = [], [], []
    for s in states:
        buf = torch.cat([s.own, torch.zeros(64, dtype=torch.uint8, device=s.own.device)])
        torch.cuda.synchronize()
        u = s.ctx.get_definitions_device(buf.data_ptr(), s.own.numel())
        s._own_buf = buf
        parsed.append(u)
        st = _effective_status(u, s.own.numel())
        t = torch.empty(0, dtype=torch.uint8, device=s.own.device)
        if st == F.OK:
            p, n = u.headers_device()
            t = torch.empty(n, dtype=torch.uint8, device=s.own.device)
            if n:
                rc = _hip().hipMemcpy(C.c_void_p(t.data_ptr()), C.c_void_p(p), C.c_size_t(n), 3)   # device to device
                if rc != 0:
                    raise RuntimeError(f"hipMemcpy of the definition headers failed: hip error {rc}")
        hdrs.append(t)
        info.append((int(st), int(t.numel())))
    info = comm.gather("status", info)
    statuses = [int(x[0]) for x in info]
    if any(st != F.OK for st in statuses):
        first = next(r for r, st in enumerate(statuses) if st != F.OK)
        return [dict(status=statuses[first], unit=u, ir=None, first_bad_rank=first) for u in parsed]
    all_hdrs = comm.gather_sized("headers", hdrs, [int(x[1]) for x in info])
    results = []
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
