"""CPU (gloo, world_size 2): the N>1 host logic — corpus sharding by cumulative bytes, max-over-ranks timing reduction."""
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from opus_amd import sharding


def test_balance_by_cumulative_bytes_is_a_partition():
    sizes = [1024 + (i * 7919) % (63 * 1024) for i in range(1000)]
    for world in (1, 2, 4, 8):
        parts = [sharding.shard_by_bytes(sizes, r, world) for r in range(world)]
        flat = sorted(i for p in parts for i in p)
        assert flat == list(range(len(sizes)))
        loads = [sum(sizes[i] for i in p) for p in parts]
        assert max(loads) - min(loads) <= 2 * max(sizes)


def test_two_rank_gloo_reduction_and_sharding(tmp_path):
    script = tmp_path / "w.py"
    script.write_text(textwrap.dedent(f"""
        import os, sys
        sys.path.insert(0, {ROOT!r})
        import torch, torch.distributed as dist
        from opus_amd import sharding
        dist.init_process_group("gloo")
        r, w = dist.get_rank(), dist.get_world_size()
        sizes = [1000 + 37 * i for i in range(101)]
        mine = sharding.shard_by_bytes(sizes, r, w)
        total = sharding.all_reduce_sum(float(sum(sizes[i] for i in mine)), dist)
        slow = sharding.all_reduce_max(1.0 + r, dist)
        assert total == float(sum(sizes)), total
        assert slow == float(w), slow
        dist.barrier()
        if r == 0: print("OK", len(mine))
    """))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                          "--master-port", "29611", str(script)], capture_output=True, text=True, env=env, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "OK" in out.stdout


def test_carry_folds_are_associative_and_match_the_device_encoding():
    from opus_amd import shard
    # context functions: composing shard functions left to right == the function of the concatenation
    import random
    rnd = random.Random(1)
    def fn(n, s, e, x1, x2, c):
        return n | (s << 3) | (e << 6) | (x1 << 9) | (x2 << 12) | (c << 15)
    table = {'"': fn(1, 0, 1, 4, 1, 5), "\\": fn(0, 2, 1, 4, 1, 5), ";": fn(5, 1, 1, 4, 1, 5), "\n": fn(0, 1, 1, 4, 1, 0), "x": fn(0, 1, 3, 4, 1, 5), "a": fn(0, 1, 1, 4, 1, 5)}
    text = "".join(rnd.choice('"\\;\nxa') for _ in range(400))
    def whole(t):
        f = shard.CTX_IDENTITY
        for ch in t:
            f = shard.ctx_compose(f, table[ch])
        return f
    for cut in (0, 1, 57, 199, 400):
        assert shard.ctx_compose(whole(text[:cut]), whole(text[cut:])) == whole(text)
    # token-kind tails: folding shard by shard == the tail of the concatenated kind sequence
    kinds = [rnd.randrange(59) for _ in range(50)]
    def tail(seq):
        out = shard.TK_NO_PREV
        for j, k in enumerate(reversed(seq[-8:])):
            out = (out & ~(0xFF << (8 * j))) | (k << (8 * j))
        return out
    for cuts in ([0, 50], [0, 3, 50], [0, 3, 5, 6, 6, 30, 50]):
        prev = shard.TK_NO_PREV
        for a, b in zip(cuts, cuts[1:]):
            part = kinds[a:b]
            prev = shard.fold_tail(prev, len(part), tail(part))
        assert prev == tail(kinds)


def test_two_rank_gloo_comm_gathers_and_point_to_point(tmp_path):
    """sharding.DistComm over gloo, world 2: the fixed-size integer all-gather (u64 words included), the sized byte all-gather, and
    the point-to-point transfers of known and unknown size (a head longer than the first message included)"""
    script = tmp_path / "g.py"
    script.write_text(textwrap.dedent(f"""
        import sys
        sys.path.insert(0, {ROOT!r})
        import torch, torch.distributed as dist
        from opus_amd import sharding
        dist.init_process_group("gloo")
        r, w = dist.get_rank(), dist.get_world_size()
        comm = sharding.DistComm(dist, torch.device("cpu"))
        rows = comm.gather("carry", [(r, 0x3B3B3B3B3B3B3B3B, (1 << 64) - 1, -5)])
        assert rows == [(q, 0x3B3B3B3B3B3B3B3B, -1, -5) for q in range(w)], rows          # u64 words travel as two's complement
        assert comm.gather("one", [7 + r]) == [7, 8]
        t = torch.full((10 + 1000 * r,), 7 + r, dtype=torch.uint8)
        got = comm.gather_sized("t", [t], [10 + 1000 * q for q in range(w)])
        assert [g.numel() for g in got] == [10 + 1000 * q for q in range(w)]
        assert all(int(g[0]) == 7 + q and int(g[-1]) == 7 + q for q, g in enumerate(got))
        assert comm.collectives == 3
        # heads: rank 1 -> rank 0, size unknown to the receiver; once short, once longer than the first message
        for n in (5, sharding.HEAD_CAP + 123, 0):
            payload = (torch.arange(n, dtype=torch.int64) % 251).to(torch.uint8)
            got = comm.p2p("heads", [[(0, payload)] if r == 1 else []], [[(1, None)] if r == 0 else []], [r])
            if r == 0:
                assert torch.equal(got[0][0].cpu(), payload), n
        # prefixes: known sizes, both directions at once
        mine = torch.full((100 + r,), r, dtype=torch.uint8)
        got = comm.p2p("prefixes", [[(1 - r, mine)]], [[(1 - r, 100 + (1 - r))]], [r])
        assert got[0][0].numel() == 100 + (1 - r) and int(got[0][0][0]) == 1 - r
        dist.barrier()
        if r == 0: print("OK")
    """))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                          "--master-port", "29612", str(script)], capture_output=True, text=True, env=env, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "OK" in out.stdout


def test_speculation_check_is_a_pure_function_of_the_gathered_carries():
    """every rank must reach the same verdict from the same carries: entry context by folding the context functions, entry depth
    by summing the nets, and the look-back flag of every rank but the first"""
    from opus_amd import shard
    ident, quote = shard.CTX_IDENTITY, 1 | (0 << 3) | (1 << 6) | (4 << 9) | (1 << 12) | (5 << 15)      # lex.h CTX_QUOTE
    ok = [[ident, 10, 0, 0, 0, 5, 100, 0], [ident, 10, 0, 0, 0, 5, 100, 0], [ident, 3, 0, 0, 0, 2, 50, 0]]
    assert shard.speculation_holds(ok)
    first_flag = [row[:] for row in ok]; first_flag[0][7] = 1
    assert shard.speculation_holds(first_flag)                      # nothing lies before rank 0
    for q, (col, val) in ((1, (7, 1)), (0, (4, 1)), (0, (0, quote)), (1, (4, (1 << 64) - 1))):
        bad = [row[:] for row in ok]
        bad[q][col] = val
        assert not shard.speculation_holds(bad), (q, col)
    both = [row[:] for row in ok]; both[0][4] = 1; both[1][4] = (1 << 64) - 1      # +1 then -1: rank 1 starts at depth 1, rank 2 at 0
    assert not shard.speculation_holds(both)
