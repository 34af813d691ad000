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


def test_two_rank_gloo_variable_size_gather(tmp_path):
    script = tmp_path / "g.py"
    script.write_text(textwrap.dedent(f"""
        import sys
        sys.path.insert(0, {ROOT!r})
        import torch, torch.distributed as dist
        from opus_amd import sharding
        dist.init_process_group("gloo")
        r, w = dist.get_rank(), dist.get_world_size()
        gather = sharding.make_gather(dist, torch.device("cpu"))
        objs = gather("o", [(r, "x" * r)])
        assert objs == [(q, "x" * q) for q in range(w)], objs
        t = torch.full((10 + 1000 * r,), 7 + r, dtype=torch.uint8)
        got = gather("t", [t])
        assert [g.numel() for g in got] == [10 + 1000 * q for q in range(w)]
        assert all(int(g[0]) == 7 + q and int(g[-1]) == 7 + q for q, g in enumerate(got))
        empty = gather("e", [torch.empty(0, dtype=torch.uint8)])
        assert all(g.numel() == 0 for g in empty)
        dist.barrier()
        if r == 0: print("OK")
    """))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                          "--master-port", "29612", str(script)], capture_output=True, text=True, env=env, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "OK" in out.stdout
