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
