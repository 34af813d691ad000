"""GPU: opusfe_shard_ctx_static (the entry point that allocates nothing) against opusfe_shard_ctx, through the C ABI.
Two calls in a row, the second with a smaller input: the static scratch keeps the first call's contents and the result
must not depend on them.  Bit-exact (an 18-bit composed transition function)."""
import pytest
import torch

import oracle_lib as O

pytestmark = pytest.mark.gpu


def ctx_fn_cpu(src: bytes) -> int:
    """the composed context-automaton function of lex.h by a plain loop (checker for tiny inputs)"""
    N, S, E, X1, X2, Cc = range(6)
    def fn_of(b):
        if b == 0x22: return (S, N, S, X2, S, Cc)
        if b == 0x5C: return (N, E, S, X2, S, Cc)
        if b == 0x3B: return (Cc, S, S, X2, S, Cc)
        if b == 0x0A: return (N, S, S, X2, S, N)
        if b == 0x78: return (N, S, X1, X2, S, Cc)
        return (N, S, S, X2, S, Cc)
    f = list(range(6))
    for b in src:
        g = fn_of(b)
        f = [g[s] for s in f]
    return sum(v << (3 * i) for i, v in enumerate(f))


def test_static_entry_point_matches_the_allocating_one(fe):
    big = O.generate(31, 3_000_000) + b'var s = "open string ; \\x4'          # ends inside an escape: a non-trivial function
    small = b'f() null\n\tvar t = "a;b" ; comment "\n' * 700 + b'"tail'
    tiny = b'x "\\'
    out = torch.zeros(4, dtype=torch.int32, device="cuda")
    stream = torch.cuda.ExternalStream(fe.stream())
    for k, src in enumerate((big, small, tiny, b"")):
        d = torch.frombuffer(bytearray(src) + bytearray(16), dtype=torch.uint8).cuda()
        torch.cuda.synchronize()
        want = fe.shard_ctx(d.data_ptr(), len(src))
        held = fe.memory_held()
        fe.shard_ctx_static(d.data_ptr(), len(src), out.data_ptr() + 4 * k)
        assert fe.memory_held() == held, "the static entry point took memory from the context's allocator"
        stream.synchronize()
        got = int(out[k].item()) & 0xFFFFFFFF
        assert got == want, (k, hex(got), hex(want))
        if len(src) < 100_000:
            assert got == ctx_fn_cpu(src)


def test_static_entry_point_rejects_ranges_beyond_its_scratch(fe):
    import opus_amd
    out = torch.zeros(1, dtype=torch.int32, device="cuda")
    d = torch.zeros(64, dtype=torch.uint8, device="cuda")
    with pytest.raises(opus_amd.OpusfeError) as e:
        fe.shard_ctx_static(d.data_ptr(), (1 << 30) + 1, out.data_ptr())     # rejected before anything is launched
    assert e.value.code == opus_amd.frontend.ERR_ARG and "1 GiB" in str(e.value)


def test_static_entry_point_on_a_context_that_never_allocated():
    """the property the entry point exists for: a fresh context (no allocating call before it) serves it, and holds no device
    memory afterwards"""
    import opus_amd
    src = b'f() null\n\tvar t = "a;b" ; comment "\n' * 300 + b'"tail'
    d = torch.frombuffer(bytearray(src) + bytearray(16), dtype=torch.uint8).cuda()
    out = torch.zeros(1, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    c = opus_amd.Context(0)
    try:
        assert c.memory_held() == 0
        c.shard_ctx_static(d.data_ptr(), len(src), out.data_ptr())
        torch.cuda.ExternalStream(c.stream()).synchronize()
        assert c.memory_held() == 0
        assert (int(out[0].item()) & 0xFFFFFFFF) == ctx_fn_cpu(src)
    finally:
        c.close()

