"""GPU: one source sharded at ARBITRARY byte positions over W simulated ranks (the carry chain of opus_amd/shard.py run in
one process on one GPU).  The concatenated per-rank IR must equal the oracle's IR of the whole source, and every rank's cut
must be a real top-level segment start."""
import pytest
import torch

import oracle_lib as O

pytestmark = pytest.mark.gpu


def run_sharded(fe, src: bytes, world: int, cuts=None):
    from opus_amd import shard
    n = len(src)
    if cuts is None:
        cuts = [n * r // world for r in range(world)] + [n]
    whole = torch.frombuffer(bytearray(src), dtype=torch.uint8).cuda()
    states = [shard.RankState(fe, whole[cuts[r]:cuts[r + 1]].clone(), r, world) for r in range(world)]

    comm = shard.LocalComm(world)
    shard.plan(states, comm)
    owned = b"".join(bytes(s.own.cpu().numpy()) for s in states)
    assert owned == src, "the owned ranges must tile the source"
    # when every shard starts in code context at depth 0 the carries travel in ONE all-gather, the halos point to point
    if states[0].speculation_held:
        assert comm.collectives == 1, comm.collectives
    else:
        assert comm.collectives == 4, comm.collectives       # carry + the dependent chain tails -> depth -> cuts
    res = shard.compile_sharded(states, comm)
    assert comm.collectives <= 6                             # + (status, header length) and the header text
    states[0].comm = comm
    return states, res


@pytest.mark.parametrize("world", [2, 3, 8])
@pytest.mark.parametrize("seed,size", [(3, 600_000), (12, 250_000)])
def test_sharded_ir_equals_whole_file_ir(fe, seed, size, world):
    src = O.generate(seed, size)
    states, res = run_sharded(fe, src, world)
    assert all(r["status"] == 0 for r in res), [r["status"] for r in res]
    got = "".join(r["unit"].dump_ir() for r in res)
    want = O.dump(src, O.IR)
    assert got == want
    # every cut is a line start at column 0 that opens a definition
    pos = 0
    for s in states:
        if s.own.numel() and pos > 0:
            assert src[pos - 1:pos] == b"\n" and src[pos:pos + 1] not in (b"\t", b" ", b"\n", b";")
        pos += s.own.numel()


def test_cuts_inside_strings_comments_and_parentheses(fe):
    body = ("f(a int32) int32\n\tvar s = \"line one\nfake() null\n\tnot code\n\"\n\tvar t = g(\na,\nb)\n\t; comment with \" quote\n"
            "\treturn a\ng(a int32, b int32) int32\n\treturn a + b\n")
    src = (body.replace("f(", "f0(") + "".join(body.replace("f(", f"f{i}(").replace("g(a int32, b", f"g{i}(a int32, b").replace("g(\n", f"g{i}(\n")
                                               for i in range(1, 40))).encode()
    src = ("g(a int32, b int32) int32\n\treturn a\n").encode() + b"".join(
        (f"f{i}(a int32) int32\n\tvar s = \"line one\nfake() null\n\tnot code\n\"\n\tvar t = g(\na,\nb)\n\t; comment with \" quote\n\treturn a\n").encode()
        for i in range(60))
    assert O.dump(src, O.ERROR) == "Ok\n"
    want = O.dump(src, O.IR)
    for world in (2, 5):
        for shift in (0, 7, 23, 41):
            n = len(src)
            cuts = [0] + [min(n, n * r // world + shift) for r in range(1, world)] + [n]
            states, res = run_sharded(fe, src, world, cuts)
            assert all(r["status"] == 0 for r in res)
            assert "".join(r["unit"].dump_ir() for r in res) == want


def test_generated_source_takes_the_single_all_gather_path(fe):
    """cuts at arbitrary bytes of a generated source start (after the head bytes moved left) in code context at depth 0: the fast path"""
    src = O.generate(3, 600_000)
    for world in (2, 8):
        states, res = run_sharded(fe, src, world)
        assert states[0].speculation_held and all(r["status"] == 0 for r in res)
        for r in res:
            r["unit"].close()


def test_syntax_error_is_reported_by_the_owning_rank(fe):
    src = O.generate(5, 200_000)
    bad = O.inject(src, 77, 0)
    want = O.dump(bad, O.ERROR)
    assert want != "Ok\n"
    states, res = run_sharded(fe, bad, 4)
    first = res[0]["first_bad_rank"]
    assert first is not None
    u = res[first]["unit"]
    e = u.error()
    offset = sum(s.own.numel() for s in states[:first])
    kind, low = want.split()[0], int(want.split()[1])
    assert u.dump_error().split()[0] == kind
    assert e.low + offset == low


def test_large_source_properties_whole_vs_sharded(fe):
    """Size-independent properties at a size the text oracle no longer handles in seconds (64 MB, deep regions injected):
    lowering the whole source, lowering it again, and lowering it as 4 byte-range shards must agree on every count that
    does not depend on the sharding (tokens and nodes of the owned bytes, instructions, variables, basic blocks)."""
    src = O.generate(21, 64 << 20, flags=1, threads=8)
    u = fe.get_definitions(src)
    assert u.status == 0 and u.compile_source() == 0
    whole = u.counts()
    u.close()
    u2 = fe.get_definitions(src)
    assert u2.status == 0 and u2.compile_source() == 0
    assert u2.counts() == whole, "the pipeline must be deterministic"
    u2.close()
    states, res = run_sharded(fe, src, 4)
    assert all(r["status"] == 0 for r in res)
    # the EndOfFile token belongs to the last rank only; every other rank's own unit ends with one of its own
    assert sum(r["own_counts"]["tokens"] for r in res) == whole["tokens"] + (len(res) - 1)
    assert sum(r["own_counts"]["nodes"] for r in res) == whole["nodes"]
    for key in ("instrs", "vars", "bbs"):
        assert sum(r["unit"].counts()[key] for r in res) == whole[key], key
    for r in res:
        r["unit"].close()


def test_unit_join_equals_parsing_the_concatenated_text(fe):
    """opusfe_unit_join(own, hdr, split) must lower exactly like one parse of before + own + after: literal pools of the
    header unit (big integers, strings), type definitions, globals, overloads on both sides of the owned text."""
    before = ('type Pair = record { a int32 b float64 }\nconst BIG nat64 = 18446744073709551615\nvar PI float64\n'
              'library "before lib"\ntype Arr = {12} nat16\nextern ext1(x int32) int32\nextern over(x int32) int32\nvar counter int32 = 7\n')
    own = ('over(x int64) int64\n\treturn x + 1i64\n'
           'f(p Pair, q @mut Pair) float64\n\tq@.a = ext1(p.a) + over(counter) + tail(2)\n\tvar big = BIG\n\tvar w Wide\n\treturn p.b * PI + SCALE\n'
           'g() int64\n\treturn over(5i64)\n')
    after = ('type Wide = record { p Pair n nat64 }\nvar SCALE float64\nlibrary "after lib"\nextern tail(x int32) int32\nconst LAST nat64 = 4294967296\n')
    whole = (before + own + after).encode()
    assert O.dump(whole, O.ERROR) == "Ok\n"
    a = fe.get_definitions(own.encode())
    h = fe.get_definitions((before + after).encode())
    assert a.status == 0 and h.status == 0
    j = a.join(h, len(before))
    assert j.compile_source() == 0, fe.message()
    assert j.dump_ir() == O.dump(whole, O.IR)
    assert j._dump(4) == O.dump(whole, O.SEM)
    # degenerate splits: everything before / everything after / no headers at all
    for text, split in ((before + after, len(before + after)), (before + after, 0), ("", 0)):
        h2 = fe.get_definitions(text.encode())
        j2 = a.join(h2, split)
        c = j2.counts()
        assert c["definitions"] == a.counts()["definitions"] + h2.counts()["definitions"]
        j2.close(); h2.close()


def test_shard_with_a_swallowed_lexical_error_that_reconverges(fe):
    """A swallowed bad escape at a top-level position: the reference resumes lexing INSIDE the string body.  Here the body
    is `;"`: the reference reads a comment up to the end of the line, the context automaton closes the string at the quote, and both
    are back in code context on the next line.  The rank that owns the trap follows the reference with a resync pass, its unit ends
    in code context at depth 0 (opusfe_unit_end_state), so the sharded result is exact: the IR of the whole source."""
    a, b = O.generate(5, 150_000), O.generate(6, 150_000)
    src = a + b'"ab\\q;"\n' + b
    assert O.dump(src, O.ERROR) == "Ok\n"
    whole = fe.get_definitions(src)
    assert whole.status == 0 and whole.resync_passes() >= 1
    whole.close()
    states, res = run_sharded(fe, src, 4)
    assert all(r["status"] == 0 for r in res), [r["status"] for r in res]
    assert "".join(r["unit"].dump_ir() for r in res) == O.dump(src, O.IR)


def test_shard_with_a_swallowed_lexical_error_that_diverges_is_never_answered_wrongly(fe):
    """The same trap without the closing quote: for the context automaton the quote opens a string that swallows the code behind it,
    so the carries of every rank behind the trap describe an inverted source (the reference itself just goes on with the next line,
    Q7).  Whatever the cuts, the sharded path may answer exactly what the reference answers — here the ranks behind the trap find no
    provable cut, their bytes travel to the rank that owns the trap, which follows the reference with a resync pass — or refuse
    (UNSUPPORTED); it must never answer anything else."""
    a, b = O.generate(5, 150_000), O.generate(6, 150_000)
    for trap, world in ((b'"ab\\q\n', 4), (b'"ab\\q\n', 3), (b'var s = 1\n"ab\\q " "\n', 4)):
        src = a + trap + b
        want = O.dump(src, O.ERROR)
        whole = fe.get_definitions(src)
        assert whole.dump_error() == want          # the single-GPU answer is the reference's
        whole.close()
        states, res = run_sharded(fe, src, world)
        first = res[0]["first_bad_rank"]
        if first is None:
            assert want == "Ok\n" and all(r["status"] == 0 for r in res)
            assert "".join(r["unit"].dump_ir() for r in res) == O.dump(src, O.IR)
            continue
        st = res[0]["status"]
        assert st in (1, 2)
        if st == 1:
            assert want not in ("Ok\n", "Unsupported\n")
            e = res[first]["unit"].error()
            offset = sum(s.own.numel() for s in states[:first])
            assert res[first]["unit"].dump_error().split()[0] == want.split()[0] and int(e.low) + offset == int(want.split()[1])


def test_two_process_gloo_run_matches_the_whole_source(fe, tmp_path):
    """The real thing in two processes: `bench.py --gpus 2` under torch.distributed (gloo, both ranks on this one GPU) runs plan()
    and compile_sharded() with real gathers between processes; what the two ranks produced together must be what the same
    logical source gives as ONE unit."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    nbytes, nblk = 16_000_000, 16         # (the generator needs blocks of at least 1 MB for the depth-1000 regions)
    blk = nbytes // nblk
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29571",
           os.path.join(root, "bench.py"), "--gpus", "2", "--backend", "gloo", "--bytes", str(nbytes), "--steps", "1", "--warmup", "0", "--no-cpu"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=root)
    assert out.returncode == 0, "\n".join(l for l in out.stderr.splitlines() if "rank" in l)[-3000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["sharded_totals"] is not None
    tot = d["sharded_totals"]
    whole = O.generate_range(3, 0, 2 * nblk, blk, flags=1)
    assert tot["source_bytes"] == len(whole) == 2 * blk * nblk
    u = fe.get_definitions(whole)
    assert u.compile_source() == 0
    c = u.counts()
    assert tot["tokens"] == c["tokens"] + 1            # every rank's unit ends with an EndOfFile token of its own
    assert (tot["nodes"], tot["definitions_own"], tot["instrs"], tot["vars"]) == (c["nodes"], c["definitions"], c["instrs"], c["vars"])
