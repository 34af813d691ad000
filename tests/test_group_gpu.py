"""GPU: a corpus of independent files answered by ONE packed pass (opusfe_parse_group; BASELINE config 5).
Every file of every group is compared with the oracle run on THAT FILE ALONE: the status, the first-error tuple (positions
relative to the file), the token text, the AST text, the IR text and the ordered diagnostics.  Nothing of one file may show in
another: the groups below hold files that define the same names with different types, files with every injected fault kind,
empty / comment-only / indented files, files with finalize-time errors and files that define operators."""
import hashlib
import math
import random

import pytest
import torch          # before the first opusfe context: the process then has ONE HIP runtime (torch's), which libopusfe.so binds to

import oracle_lib as O

pytestmark = pytest.mark.gpu


def check_file(g, f, src, full=True):
    want_err = O.dump(src, O.ERROR)
    st = g.status(f)
    if want_err == "Unsupported\n":
        assert st == 2, (f, st)
        return "unsupported"
    if want_err != "Ok\n":
        assert st == 1, (f, st, want_err, src[:80])
        assert g.dump_error(f) == want_err, (f, src)
        e = g.error(f)
        assert e is not None and e.low <= len(src)
        if full:
            assert g.dump_tokens(f) == O.dump(src, O.TOKENS), (f, src)
        return "syntax"
    want_ir, want_sem = O.dump(src, O.IR), O.dump(src, O.SEM)
    if want_ir == "" and want_sem != "":
        assert st == 7, (f, st, src)
        assert g.dump_semantic_errors(f) == want_sem
        return "semantic"
    assert st == 0, (f, st, g.ctx.message(), src[:200])
    if full:
        assert g.dump_tokens(f) == O.dump(src, O.TOKENS), (f, src)
        assert g.dump_ast(f) == O.dump(src, O.AST), (f, src)
        assert g.dump_ir(f) == want_ir, (f, src)
    else:
        assert hashlib.sha256(g.dump_ir(f).encode()).digest() == hashlib.sha256(want_ir.encode()).digest(), f
    assert g.dump_semantic_errors(f) == want_sem, (f, src)
    return "ok"


def check_group(fe, files, full=True, lower=True):
    g = fe.compile_group(files, lower=lower)
    assert len(g) == len(files)
    seen = {}
    for f, src in enumerate(files):
        k = check_file(g, f, src, full)
        seen[k] = seen.get(k, 0) + 1
    stats = g.stats()
    assert stats[0] + stats[1] == len(files)
    g.close()
    return seen, stats


def test_two_clean_files_share_one_pass(fe):
    a = O.generate(11, 4000)
    b = O.generate(12, 6000)
    seen, stats = check_group(fe, [a, b])
    assert seen == {"ok": 2} and stats == (2, 0, 0)


def test_same_names_different_meaning_do_not_leak(fe):
    """every file defines `f`, `T` and `g` differently; a name that only another file defines must stay undefined"""
    files = [
        b"type T = int32\nf(a T) T\n\treturn a + 1\ng() T\n\treturn f(2)\n",
        b"type T = bool\nf(a T) T\n\treturn not a\ng() T\n\treturn f(true)\n",
        b"f(a float64) float64\n\treturn a\ng() float64\n\treturn f(1.5)\nh() null\n\tvar x T = 1\n",      # T undefined HERE
        b"var f int32 = 3\ng() int32\n\treturn f\n",
        b"g() int32\n\treturn f(1)\n",                                                                          # f undefined HERE
        b"const K int32 = 5\nf() int32\n\treturn K\n",
        b"f() int32\n\treturn K\n",                                                                             # K undefined HERE
    ]
    seen, stats = check_group(fe, files)
    assert stats[2] == 0


def test_every_fault_kind_inside_a_group(fe):
    files = []
    for i in range(40):
        src = O.generate(81_000 + i, 1500 + 211 * i)
        if i % 2:
            src = O.inject(src, 81_000 + i, (i // 2) % 10) or src
        files.append(src)
    seen, stats = check_group(fe, files)
    assert seen.get("syntax", 0) >= 15 and seen.get("ok", 0) >= 15
    assert stats[0] >= 30                     # the faulty files stay on the packed path: a syntax error is a result, not an irregularity


def test_edge_files(fe):
    files = [
        b"",
        b"; only a comment\n",
        b"\n\n\n",
        b"f() int32\n\treturn 1\n",
        b"\tf() int32\n\t\treturn 1\n",          # first line indented
        b"f() int32\n\treturn 1",                  # no newline at the end of the file
        b"f() int32\n\treturn (1 +\n",             # open bracket at the end of the file
        b'f() null\n\tvar s = "abc',               # unterminated string at the end of the file
        b"f() int32\n\treturn 1\n\n\n\n",
        b"f() float64\n\treturn 1.5f",             # the reference hangs on this one alone (Unsupported); its neighbours must not notice
        b"g() int32\n\treturn 2\n",
        b"operator +(a bool, b bool) bool\n\treturn a\nf() bool\n\treturn true + false\n",      # defines a name by an operator symbol
        b"f() bool\n\treturn true + false\n",                                                    # ... which this file must NOT see
        b"type A = B\ntype B = A\nf() null\n\treturn null\n",                                    # finalize-time error
        b"f(a int32 null\n",
        b"x" * 63, b"y" * 64, b"z" * 65,          # lengths around the packing granule: the last byte of a file next to the padding
        b"f() int32\n\treturn 1 ;" + b"c" * 100,  # a comment that runs to the end of the file
        b"f() null\n\treturn null\n" + b";" * 64,
    ]
    seen, stats = check_group(fe, files)
    assert 3 <= stats[1] <= 8, stats          # empty, indented, unsupported and finalize-error files took the single-file path, not the rest


def test_swallowed_lexical_errors_inside_a_group(fe):
    """trap mutants (swallowed lexical errors that need resync passes) next to clean files"""
    import mutate as M
    traps = M.trap_mutants(30, seed0=91_000)
    files = []
    for i, t in enumerate(traps):
        files.append(O.generate(92_000 + i, 2000))
        files.append(t)
    check_group(fe, files)


def test_semantic_mutants_inside_groups(fe):
    import mutate as M
    srcs = M.mutants(300, seed0=93_000) + M.byte_mutants(200, seed0=93_000)
    seen, stats = check_group(fe, srcs)
    assert seen.get("ok", 0) > 100 and seen.get("syntax", 0) > 30


def test_group_without_lowering(fe):
    files = [O.generate(94_000 + i, 3000) for i in range(8)]
    files[3] = O.inject(files[3], 1, 2) or files[3]
    g = fe.compile_group(files, lower=False)
    for f, src in enumerate(files):
        want_err = O.dump(src, O.ERROR)
        if want_err != "Ok\n":
            assert g.status(f) == 1 and g.dump_error(f) == want_err
        else:
            assert g.status(f) == 0
        assert g.dump_tokens(f) == O.dump(src, O.TOKENS)
    g.close()


def test_device_resident_group_and_ranges(fe):
    """opusfe_parse_group_device over a buffer packed by the caller; the per-file ranges index the packed unit's arrays"""
    from opus_amd import frontend as F
    files = [O.generate(95_000 + i, 2500 + 300 * i) for i in range(12)]
    packed, starts, lens, total = F.pack_group(files)
    d = torch.frombuffer(bytearray(packed), dtype=torch.uint8).cuda()
    g = fe.compile_group_device(d.data_ptr(), total, starts, lens)
    unit = g.unit()
    toks = unit.tokens()
    for f, src in enumerate(files):
        assert check_file(g, f, src) == "ok"
        r = g.ranges(f)
        assert r[8] == starts[f] and r[0] < r[1] and r[2] < r[3] and r[4] < r[5]
        assert all(starts[f] <= toks[k].low <= starts[f] + lens[f] for k in range(r[0], r[1]))
        if f:
            assert r[0] == g.ranges(f - 1)[1] and r[2] == g.ranges(f - 1)[3] and r[4] == g.ranges(f - 1)[5]
    # a layout the entry point must refuse: unaligned start
    with pytest.raises(F.OpusfeError):
        fe.compile_group_device(d.data_ptr(), total, [0, starts[1] + 1], lens[:2])
    g.close()


def test_config5_shape_through_the_packed_path(fe):
    """BASELINE config 5 in small: 1000 files, sizes log-uniform in [1 KB, 64 KB], i = 7 (mod 100) carries exactly one injected
    fault (all ten kinds in turn): error tuple for the faulty files, hash of the IR text for the rest"""
    rnd = random.Random(55)
    files = []
    for i in range(1000):
        size = int(math.exp(rnd.uniform(math.log(1024), math.log(65536))))
        src = O.generate(5_000_000 + i, size)
        if i % 100 == 7:
            src = O.inject(src, 5_000_000 + i, (i // 100) % 10) or src
        files.append(src)
    seen, stats = check_group(fe, files, full=False)
    assert seen.get("syntax", 0) >= 8
    assert stats[1] <= 10, stats              # the generated corpus is the fast path: (nearly) every file is answered by the packed pass
