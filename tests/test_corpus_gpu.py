"""GPU: corpus of independent files (BASELINE config 5) — per-file IR / first-error parity with the oracle."""
import hashlib

import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu


def make_corpus(n, seed0=5_000_000):
    files = []
    for i in range(n):
        size = 1024 + (i * 7919) % (24 * 1024)
        src = O.generate(seed0 + i, size)
        if i % 10 == 7:                      # a share of the files carries exactly one injected fault
            bad = O.inject(src, seed0 + i, (i // 10) % 10)      # all ten fault kinds in turn
            src = bad or src
        files.append(src)
    return files


def test_corpus_matches_oracle_per_file(fe):
    files = make_corpus(60)
    units = fe.compile_corpus(files)
    assert len(units) == len(files)
    errors = 0
    for src, u in zip(files, units):
        want_err = O.dump(src, O.ERROR)
        if want_err != "Ok\n":
            errors += 1
            assert u.status == 1 and u.dump_error() == want_err
            continue
        assert u.status == 0, (u.status, src[:60])
        got, want = u.dump_ir(), O.dump(src, O.IR)
        assert hashlib.sha256(got.encode()).digest() == hashlib.sha256(want.encode()).digest()
    assert errors >= 3
    for u in units:
        u.close()


def test_empty_corpus_and_empty_file(fe):
    assert fe.compile_corpus([]) == []
    u = fe.compile_corpus([b"", b"; nothing\n"])
    assert [x.status for x in u] == [0, 0]
    assert all(x.dump_ir() == "" for x in u)


def test_differential_fuzz_small_sources(fe):
    """tools/fuzz_parity.py in small: generated sources of random size, every fifth with an injected syntax error, some
    truncated at a random byte; whole path through opusfe_parse_batch against the oracle"""
    import random
    rnd = random.Random(4)
    srcs = []
    for k in range(160):
        s = O.generate(7000 + k, rnd.randint(2000, 30000))
        if k % 5 == 3:
            s = O.inject(s, 7000 + k, k % 3)
        if k % 11 == 7:
            s = s[:rnd.randint(1, len(s))]
        srcs.append(s)
    units = fe.compile_corpus(srcs, lower=True)
    for s, u in zip(srcs, units):
        want_err = O.dump(s, O.ERROR)
        if want_err == "Unsupported\n":
            assert u.status == 2
        elif want_err != "Ok\n":
            assert u.status == 1 and u.dump_error() == want_err
        else:
            want_ir, want_sem = O.dump(s, O.IR), O.dump(s, O.SEM)
            if want_ir == "" and want_sem != "":
                assert u.status == 7 and u._dump(4) == want_sem
            else:
                assert u.status == 0 and u.dump_ir() == want_ir and u._dump(4) == want_sem
        u.close()


def test_semantic_mutants(fe):
    """ill-typed programs (tests/mutate.py): the 25 semantic error kinds, Error instructions and the panic paths of the
    reference's IR generation — IR text and ordered diagnostics against the oracle"""
    import mutate as M
    srcs = M.mutants(400, seed0=50_000) + M.byte_mutants(300, seed0=50_000) + M.trap_mutants(300, seed0=50_000)
    units = fe.compile_corpus(srcs, lower=True)
    seen = {}
    for s, u in zip(srcs, units):
        w = M.expectation(s)
        seen[w[0]] = seen.get(w[0], 0) + 1
        if w[0] == "unsupported":
            assert u.status == 2, s
        elif w[0] == "syntax":
            assert u.status == 1 and u.dump_error() == w[1], s
        elif w[0] == "semantic":
            assert u.status == 7 and u._dump(4) == w[1], s
        else:
            assert u.status == 0, (u.status, s)
            assert u.dump_ir() == w[1], s
            assert u._dump(4) == w[2], s
            assert u.ir_view().text() == w[1]
        u.close()
    assert seen.get("ir", 0) > 100 and seen.get("syntax", 0) > 50 and seen.get("semantic", 0) > 0


def test_config5_shape_at_scale(fe):
    """BASELINE config 5 in small: 1000 files, sizes log-uniform in [1 KB, 64 KB], i = 7 (mod 100) carries exactly one injected
    fault (all ten kinds in turn): error tuple for the faulty files, hash of the IR text for the rest"""
    import math
    import random
    rnd = random.Random(55)
    files = []
    for i in range(1000):
        size = int(math.exp(rnd.uniform(math.log(1024), math.log(65536))))
        src = O.generate(5_000_000 + i, size)
        if i % 100 == 7:
            src = O.inject(src, 5_000_000 + i, (i // 100) % 10) or src
        files.append(src)
    units = fe.compile_corpus(files, lower=True)
    faulty = 0
    for src, u in zip(files, units):
        want_err = O.dump(src, O.ERROR)
        if want_err != "Ok\n":
            faulty += 1
            assert u.status == 1 and u.dump_error() == want_err
        else:
            assert u.status == 0, (u.status, src[:60])
            assert hashlib.sha256(u.dump_ir().encode()).digest() == hashlib.sha256(O.dump(src, O.IR).encode()).digest()
        u.close()
    assert faulty >= 8


def test_every_diagnostic_of_the_mutants_renders(fe):
    """the payload of the diagnostic records (name, argument types, type ids) is usable for every error the mutants raise: the
    type ids index OPUSFE_A_TYPES, the names are tokens of the unit, and render.py produces a message for each record"""
    import mutate as M
    from opus_amd import render as R
    kinds = set()
    n_rendered = n_records = 0
    for src in M.mutants(250, seed0=61_000):
        u = fe.get_definitions(src)
        if u.status != 0:
            continue
        rc = u.compile_source()
        if rc not in (0, 7):
            continue
        view = u.ir_view()
        base = n_records
        # the whole stderr text against the oracle's independent restatement of print_error + print_span (oracle dump 6)
        assert R.render_unit_errors(u, "input.opus") == O.dump(src, O.RENDERED), src
        for rec in view.sem_errors:
            n_records += 1
            for t in rec["t"]:
                assert int(t) == 0xFFFFFFFF or int(t) < len(view.types)
            name = int(rec["name"])
            assert name == 0xFFFFFFFF or (name & 0x80000000) or name < len(view.tokens)
            msg = R.semantic_message_of_record(rec, view, view.sem_error_arg_types[n_records - 1 - base])
            if msg is None:
                raise AssertionError(("a record that does not render", rec))
            else:
                n_rendered += 1
                kinds.add(int(rec["kind"]))
                assert "\n" not in msg and "?" not in msg.split(":")[0]
        u.close()
    assert n_records > 50 and n_rendered >= 0.95 * n_records and len(kinds) >= 8


def test_rendered_errors_match_the_oracle(fe):
    """the text the reference writes to stderr (compile.rs:782-917 print_error, frontend.rs:201-256 print_span),
    rendered by opus_amd/render.py from the arrays of the HIP path, against the oracle's own restatement of the same lines
    (oracle_dump.cpp dump_rendered_errors) — syntax errors, finalize-time errors and the diagnostics of IR generation"""
    import mutate as M
    from opus_amd import render as R
    many_args = (b"g(a int32, b int32, c int32, d int32, e int32, f int32, h int32, i int32, j int32, k int32, l int32, m int32, n int32) null\n\treturn null\n"
                 b"f() null\n\tg(1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, true)\n\tg(1, 2, 3, 4, 5, true)\n\tnosuch(1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14)\n")
    srcs = [many_args] + M.mutants(300, seed0=71_000) + M.byte_mutants(150, seed0=71_000)
    for i in range(60):
        s = O.generate(72_000 + i, 3000)
        srcs.append(O.inject(s, i, i % 10) or s)
    n_sem = n_syn = 0
    for src in srcs:
        want = O.dump(src, O.RENDERED)
        if want == "Unsupported\n":
            continue
        u = fe.get_definitions(src)
        if u.status == 0:
            rc = u.compile_source()
            if rc == 2:
                u.close()
                continue
        elif u.status != 1:
            u.close()
            continue
        got = R.render_unit_errors(u, "input.opus")
        assert got == want, (src, got, want)
        n_syn += u.status == 1
        n_sem += u.status in (0, 7) and want != ""
        u.close()
    assert n_sem > 50 and n_syn > 50

