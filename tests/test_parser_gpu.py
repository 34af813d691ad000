"""GPU parser parity: AST dump (Rust {:?} text + span table) and first-syntax-error tuple vs the CPU oracle (bit-exact)."""
import json
import os

import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
with open(os.path.join(HERE, "golden", "kats.json")) as f:
    KATS = json.load(f)


def first_diff(got, want):
    g, w = got.split("\n"), want.split("\n")
    for i, (a, b) in enumerate(zip(g, w)):
        if a != b:
            return f"line {i}: got {a[:300]!r} want {b[:300]!r}"
    return f"length differs: got {len(g)} want {len(w)}"


def check(fe, src: bytes):
    want_err = O.dump(src, O.ERROR)
    u = fe.get_definitions(src)
    got_err = u.dump_error()
    assert got_err == want_err, (got_err, want_err)
    if want_err == "Ok\n":
        assert u.status == 0
        got, want = u.dump_ast(), O.dump(src, O.AST)
        assert got == want, first_diff(got, want)
        assert u.dump_tokens() == O.dump(src, O.TOKENS)
    return u


from cases import PARSE_CASES as HAND


@pytest.mark.parametrize("src", HAND, ids=[f"hand{i}" for i in range(len(HAND))])
def test_hand_written_cases(fe, src):
    check(fe, src.encode())      # includes a swallowed bad escape (Q6): followed exactly by the resync passes


def test_kats(fe):
    for k in ("kat1", "kat2", "kat3"):
        check(fe, KATS[k]["src"].encode())
    for c in KATS["kat_e"]:
        u = check(fe, c["src"].encode())
        assert u.dump_error() == c["error"]


@pytest.mark.parametrize("seed,size", [(1, 4096), (2, 70000), (3, 1 << 20), (11, 3 << 20)])
def test_generated_ast(fe, seed, size):
    check(fe, O.generate(seed, size))


def test_deep_regions_ast(fe):
    check(fe, O.generate(3, 2 << 20, flags=1))


def test_c1_ten_thousand_lines(fe):
    check(fe, O.generate_lines(1, 10000))


@pytest.mark.parametrize("kind", list(range(10)))
def test_injected_faults_error_tuple(fe, kind):
    base = O.generate(40 + kind, 30000)
    for s in range(12):
        bad = O.inject(base, 1000 * kind + s, kind)
        if not bad:
            continue
        want = O.dump(bad, O.ERROR)
        u = fe.get_definitions(bad)
        assert u.dump_error() == want
        if want == "Ok\n":
            assert u.dump_ast() == O.dump(bad, O.AST)


def test_truncations_of_a_program(fe):
    """Every prefix of a small program: exercises end-of-file inside every construct."""
    src = O.generate(77, 1500)
    for n in range(0, len(src), 7):
        data = src[:n]
        want = O.dump(data, O.ERROR)
        u = fe.get_definitions(data)
        if want == "Unsupported\n":
            assert u.status == 2
            continue
        assert u.dump_error() == want, (n, data[-40:])
        if want == "Ok\n":
            assert u.dump_ast() == O.dump(data, O.AST), n


def test_resync_passes_only_on_swallowed_lexical_errors(fe):
    ok = b"f() int32\n\treturn 1\n"
    assert fe.get_definitions(ok).resync_passes() == 0
    reported = b'f() null\n\tvar s = "ab\\q"\n'                       # the Err token is the first syntax error: one pass
    u = fe.get_definitions(reported)
    assert u.status == 1 and u.resync_passes() == 0 and u.dump_error() == O.dump(reported, O.ERROR)
    for src in (b'f() int32\n\treturn 1\n"ab\\q\ng() int32\n\treturn 2\n',                      # Q7: dropped at top level
                b'f() int32\n\treturn 1\n"a\\xZ1 g() int32\n\treturn 2\n',
                b'f() null\n\tx = 1.5f";"\n\ty = 2\n"zz\\q\n"yy\\w\nh() null\n\treturn\n',          # several in one file
                b'f(a bool) null\n\tif a\n\t\treturn\n\t"bad \\q esc" g()\n'):                        # Q6: the else probe
        u = fe.get_definitions(src)
        assert u.dump_error() == O.dump(src, O.ERROR), src
        assert u.dump_tokens() == O.dump(src, O.TOKENS), src
        if u.status == 0:
            assert u.dump_ast() == O.dump(src, O.AST)
            assert u.compile_source() in (0, 7)
            assert u.dump_ir() == O.dump(src, O.IR)
    assert fe.get_definitions(b'f() int32\n\treturn 1\n"ab\\q\ng() int32\n\treturn 2\n').resync_passes() == 1


def test_resync_pass_limit_is_unsupported_not_wrong(fe):
    """16 swallowed lexical errors per unit are followed exactly; one more and the answer is OPUSFE_UNSUPPORTED"""
    def src(k):
        return b"".join(b'"a\\q\nf%d() int32\n\treturn %d\n' % (i, i) for i in range(k))
    u = fe.get_definitions(src(16))
    assert u.status == 0 and u.resync_passes() == 16 and u.dump_ast() == O.dump(src(16), O.AST)
    assert u.compile_source() == 0 and u.dump_ir() == O.dump(src(16), O.IR)
    assert fe.get_definitions(src(17)).status == 2


with open(os.path.join(HERE, "golden", "kats3.json")) as f:
    KATS3 = json.load(f)["cases"]


@pytest.mark.parametrize("case", KATS3, ids=[c["name"] for c in KATS3])
def test_kats3_gpu_against_the_hand_derived_text(fe, case):
    """the HIP path against the hand-derived token / AST / span text directly (no oracle in between), on both parsers"""
    src = case["src"].encode()
    for mode in (0, 1):
        fe.set_parser_mode(mode)
        try:
            u = fe.get_definitions(src)
            if "tokens" in case:
                assert u.dump_tokens() == case["tokens"]
            if "ast" in case:
                assert u.dump_ast() == case["ast"]
            assert u.dump_error() == case.get("error", "Ok\n")
            u.close()
        finally:
            fe.set_parser_mode(0)

