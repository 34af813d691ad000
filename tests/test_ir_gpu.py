"""GPU IR-generation parity: the `opus -Z` text (impl Display for IrGenerator) and the semantic diagnostics of IR generation
vs the CPU oracle, bit-exact."""
import json
import os

import pytest

import oracle_lib as O
from cases import FINALIZE_ERROR_CASES, FNPTR_CASE, IR_CASES, IS_CASE, PARSE_CASES, PROC_CASE, big_function as _big_function

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
with open(os.path.join(HERE, "golden", "kats.json")) as f:
    KATS = json.load(f)
with open(os.path.join(HERE, "golden", "kats2.json")) as f:
    KATS2 = json.load(f)["kats"]


def first_diff(got, want):
    g, w = got.split("\n"), want.split("\n")
    for i, (a, b) in enumerate(zip(g, w)):
        if a != b:
            return f"line {i}: got {a[:200]!r} want {b[:200]!r}; context {w[max(0, i - 8):i + 2]}"
    return f"length differs: got {len(g)} want {len(w)}"


def check_ir(fe, src: bytes):
    if O.dump(src, O.ERROR) != "Ok\n":
        return None
    want_ir, want_sem = O.dump(src, O.IR), O.dump(src, O.SEM)
    u = fe.get_definitions(src)
    assert u.status == 0
    rc = u.compile_source()
    if want_ir == "" and want_sem != "":
        assert rc == 7, "finalize errors: the reference generates no IR (frontend.rs:145)"
        assert u._dump(4) == want_sem, "finalize-time diagnostics in report order"
        return u
    assert rc == 0, (rc, fe.message())
    got = u.dump_ir()
    assert got == want_ir, first_diff(got, want_ir)
    assert u._dump(4) == want_sem
    # backend hand-off: the flat arrays alone reproduce the same text
    view_text = u.ir_view().text()
    assert view_text == want_ir, first_diff(view_text, want_ir)
    return u


def test_kat_ir_matches_hand_derived_fixtures(fe):
    for k in ("kat1", "kat2", "kat3"):
        u = fe.get_definitions(KATS[k]["src"].encode())
        assert u.compile_source() == 0
        assert u.dump_ir() == KATS[k]["ir"]
    e2 = KATS["kat_e"][1]
    u = fe.get_definitions(e2["src"].encode())
    assert u.compile_source() == 0 and u.dump_ir() == e2["ir"]


@pytest.mark.parametrize("case", KATS2, ids=[c["name"] for c in KATS2])
def test_kats2_gpu_against_the_hand_derived_text(fe, case):
    """the HIP path against the expectations derived by hand from the Rust (no oracle in between): IR text, the same text
    rebuilt from the flat arrays, the semantic diagnostics (kind, position, length) and, where given, the AST shape"""
    src = case["src"].encode()
    u = fe.get_definitions(src)
    assert u.status == 0
    if case["ast_contains"]:
        assert case["ast_contains"] in u.dump_ast()
    assert u.compile_source() == 0
    assert u.dump_ir() == case["ir"], first_diff(u.dump_ir(), case["ir"])
    assert u.ir_view().text() == case["ir"]
    assert u._dump(4).splitlines() == case["errors"]
    par, seq = u.lowering_stats()
    assert par + seq == 1
    # the message line of every diagnostic (compile.rs:816-915), rendered from the record's payload and the type arrays
    from opus_amd import render as R
    view = u.ir_view()
    assert len(view.sem_errors) == len(case["messages"])
    for rec, want in zip(view.sem_errors, case["messages"]):
        if want is not None:
            assert R.semantic_message_of_record(rec, view) == want


def test_pointer_identity_takes_the_node_parallel_path(fe):
    u = check_ir(fe, IS_CASE.encode())
    assert u.lowering_stats() == (1, 0)


def test_proc_expressions_and_calls_through_function_variables(fe):
    """declined by the node-parallel path (lowering them there was tried: the two extra branches of the inlined visit cost 19 % of the
    kernel's time in register pressure for every function); the sequential machine gives the reference's IR"""
    for src in (PROC_CASE, FNPTR_CASE):
        u = check_ir(fe, src.encode())
        assert u.lowering_stats() == (0, 1), src


def test_finalize_time_invalid_type_is_rendered_with_its_type(fe):
    """compile.rs:1031-1034, 844-847: a global whose initialiser does not fit its type; no IR (frontend.rs:145), but the record
    names the type and OPUSFE_A_TYPES stays readable (hand-derived message, unpinned like all rendering)"""
    from opus_amd import render as R
    src = b"type T = record { x int32 }\nvar a bool = 5\nvar b @T = -1\nf() null\n\tnosuch()\n"
    u = fe.get_definitions(src)
    assert u.compile_source() == 7
    view = u.ir_view()
    got = [R.semantic_message_of_record(rec, view) for rec in view.sem_errors]
    assert got == ["invalid type: bool", "invalid type: @T"]
    text = R.render_semantic_record(view.sem_errors[1], view, "g.opus")
    assert text.startswith("\x1b[31;1mError:\x1b[0m invalid type: @T\n") and "var b @T = " in text


@pytest.mark.parametrize("src", IR_CASES, ids=[f"ir{i}" for i in range(len(IR_CASES))])
def test_ir_hand_cases(fe, src):
    check_ir(fe, src.encode())


def test_finalize_time_diagnostics_are_ordered(fe):
    """with finalize errors the reference prints them in definition order and generates no IR"""
    from opus_amd import ir_view as V
    for s in FINALIZE_ERROR_CASES:
        src = s.encode()
        want = O.dump(src, O.SEM)
        assert want != "" and O.dump(src, O.IR) == "", s
        u = check_ir(fe, src)
        errs = u.array(V.A_SEM_ERRORS)           # the same list as an array (the boundary's "ordered array")
        assert [f"{int(e['low'])} {int(e['len'])}" for e in errs] == [ln.split(" ", 1)[1] for ln in want.splitlines()]
        n = __import__("ctypes").c_uint64()
        import opus_amd.frontend as F
        F.lib().opusfe_semantic_error_count(u.handle, n)
        assert n.value == len(errs)


def test_ir_of_parser_cases(fe):
    for s in PARSE_CASES:
        check_ir(fe, s.encode())


@pytest.mark.parametrize("seed,size", [(1, 4096), (2, 70000), (3, 1 << 20), (11, 3 << 20)])
def test_generated_ir(fe, seed, size):
    u = check_ir(fe, O.generate(seed, size))
    c = u.counts()
    assert c["functions"] > 0 and c["instrs"] > 0 and c["vars"] > 0


def test_deep_regions_ir(fe):
    check_ir(fe, O.generate(3, 2 << 20, flags=1))


def test_c1_ten_thousand_lines_all_dumps(fe):
    """BASELINE config 1: the 10k-line source, every dump compared."""
    src = O.generate_lines(1, 10000)
    u = check_ir(fe, src)
    assert u.dump_tokens() == O.dump(src, O.TOKENS)
    assert u.dump_ast() == O.dump(src, O.AST)


def test_lowering_twice_is_idempotent(fe):
    src = O.generate(21, 50000)
    u = fe.get_definitions(src)
    assert u.compile_source() == 0
    a = u.dump_ir()
    assert u.compile_source() == 0
    assert u.dump_ir() == a == O.dump(src, O.IR)


def test_ir_big_function_spills_every_lds_table(fe):
    src = _big_function().encode()
    u = check_ir(fe, src)
    assert u is not None, "the stress source must be well formed"
    c = u.counts()
    assert c["vars"] > 1500 and c["bbs"] > 300, c


def test_ir_many_big_functions(fe):
    """several oversized functions side by side (one wave each) plus small ones in between"""
    parts = []
    for j in range(6):
        parts.append(_big_function(80 + 20 * j, 100 + 10 * j, 60, 40).replace("big(", f"big{j}("))
        parts.append(f"small{j}(x int32) int32\n\treturn x + {j}\n")
    check_ir(fe, "".join(parts).encode())
