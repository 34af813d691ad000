"""CPU: the REAL device code (opus_amd/csrc kernels' bodies) compiled for the host with g++ under
AddressSanitizer/UBSan (tests/emu) and compared bit-exactly with the oracle.  This is how indexing bugs are
found before a kernel ever runs on a GPU (a faulting kernel can reset a whole node); it is test
infrastructure, not a CPU path of the product."""
import pytest

import emu_lib as E
import oracle_lib as O
from cases import FINALIZE_ERROR_CASES, IR_CASES, LEX_CASES, PARSE_CASES, big_function


def compare_parse(cases):
    errs = E.dump_many(cases, 3)
    asts = E.dump_many(cases, 1)
    toks = E.dump_many(cases, 0)
    for c, (st, e), (_, a), (_, t) in zip(cases, errs, asts, toks):
        want = O.dump(c, O.ERROR)
        assert e == want, (c[:80], e, want)
        if want == "Ok\n":
            assert a == O.dump(c, O.AST), c[:80]
            assert t == O.dump(c, O.TOKENS), c[:80]


def test_emu_lexer_alone():
    cases = [s.encode() for s in LEX_CASES]
    got = E.dump_many(cases, 5)
    for c, (st, g) in zip(cases, got):
        want = O.dump(c, O.LEX)
        if want == "Unsupported\n":
            assert st == 2
            continue
        if "Err(InvalidEscapeSequence" in want:
            w, gl = want.split("\n"), g.split("\n")
            k = next(i for i, l in enumerate(w) if "Err(InvalidEscapeSequence" in l)
            assert gl[:k] == w[:k]
            continue
        assert g == want, c


def test_emu_parser_hand_cases():
    compare_parse([s.encode() for s in PARSE_CASES])


def test_emu_parser_generated_and_deep():
    compare_parse([O.generate(1, 4096), O.generate(2, 70000), O.generate(3, 2 << 20, flags=1)])


def test_emu_parser_injected_faults():
    cases = []
    for kind in range(10):
        base = O.generate(40 + kind, 20000)
        for s in range(6):
            bad = O.inject(base, 1000 * kind + s, kind)
            if bad:
                cases.append(bad)
    compare_parse(cases)


def test_emu_parser_truncations():
    src = O.generate(77, 1500)
    compare_parse([src[:n] for n in range(0, len(src), 5)])


def compare_ir(cases):
    cases = [c for c in cases if O.dump(c, O.ERROR) == "Ok\n"]
    irs = E.dump_many(cases, 2)
    sems = E.dump_many(cases, 4)
    for c, (st, ir), (_, sem) in zip(cases, irs, sems):
        want_ir, want_sem = O.dump(c, O.IR), O.dump(c, O.SEM)
        if want_ir == "" and want_sem != "":
            assert st == 7, c[:80]
            assert sem == want_sem, c[:80]      # the finalize-time diagnostics, in report order
            continue
        assert st == 0, (st, c[:80])
        assert ir == want_ir, c[:80]
        assert sem == want_sem, c[:80]


def test_emu_ir_hand_cases():
    compare_ir([s.encode() for s in IR_CASES] + [s.encode() for s in PARSE_CASES] + [s.encode() for s in FINALIZE_ERROR_CASES])


def test_emu_ir_generated_and_deep():
    compare_ir([O.generate(1, 4096), O.generate(2, 70000), O.generate(3, 2 << 20, flags=1)])


def test_emu_flat_lowering_launch_plans():
    """the flat lowering (lower_par.h) under the launch plans run_lower chooses between — one kernel per kind class and round, one kernel
    for a whole round, one workgroup for the tail of the schedule — and with every function REQUIRED on the flat path: same IR text"""
    cases = [O.generate(5, 30000), O.generate(6, 300000)]
    want = [O.dump(c, O.IR) for c in cases]
    for env in ({"EMU_REQUIRE_PAR": "1"}, {"EMU_REQUIRE_PAR": "1", "EMU_FLAT_TAIL": "100000000"}, {"EMU_REQUIRE_PAR": "1", "EMU_FLAT_TAILKERNEL": "100000000"},
                {"EMU_FLAT_TAIL": "0", "EMU_FLAT_TAILKERNEL": "0"}):
        got = E.dump_many(cases, 2, env=env)
        assert [g for g in got] == [(0, w) for w in want], env


def test_emu_ir_big_function_spill_paths():
    """every LDS-resident table of the two machines overflows into its sparse global region (AddressSanitizer watches)"""
    compare_ir([big_function().encode(), big_function(70, 40, 300, 20).encode()])


def test_emu_float_literals_outside_the_fast_path():
    lits = ["0.30000000000000004", "9007199254740993.0", "9007199254740992.5", "4503599627370497.5", "123456789012345678901234567890.5",
            "0.000000000000000000000000000001", "1.0000000000000000000000000000000000000001", "3.14159265358979323846264338327950288",
            "1.50000000000000000000000000", "000.000", "179769313486231570000000000000000000000.0", "0.1000000000000000055511151231257827"]
    src = "".join(f"const c{i} float64 = {t}\n" for i, t in enumerate(lits)).encode()
    (st, got), = E.dump_many([src], 0)
    assert st == 0 and got == O.dump(src, O.TOKENS)


def test_emu_semantic_mutants():
    """ill-typed programs (tests/mutate.py): the error paths of IR generation, emulated under ASan before any GPU sees them"""
    import mutate as M
    cases = M.mutants(260) + M.byte_mutants(240) + M.trap_mutants(300)
    want = [M.expectation(c) for c in cases]
    errs, irs, sems = E.dump_many(cases, 3), E.dump_many(cases, 2), E.dump_many(cases, 4)
    seen = set()
    for c, w, (_, err), (st, ir), (_, sem) in zip(cases, want, errs, irs, sems):
        seen.add(w[0])
        if w[0] == "unsupported":
            assert st == 2, c
        elif w[0] == "syntax":
            assert st == 1 and err == w[1], c
        elif w[0] == "semantic":
            assert st == 7 and sem == w[1], c
        else:
            assert st == 0, (st, c)
            assert ir == w[1], c
            assert sem == w[2], c
    assert {"syntax", "semantic", "ir"} <= seen
