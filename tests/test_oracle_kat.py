"""CPU: the oracle (C++ restatement of the reference) against the hand-derived known-answer fixtures.

Parity is UNPINNED by the reference itself (it ships no tests / golden vectors and cannot be built here); the
fixtures in tests/golden/kats.json were derived by hand from the cited Rust.
"""
import json
import os

import pytest

import oracle_lib as O

HERE = os.path.dirname(os.path.abspath(__file__))
with open(os.path.join(HERE, "golden", "kats.json")) as f:
    KATS = json.load(f)


def test_kat1_tokens_ast_ir():
    k = KATS["kat1"]
    src = k["src"].encode()
    assert O.dump(src, O.TOKENS) == k["tokens"]
    assert O.dump(src, O.AST).split("\n")[0].startswith(k["ast_head"])
    assert O.dump(src, O.IR) == k["ir"]


def test_kat2_token_kinds_and_ir():
    k = KATS["kat2"]
    src = k["src"].encode()
    kinds = [l.split(" ", 2)[2].split("(")[0] for l in O.dump(src, O.TOKENS).splitlines()]
    assert kinds == k["token_kinds"]
    assert O.dump(src, O.IR) == k["ir"]


def test_kat3_while_break_and_phi():
    k = KATS["kat3"]
    assert O.dump(k["src"].encode(), O.IR) == k["ir"]


@pytest.mark.parametrize("case", KATS["kat_l"], ids=[repr(c["src"]) for c in KATS["kat_l"]])
def test_kat_l_lexer_quirks(case):
    got = O.dump(case["src"].encode(), O.LEX).splitlines()
    # the raw stream ends with EndOfFile unless a lexical error is the only thing in it
    body = [l for l in got if not l.endswith(" EndOfFile")]
    assert body == case["tokens"]


@pytest.mark.parametrize("case", KATS["kat_e"], ids=[c["name"] for c in KATS["kat_e"]])
def test_kat_e_first_error_and_swallows(case):
    src = case["src"].encode()
    assert O.dump(src, O.ERROR) == case["error"]
    if case["ir"] is not None:
        assert O.dump(src, O.IR) == case["ir"]


def test_kat_e3_if_span_ends_after_swallowed_char():
    src = KATS["kat_e"][2]["src"].encode()
    ast = O.dump(src, O.AST)
    hash_pos = src.index(b"#")
    assert f"  S If 11 {hash_pos + 1}\n" in ast


def test_generated_programs_are_well_typed():
    for seed in (1, 2, 3):
        src = O.generate(seed, 200_000)
        assert O.dump(src, O.ERROR) == "Ok\n"
        assert O.dump(src, O.SEM) == ""


def test_generator_is_deterministic_and_injector_produces_errors():
    a, b = O.generate(5, 50_000), O.generate(5, 50_000)
    assert a == b
    hits = 0
    for kind in range(10):
        bad = O.inject(a, 1234 + kind, kind)
        if bad and O.dump(bad, O.ERROR) != "Ok\n":
            hits += 1
    assert hits >= 6


# ---- second set: 53 hand-derived IR-generation cases (tests/golden/make_kats2.py; first-contact record in oracle/REVIEW.md)
with open(os.path.join(HERE, "golden", "kats2.json")) as f:
    KATS2 = json.load(f)["kats"]


@pytest.mark.parametrize("case", KATS2, ids=[c["name"] for c in KATS2])
def test_kats2_ir_and_diagnostics(case):
    src = case["src"].encode()
    assert O.dump(src, O.ERROR) == "Ok\n"
    assert O.dump(src, O.IR) == case["ir"]
    assert O.dump(src, O.SEM).splitlines() == case["errors"]
    if case["ast_contains"]:
        assert case["ast_contains"] in O.dump(src, O.AST)


# ---- round 3: hand-derived TOKEN + AST + SPAN cases (tests/golden/make_kats3.py) ----
with open(os.path.join(HERE, "golden", "kats3.json")) as f:
    KATS3 = json.load(f)["cases"]


@pytest.mark.parametrize("case", KATS3, ids=[c["name"] for c in KATS3])
def test_kats3_tokens_ast_spans_and_error_tuples(case):
    src = case["src"].encode()
    if "tokens" in case:
        assert O.dump(src, O.TOKENS) == case["tokens"]
    if "ast" in case:
        assert O.dump(src, O.AST) == case["ast"]
    assert O.dump(src, O.ERROR) == case.get("error", "Ok\n")

