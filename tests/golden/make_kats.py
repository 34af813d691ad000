"""Writes the hand-derived known-answer fixtures.

HAND-DERIVED, NOT MACHINE-VERIFIED: the reference (Rust, LLVM 7) cannot be built in this image and ships no
tests or golden vectors, so these expectations were derived by executing the cited Rust by hand.  The files
written here are pure data (source text in, expected dump text out); this script only exists so that the
TAB characters and byte-exact line endings are reproducible.  Run:  python tests/golden/make_kats.py
"""
import json
import os

HERE = os.path.dirname(os.path.abspath(__file__))

KAT1_SRC = "main() int32\n\treturn 1 + 2 * 3\n"
KAT1_TOKENS = """0 4 Identifier("main")
4 5 LeftParenthesis
5 6 RightParenthesis
7 12 Identifier("int32")
14 14 Indent
14 20 Return
21 22 Integer(1, None)
23 24 Plus
25 26 Integer(2, None)
27 28 Asterisk
29 30 Integer(3, None)
31 31 Dedent
31 31 EndOfFile
"""
KAT1_AST_HEAD = 'Function(main[), [Return(Operator(Plus, Integer(1, None), Operator(Times, Integer(2, None), Integer(3, None))))])'
KAT1_IR = """main[]: int32
VAR %0: int32
VAR %1: int32
VAR %2: int32
VAR %3: int32
VAR %4: int32
@0
  %0 = 1
  %1 = 2
  %2 = 3
  %3 = multiply %1, %2
  %4 = add %0, %3
  return %4
"""

KAT2_SRC = "f(a int32) int32\n\tif a < 1\n\t\treturn 0\n\treturn a\n"
KAT2_TOKEN_KINDS = ["Identifier", "LeftParenthesis", "Identifier", "Identifier", "RightParenthesis", "Identifier", "Indent", "If",
                    "Identifier", "LessThan", "Integer", "Indent", "Return", "Integer", "Dedent", "Return", "Identifier", "Dedent",
                    "EndOfFile"]
KAT2_IR = """f[:int32]: int32
VAR %0: int32
VAR %1: @mut int32
VAR %2: @int32
VAR %3: int32
VAR %4: int32
VAR %5: bool
VAR %6: int32
VAR %7: @int32
VAR %8: int32
@0
  %1 = alloc
  in %1 store %0
  %2 = cast PointerType [@mut int32 to @int32] %1
  %3 = load %2
  %4 = 1
  %5 = lessthan %3, %4
  branch %5, @1, @2
@1
  %6 = 0
  return %6
  jump @3
@2
  jump @3
@3
  %7 = cast PointerType [@mut int32 to @int32] %1
  %8 = load %7
  return %8
"""

KAT3_SRC = "g(n int32) int32\n\tvar i = 0\n\twhile i < n and true\n\t\ti = i + 1\n\t\tbreak\n\treturn i\n"
KAT3_IR = """g[:int32]: int32
VAR %0: int32
VAR %1: @mut int32
VAR %2: int32
VAR %3: @mut int32
VAR %4: @int32
VAR %5: int32
VAR %6: @int32
VAR %7: int32
VAR %8: bool
VAR %9: bool
VAR %10: bool
VAR %11: @int32
VAR %12: int32
VAR %13: int32
VAR %14: int32
VAR %15: @int32
VAR %16: int32
@0
  %1 = alloc
  in %1 store %0
  %2 = 0
  %3 = alloc
  in %3 store %2
  jump @1
@1
  %4 = cast PointerType [@mut int32 to @int32] %3
  %5 = load %4
  %6 = cast PointerType [@mut int32 to @int32] %1
  %7 = load %6
  %8 = lessthan %5, %7
  branch %8, @2, @3
@2
  %9 = true
  jump @3
@3
  %10 = phi @1 => %8, @2 => %9
  branch %10, @4, @6
@4
  %11 = cast PointerType [@mut int32 to @int32] %3
  %12 = load %11
  %13 = 1
  %14 = add %12, %13
  in %3 store %14
  jump @7
@5
  jump @1
@6
  jump @7
@7
  %15 = cast PointerType [@mut int32 to @int32] %3
  %16 = load %15
  return %16
"""

# KAT-L: source -> token lines before the EndOfFile token (lexer alone, ignore_dents == 0)
KAT_L = [
    ("5s", ['0 1 Integer(5, Some((false, 0)))', '1 2 Identifier("s")']),
    ("5i8", ['0 3 Integer(5, Some((true, 8)))']),
    ("5n", ['0 2 Integer(5, Some((false, 32)))']),
    ("5i80", ['0 3 Integer(5, Some((true, 8)))', '3 4 Integer(0, None)']),
    ("5in", ['0 2 Integer(5, Some((true, 32)))', '2 3 Identifier("n")']),
    ("7i12", ['0 4 Err(InvalidNumericSize 2 2 12)']),
    ("1.5f32", ['0 6 Float(1.5, Some(32))']),
    ("2.", ['0 2 Float(2.0, None)']),
    ("1..2", ['0 2 Float(1.0, None)', '2 3 Dot', '3 4 Integer(2, None)']),
    ("a\\b", ['0 3 Identifier("a\\\\b")']),
    ('"a\\x41\\n"', ['0 9 String([97, 65, 10])']),
    ("x<=y", ['0 1 Identifier("x")', '1 3 LessThanEquals', '3 4 Identifier("y")']),
    ("; c\n\tfoo", ['5 5 Indent', '5 8 Identifier("foo")', '8 8 Dedent']),
    ("123abc", ['0 3 Integer(123, None)', '3 6 Identifier("abc")']),
]

KAT_E = [
    # (name, source, expected error line, expected IR or None)
    ("E1", "f() int32\n\treturn #\n", "UnexpectedCharacter 18 1 35 line 2\n", None),
    ("E2", "#\nf() int32\n\treturn 1\n", "Ok\n", "f[]: int32\nVAR %0: int32\n@0\n  %0 = 1\n  return %0\n"),
    ("E3", "f() int32\n\tif true\n\t\treturn 1\n\t#\n\treturn 2\n", "Ok\n", None),
]


def main():
    kats = {
        "_note": "HAND-DERIVED from the reference's Rust (SURVEY.md §8c, Appendix D); not machine-verified.",
        "kat1": {"src": KAT1_SRC, "tokens": KAT1_TOKENS, "ast_head": KAT1_AST_HEAD, "ir": KAT1_IR},
        "kat2": {"src": KAT2_SRC, "token_kinds": KAT2_TOKEN_KINDS, "ir": KAT2_IR},
        "kat3": {"src": KAT3_SRC, "ir": KAT3_IR},
        "kat_l": [{"src": s, "tokens": t} for s, t in KAT_L],
        "kat_e": [{"name": n, "src": s, "error": e, "ir": ir} for n, s, e, ir in KAT_E],
    }
    with open(os.path.join(HERE, "kats.json"), "w") as f:
        json.dump(kats, f, indent=1)
        f.write("\n")


if __name__ == "__main__":
    main()
