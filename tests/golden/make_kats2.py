"""Writes tests/golden/kats2.json: the second set of hand-derived known answers for IR generation.

HAND-DERIVED, NOT MACHINE-VERIFIED.  Every expectation below was written by executing the cited Rust by hand
(generate_ir.rs, compile.rs, parse.rs of the reference) BEFORE the case was run through oracle/ or the GPU path; the
cases the oracle (or this derivation) got wrong on first contact are listed in oracle/REVIEW.md, never silently fixed.
Each entry: source text, the expected `--debug` IR text (frontend.rs:168: one eprintln!("{}", ir_generator) per
function, generate_ir.rs:1201-1283), and the semantic errors raised while generating it as (kind, the span's text,
which occurrence of that text in the source) — positions are computed from the text here so that no offset is typed
by hand.  Pure data; the script only exists so that TAB characters are reproducible.
Run:  python tests/golden/make_kats2.py
"""
import json
import os

HERE = os.path.dirname(os.path.abspath(__file__))

K = []


def kat(name, cites, src, ir, errors=(), ast_contains=None):
    errs, msgs = [], []
    for kind, text, nth, *message in errors:
        pos = -1
        for _ in range(nth + 1):
            pos = src.index(text, pos + 1)
        errs.append(f"{kind} {pos} {len(text)}")
        msgs.append(message[0] if message else None)
    # "messages": the message line of compile.rs:816-915 for each error, hand-derived like the rest (None = not derived)
    K.append({"name": name, "cites": cites, "src": src, "ir": ir.lstrip("\n"), "errors": errs, "messages": msgs, "ast_contains": ast_contains})


FOR_EXTERNS = "extern operator for(n int32) int32\nextern operator continue(it @mut int32, x @mut int32) bool\n"

kat("for_typed_break", "generate_ir.rs:363-469 (for desugaring), 470-476 (break placeholder), 1230-1235 (call operands print bare)",
    FOR_EXTERNS + "f() null\n\tfor x int32 in 3\n\t\tbreak\n", """
f[]: null
VAR %0: int32
VAR %1: [int32] int32
VAR %2: int32
VAR %3: @mut int32
VAR %4: @mut int32
VAR %5: [@mut int32 @mut int32] bool
VAR %6: bool
VAR %7: null
@0
  %0 = 3
  %1 = global $for[:int32]: int32
  %2 = call %1, 0
  %3 = alloc
  in %3 store %2
  jump @1
@1
  %4 = alloc
  %5 = global $continue[:@mut int32 :@mut int32]: bool
  %6 = call %5, 3, 4
  branch %6, @2, @4
@2
  jump @5
@3
  jump @1
@4
  jump @5
@5
  %7 = null
  return %7
""")

kat("for_untyped_q13", "generate_ir.rs:405-409 (loop variable of type Generic), 207-224 (autocast retypes a Generic variable in place)",
    "extern operator for(n int32) int32\nextern operator continue(it @mut int32, x @mut nat8) bool\ng() null\n\tfor x in 7\n\t\tx = 1\n", """
g[]: null
VAR %0: int32
VAR %1: [int32] int32
VAR %2: int32
VAR %3: @mut int32
VAR %4: @mut nat8
VAR %5: [@mut int32 @mut nat8] bool
VAR %6: bool
VAR %7: nat8
VAR %8: null
@0
  %0 = 7
  %1 = global $for[:int32]: int32
  %2 = call %1, 0
  %3 = alloc
  in %3 store %2
  jump @1
@1
  %4 = alloc
  %5 = global $continue[:@mut int32 :@mut nat8]: bool
  %6 = call %5, 3, 4
  branch %6, @2, @3
@2
  %7 = 1
  in %4 store %7
  jump @1
@3
  jump @4
@4
  %8 = null
  return %8
""")

kat("for_without_iterator_functions", "generate_ir.rs:387-391,424-430 (two errors per failed lookup, the call is still pushed), compile.rs:1135-1137",
    "h() null\n\tfor x int32 in 1\n\t\tcontinue\n", """
h[]: null
VAR %0: int32
VAR %1: <error>
VAR %2: <error>
VAR %3: @mut <error>
VAR %4: @mut int32
VAR %5: <error>
VAR %6: <error>
VAR %7: bool
VAR %8: null
@0
  %0 = 1
  %1 = error
  %2 = error
  %2 = call %1, 0
  %3 = alloc
  in %3 store %2
  jump @1
@1
  %4 = alloc
  %5 = error
  %6 = error
  %6 = call %5, 3, 4
  %7 = cast Error [<error> to bool] %6
  branch %7, @2, @4
@2
  jump @1
@3
  jump @1
@4
  jump @5
@5
  %8 = null
  return %8
""", errors=[("UndefinedFunction", "1", 0, "undefined function: for[:int32]"), ("UndefinedFunction", "1", 0, "undefined function: continue[:@mut <error> :@mut int32]")])

kat("record_literal", "generate_ir.rs:1064-1078 (no autocast of field values), 1090-1101, 1103-1128; compile.rs:204-207 (alias name printed)",
    "type P = record { x int32 y nat8 }\nmk() P\n\treturn record P { x = 1 y = 2 }\n", """
mk[]: P
VAR %0: @mut P
VAR %1: int32
VAR %2: @mut int32
VAR %3: nat8
VAR %4: @mut nat8
VAR %5: P
@0
  %0 = alloc
  %1 = 1
  %2 = fieldptr %0, x
  in %2 store %1
  %3 = 2
  %4 = fieldptr %0, y
  in %4 store %3
  %5 = load %0
  return %5
""")

kat("record_literal_unknown_field", "generate_ir.rs:1119-1127 (FieldDoesNotExist at the value's span; the store still happens)",
    "type P = record { x int32 }\nmk() P\n\treturn record P { z = 1 }\n", """
mk[]: P
VAR %0: @mut P
VAR %1: int32
VAR %2: <error>
VAR %3: P
@0
  %0 = alloc
  %1 = 1
  %2 = error
  in %2 store %1
  %3 = load %0
  return %3
""", errors=[("FieldDoesNotExist", "1", 0, "type P has no field named 'z'")])

kat("record_literal_of_non_record", "generate_ir.rs:1144-1147", "f() int32\n\treturn record int32 { x = 1 }\n", """
f[]: int32
VAR %0: @mut int32
VAR %1: int32
VAR %2: <error>
VAR %3: int32
@0
  %0 = alloc
  %1 = 1
  %2 = error
  in %2 store %1
  %3 = load %0
  return %3
""", errors=[("FieldAccessOnNonRecord", "1", 0, "attempt to set or access field of non-record: int32")])

kat("string_length", "generate_ir.rs:1129-1143 (builtin field 1 = length: size)", "len(s string) size\n\treturn s.length\n", """
len[:string]: size
VAR %0: string
VAR %1: @mut string
VAR %2: @string
VAR %3: @size
VAR %4: size
@0
  %1 = alloc
  in %1 store %0
  %2 = cast PointerType [@mut string to @string] %1
  %3 = builtinfieldptr %2, 1
  %4 = load %3
  return %4
""")

kat("string_data_index", "generate_ir.rs:1129-1143 (field 0 = data: {} nat8), 1053-1063, 905-927 (IndexPointer then Array -> Reference cast)",
    "first(s string) nat8\n\treturn s.data(0)\n", """
first[:string]: nat8
VAR %0: string
VAR %1: @mut string
VAR %2: @string
VAR %3: @{} nat8
VAR %4: {} nat8
VAR %5: size
VAR %6: {} nat8
VAR %7: @nat8
VAR %8: nat8
@0
  %1 = alloc
  in %1 store %0
  %2 = cast PointerType [@mut string to @string] %1
  %3 = builtinfieldptr %2, 0
  %4 = load %3
  %5 = 0
  %6 = indexptr %4, %5
  %7 = cast PointerType [{} nat8 to @nat8] %6
  %8 = load %7
  return %8
""")

kat("array_index", "generate_ir.rs:928-947 (IndexArray keeps the indexee's pointer kind), 1008-1021, 1131-1158 of parse.rs (`var a T` is an expression statement)",
    "at(i size) int32\n\tvar a {4} int32\n\treturn a(i)\n", """
at[:size]: int32
VAR %0: size
VAR %1: @mut size
VAR %2: @mut {4} int32
VAR %3: @{4} int32
VAR %4: {4} int32
VAR %5: @{4} int32
VAR %6: @size
VAR %7: size
VAR %8: @int32
VAR %9: int32
@0
  %1 = alloc
  in %1 store %0
  %2 = alloc
  %3 = cast PointerType [@mut {4} int32 to @{4} int32] %2
  %4 = load %3
  %5 = cast PointerType [@mut {4} int32 to @{4} int32] %2
  %6 = cast PointerType [@mut size to @size] %1
  %7 = load %6
  %8 = indexarray %5, %7
  %9 = load %8
  return %9
""")

kat("index_store_through_mutable_array_pointer", "generate_ir.rs:265-274, 905-927 (ArrayMutable -> Mutable cast), 158-165 (null epilogue)",
    "set(p {} mut int32) null\n\tp(1) = 5\n", """
set[:mut {} int32]: null
VAR %0: mut {} int32
VAR %1: @mut mut {} int32
VAR %2: mut {} int32
VAR %3: size
VAR %4: mut {} int32
VAR %5: @mut int32
VAR %6: int32
VAR %7: null
@0
  %1 = alloc
  in %1 store %0
  %2 = load %1
  %3 = 1
  %4 = indexptr %2, %3
  %5 = cast PointerType [mut {} int32 to @mut int32] %4
  %6 = 5
  in %5 store %6
  %7 = null
  return %7
""")

kat("store_through_immutable_reference", "generate_ir.rs:1022-1026, 831-835 (ImmutableLvalue at the operand's span), 239 (lvalue of <error>), 543, 225-230",
    "w(p @int32) null\n\tp@ = 1\n", """
w[:@int32]: null
VAR %0: @int32
VAR %1: @mut @int32
VAR %2: @@int32
VAR %3: @int32
VAR %4: <error>
VAR %5: int32
VAR %6: <error>
VAR %7: null
@0
  %1 = alloc
  in %1 store %0
  %2 = cast PointerType [@mut @int32 to @@int32] %1
  %3 = load %2
  %4 = error
  %5 = 1
  %6 = cast Error [int32 to <error>] %5
  in %4 store %6
  %7 = null
  return %7
""", errors=[("ImmutableLvalue", "p", 1, "lvalue is immutable")])

kat("casts_integer_and_pointer", "generate_ir.rs:720-732; compile.rs:682-696 (Integer, PointerToInteger, IntegerToPointer)",
    "c(x int32, p @int32) nat64\n\tvar a = x:nat8\n\tvar b = p:size\n\tvar q = b:@nat8\n\treturn x:nat64\n", """
c[:int32 :@int32]: nat64
VAR %0: int32
VAR %1: @int32
VAR %2: @mut int32
VAR %3: @mut @int32
VAR %4: @int32
VAR %5: int32
VAR %6: nat8
VAR %7: @mut nat8
VAR %8: @@int32
VAR %9: @int32
VAR %10: size
VAR %11: @mut size
VAR %12: @size
VAR %13: size
VAR %14: @nat8
VAR %15: @mut @nat8
VAR %16: @int32
VAR %17: int32
VAR %18: nat64
@0
  %2 = alloc
  in %2 store %0
  %3 = alloc
  in %3 store %1
  %4 = cast PointerType [@mut int32 to @int32] %2
  %5 = load %4
  %6 = cast Integer [int32 to nat8] %5
  %7 = alloc
  in %7 store %6
  %8 = cast PointerType [@mut @int32 to @@int32] %3
  %9 = load %8
  %10 = cast PointerToInteger [@int32 to size] %9
  %11 = alloc
  in %11 store %10
  %12 = cast PointerType [@mut size to @size] %11
  %13 = load %12
  %14 = cast IntegerToPointer [size to @nat8] %13
  %15 = alloc
  in %15 store %14
  %16 = cast PointerType [@mut int32 to @int32] %2
  %17 = load %16
  %18 = cast Integer [int32 to nat64] %17
  return %18
""")

kat("cast_pointer_to_pointer", "compile.rs:690 (CastType::Pointer: same pointer kind, different pointee)", "pp(p @int32) @nat8\n\treturn p:@nat8\n", """
pp[:@int32]: @nat8
VAR %0: @int32
VAR %1: @mut @int32
VAR %2: @@int32
VAR %3: @int32
VAR %4: @nat8
@0
  %1 = alloc
  in %1 store %0
  %2 = cast PointerType [@mut @int32 to @@int32] %1
  %3 = load %2
  %4 = cast Pointer [@int32 to @nat8] %3
  return %4
""")

kat("autocast_reference_to_byte_array", "compile.rs:644-649 (@T -> {} nat8 is an automatic PointerType cast)", "bytes(p @int32) {} nat8\n\treturn p\n", """
bytes[:@int32]: {} nat8
VAR %0: @int32
VAR %1: @mut @int32
VAR %2: @@int32
VAR %3: @int32
VAR %4: {} nat8
@0
  %1 = alloc
  in %1 store %0
  %2 = cast PointerType [@mut @int32 to @@int32] %1
  %3 = load %2
  %4 = cast PointerType [@int32 to {} nat8] %3
  return %4
""")

kat("invalid_cast", "generate_ir.rs:728-731; compile.rs:682-696 (bool is not integral)", "ic(b bool) int32\n\treturn b:int32\n", """
ic[:bool]: int32
VAR %0: bool
VAR %1: @mut bool
VAR %2: @bool
VAR %3: bool
VAR %4: <error>
VAR %5: int32
@0
  %1 = alloc
  in %1 store %0
  %2 = cast PointerType [@mut bool to @bool] %1
  %3 = load %2
  %4 = error
  %5 = cast Error [<error> to int32] %4
  return %5
""", errors=[("InvalidCast", "b:int32", 0, "invalid cast: bool to int32")])

kat("negate_typed_literal_is_not_folded", "generate_ir.rs:749-765 (the literal takes the expected type int8 first, so Negate is emitted)",
    "n() int8\n\treturn -5\n", """
n[]: int8
VAR %0: int8
VAR %1: int8
@0
  %0 = 5
  %1 = negate %0
  return %1
""")

kat("negate_generic_literal_is_folded", "generate_ir.rs:766-780 (fold into the Integer instruction), compile.rs:1104-1131 (GenericInteger defaults to int32)",
    "m() int32\n\treturn 1 + -2\n", """
m[]: int32
VAR %0: int32
VAR %1: int32
VAR %2: int32
@0
  %0 = 1
  %1 = -2
  %2 = add %0, %1
  return %2
""")

kat("negate_unsigned", "generate_ir.rs:784-787 (InvalidOperandType at the whole negation)", "u(x nat32) nat32\n\treturn -x\n", """
u[:nat32]: nat32
VAR %0: nat32
VAR %1: @mut nat32
VAR %2: @nat32
VAR %3: nat32
VAR %4: <error>
VAR %5: nat32
@0
  %1 = alloc
  in %1 store %0
  %2 = cast PointerType [@mut nat32 to @nat32] %1
  %3 = load %2
  %4 = error
  %5 = cast Error [<error> to nat32] %4
  return %5
""", errors=[("InvalidOperandType", "-x", 0, "invalid operand type: nat32")])

kat("is_on_references", "generate_ir.rs:616-628", "same(a @int32, b @int32) bool\n\treturn a is b\n", """
same[:@int32 :@int32]: bool
VAR %0: @int32
VAR %1: @int32
VAR %2: @mut @int32
VAR %3: @mut @int32
VAR %4: @@int32
VAR %5: @int32
VAR %6: @@int32
VAR %7: @int32
VAR %8: bool
@0
  %2 = alloc
  in %2 store %0
  %3 = alloc
  in %3 store %1
  %4 = cast PointerType [@mut @int32 to @@int32] %2
  %5 = load %4
  %6 = cast PointerType [@mut @int32 to @@int32] %3
  %7 = load %6
  %8 = equals %5, %7
  return %8
""")

kat("is_on_integers", "generate_ir.rs:629-632 (InvalidOperandTypes at the whole expression)", "bad(a int32) bool\n\treturn a is 1\n", """
bad[:int32]: bool
VAR %0: int32
VAR %1: @mut int32
VAR %2: @int32
VAR %3: int32
VAR %4: int32
VAR %5: <error>
VAR %6: bool
@0
  %1 = alloc
  in %1 store %0
  %2 = cast PointerType [@mut int32 to @int32] %1
  %3 = load %2
  %4 = 1
  %5 = error
  %6 = cast Error [<error> to bool] %5
  return %6
""", errors=[("InvalidOperandTypes", "a is 1", 0, "invalid operand types: int32 is int32")])

kat("is_on_different_pointer_kinds", "generate_ir.rs:624 (pointer kinds must be equal, no autocast)", "d(a @mut int32, b @int32) bool\n\treturn a is b\n", """
d[:@mut int32 :@int32]: bool
VAR %0: @mut int32
VAR %1: @int32
VAR %2: @mut @mut int32
VAR %3: @mut @int32
VAR %4: @@mut int32
VAR %5: @mut int32
VAR %6: @@int32
VAR %7: @int32
VAR %8: <error>
VAR %9: bool
@0
  %2 = alloc
  in %2 store %0
  %3 = alloc
  in %3 store %1
  %4 = cast PointerType [@mut @mut int32 to @@mut int32] %2
  %5 = load %4
  %6 = cast PointerType [@mut @int32 to @@int32] %3
  %7 = load %6
  %8 = error
  %9 = cast Error [<error> to bool] %8
  return %9
""", errors=[("InvalidOperandTypes", "a is b", 0, "invalid operand types: @mut int32 is @int32")])

kat("q9_equals_not", "parse.rs:1027-1038, 837-869 (`a == not b` rotates to Not(Equals(a, Null), b)); generate_ir.rs:694-718 (only the right operand is evaluated)",
    "q(a bool, b bool) bool\n\treturn a == not b\n", """
q[:bool :bool]: bool
VAR %0: bool
VAR %1: bool
VAR %2: @mut bool
VAR %3: @mut bool
VAR %4: @bool
VAR %5: bool
VAR %6: bool
@0
  %2 = alloc
  in %2 store %0
  %3 = alloc
  in %3 store %1
  %4 = cast PointerType [@mut bool to @bool] %3
  %5 = load %4
  %6 = not %5
  return %6
""", ast_contains='Return(Operator(Not, Operator(Equals, Variable("a"), Null), Variable("b")))')

kat("q12_statements_after_return", "generate_ir.rs:275-283 (no new block after return)", "r() int32\n\treturn 1\n\treturn 2\n", """
r[]: int32
VAR %0: int32
VAR %1: int32
@0
  %0 = 1
  return %0
  %1 = 2
  return %1
""")

kat("q14_alias_of_later_alias", "compile.rs:1055-1062 (A copies B's placeholder, Type::Error, silently), generate_ir.rs:1184-1186 (an error-typed operand: no lookup, no report)",
    "type A = B\ntype B = int32\nf(x A) B\n\treturn x + 1\n", """
f[:A]: B
VAR %0: A
VAR %1: @mut A
VAR %2: @A
VAR %3: A
VAR %4: <integer>
VAR %5: <error>
VAR %6: B
@0
  %1 = alloc
  in %1 store %0
  %2 = cast PointerType [@mut A to @A] %1
  %3 = load %2
  %4 = 1
  %5 = error
  %6 = cast Error [<error> to B] %5
  return %6
""")

kat("q15_and_with_non_bool_rhs", "generate_ir.rs:635-651 (early return: no Branch, the current block stays the RHS block)", "k(a bool) bool\n\treturn a and 1\n", """
k[:bool]: bool
VAR %0: bool
VAR %1: @mut bool
VAR %2: @bool
VAR %3: bool
VAR %4: int32
VAR %5: <error>
VAR %6: bool
@0
  %1 = alloc
  in %1 store %0
  %2 = cast PointerType [@mut bool to @bool] %1
  %3 = load %2
@1
  %4 = 1
  %5 = error
  %6 = cast Error [<error> to bool] %5
  return %6
""", errors=[("InvalidOperandType", "1", 0, "invalid operand type: int32")])

kat("or_phi", "generate_ir.rs:635-669 (`or` branches to merge on true; Phi from the branching block and the end of the RHS)",
    "o(a bool, b bool) bool\n\treturn a or b\n", """
o[:bool :bool]: bool
VAR %0: bool
VAR %1: bool
VAR %2: @mut bool
VAR %3: @mut bool
VAR %4: @bool
VAR %5: bool
VAR %6: @bool
VAR %7: bool
VAR %8: bool
@0
  %2 = alloc
  in %2 store %0
  %3 = alloc
  in %3 store %1
  %4 = cast PointerType [@mut bool to @bool] %2
  %5 = load %4
  branch %5, @2, @1
@1
  %6 = cast PointerType [@mut bool to @bool] %3
  %7 = load %6
  jump @2
@2
  %8 = phi @0 => %5, @1 => %7
  return %8
""")

kat("q17_argument_can_be_redefined", "generate_ir.rs:143-149 (arguments are not pushed on locals_stack), 252-264", "s(x int32) int32\n\tvar x = 2\n\treturn x\n", """
s[:int32]: int32
VAR %0: int32
VAR %1: @mut int32
VAR %2: int32
VAR %3: @mut int32
VAR %4: @int32
VAR %5: int32
@0
  %1 = alloc
  in %1 store %0
  %2 = 2
  %3 = alloc
  in %3 store %2
  %4 = cast PointerType [@mut int32 to @int32] %3
  %5 = load %4
  return %5
""")

kat("q17_scope_end_removes_the_argument", "generate_ir.rs:180-185 (end_scope removes the name, argument included), 888-892, 239, 495-503",
    "t(x int32) int32\n\tif true\n\t\tvar x = 2\n\treturn x\n", """
t[:int32]: int32
VAR %0: int32
VAR %1: @mut int32
VAR %2: bool
VAR %3: int32
VAR %4: @mut int32
VAR %5: <error>
VAR %6: <error>
VAR %7: int32
@0
  %1 = alloc
  in %1 store %0
  %2 = true
  branch %2, @1, @2
@1
  %3 = 2
  %4 = alloc
  in %4 store %3
  jump @3
@2
  jump @3
@3
  %5 = error
  %6 = load %5
  %7 = cast Error [<error> to int32] %6
  return %7
""", errors=[("UndefinedVariable", "x", 2, "undefined variable: x")])

kat("shadowed_local", "generate_ir.rs:255-257 (the value is generated, nothing is allocated; span = the statement)", "v() null\n\tvar a = 1\n\tvar a = 2\n", """
v[]: null
VAR %0: int32
VAR %1: @mut int32
VAR %2: int32
VAR %3: null
@0
  %0 = 1
  %1 = alloc
  in %1 store %0
  %2 = 2
  %3 = null
  return %3
""", errors=[("ShadowedName", "var a = 2", 0, "shadowed variable: a")])

kat("ambiguous_overload", "compile.rs:1104-1134 (a GenericInteger argument that no int32 overload takes)",
    "extern g(a int8) null\nextern g(a int16) null\namb() null\n\tg(1)\n", """
amb[]: null
VAR %0: <integer>
VAR %1: <error>
VAR %2: null
@0
  %0 = 1
  %1 = error
  %2 = null
  return %2
""", errors=[("AmbiguousOverload", "g(1)", 0, "call to overloaded function is ambiguous: g[:<integer>]")])

kat("overload_defaults_to_int32", "compile.rs:1120-1131 (types_match_default_generic), generate_ir.rs:1166-1176, 214-218",
    "extern g(a int8) null\nextern g(a int32) null\npick() null\n\tg(1)\n", """
pick[]: null
VAR %0: int32
VAR %1: [int32] null
VAR %2: null
VAR %3: null
@0
  %0 = 1
  %1 = global $g[:int32]: null
  %2 = call %1, 0
  %3 = null
  return %3
""")

kat("overload_single_candidate_takes_generic", "compile.rs:1091-1112 (one candidate left: every argument ignorable, the literal becomes int8)",
    "extern h(a int8, b @int32) null\nextern h(a int8, b @nat8) null\nign(p @int32) null\n\th(1, p)\n", """
ign[:@int32]: null
VAR %0: @int32
VAR %1: @mut @int32
VAR %2: int8
VAR %3: @@int32
VAR %4: @int32
VAR %5: [int8 @int32] null
VAR %6: null
VAR %7: null
@0
  %1 = alloc
  in %1 store %0
  %2 = 1
  %3 = cast PointerType [@mut @int32 to @@int32] %1
  %4 = load %3
  %5 = global $h[:int8 :@int32]: null
  %6 = call %5, 2, 4
  %7 = null
  return %7
""")

kat("no_overload", "compile.rs:1098-1101", "extern h(a int8) null\nno() null\n\th(true)\n", """
no[]: null
VAR %0: bool
VAR %1: <error>
VAR %2: null
@0
  %0 = true
  %1 = error
  %2 = null
  return %2
""", errors=[("NoOverloadForFunction", "h(true)", 0, "no matching overload for function: h[:bool]")])

kat("index_with_two_arguments", "generate_ir.rs:910-914 (all arguments evaluated, InvalidNumberOfArgs, the LAST one indexes)",
    "ix(p {} int32) int32\n\treturn p(1, 2)\n", """
ix[:{} int32]: int32
VAR %0: {} int32
VAR %1: @mut {} int32
VAR %2: @{} int32
VAR %3: {} int32
VAR %4: size
VAR %5: size
VAR %6: {} int32
VAR %7: @int32
VAR %8: int32
@0
  %1 = alloc
  in %1 store %0
  %2 = cast PointerType [@mut {} int32 to @{} int32] %1
  %3 = load %2
  %4 = 1
  %5 = 2
  %6 = indexptr %3, %5
  %7 = cast PointerType [{} int32 to @int32] %6
  %8 = load %7
  return %8
""", errors=[("InvalidNumberOfArgs", "p(1, 2)", 0, "invalid number of indices")])

kat("call_through_local_function_pointer", "generate_ir.rs:871-887 (function name as a value), 1042-1046, 948-960 (arguments with no expected type, no autocast)",
    "extern g(a int32) int32\ncall() int32\n\tvar fp = g\n\treturn fp(2)\n", """
call[]: int32
VAR %0: [int32] int32
VAR %1: @mut [int32] int32
VAR %2: @[int32] int32
VAR %3: [int32] int32
VAR %4: int32
VAR %5: int32
@0
  %0 = global $g[:int32]: int32
  %1 = alloc
  in %1 store %0
  %2 = cast PointerType [@mut [int32] int32 to @[int32] int32] %1
  %3 = load %2
  %4 = 2
  %5 = call %3, 4
  return %5
""")

kat("global_variable_and_constant", "generate_ir.rs:861-870 (GlobalVariable is an lvalue of @mut T), 993-998 (GlobalConstant is an rvalue)",
    "var gv int32 = 5\nconst kc nat8 = 7\nrd() int32\n\tgv = gv + 1\n\treturn kc:int32\n", """
rd[]: int32
VAR %0: @mut int32
VAR %1: @mut int32
VAR %2: @int32
VAR %3: int32
VAR %4: int32
VAR %5: int32
VAR %6: nat8
VAR %7: int32
@0
  %0 = globalvar %0
  %1 = globalvar %0
  %2 = cast PointerType [@mut int32 to @int32] %1
  %3 = load %2
  %4 = 1
  %5 = add %3, %4
  in %0 store %5
  %6 = globalconst %1
  %7 = cast Integer [nat8 to int32] %6
  return %7
""")

kat("call_of_constant", "generate_ir.rs:1043-1050 (arguments are not evaluated; is_lvalue stays true, so 579-582 loads the error)", "const kc nat8 = 7\ncc() null\n\tkc(1)\n", """
cc[]: null
VAR %0: <error>
VAR %1: <error>
VAR %2: null
@0
  %0 = error
  %1 = load %0
  %2 = null
  return %2
""", errors=[("AttemptToCallOrIndexConstant", "kc(1)", 0, "attempt to call or index constant")])

kat("function_might_not_return", "generate_ir.rs:157-171 (span = the signature, parse.rs:1324-1325), 310-311 (jump after return)",
    "fm(a bool) int32\n\tif a\n\t\treturn 1\n", """
fm[:bool]: int32
VAR %0: bool
VAR %1: @mut bool
VAR %2: @bool
VAR %3: bool
VAR %4: int32
@0
  %1 = alloc
  in %1 store %0
  %2 = cast PointerType [@mut bool to @bool] %1
  %3 = load %2
  branch %3, @1, @2
@1
  %4 = 1
  return %4
  jump @3
@2
  jump @3
@3
""", errors=[("FunctionMightNotReturn", "fm(a bool) int32", 0, "function might not return")])

kat("infinite_loop_diverges", "generate_ir.rs:321, 345 (while true without break diverges), 481-485, 165-166 (unreachable)",
    "lp() int32\n\twhile true\n\t\tcontinue\n", """
lp[]: int32
VAR %0: bool
@0
  jump @1
@1
  %0 = true
  branch %0, @2, @4
@2
  jump @1
@3
  jump @1
@4
  jump @5
@5
  unreachable
""")

kat("proc_expression_and_builtin", "generate_ir.rs:586-602; parse.rs:773-795 (`operator +` as an identifier), 1072-1077",
    "extern g(a int32) int32\npr() null\n\tvar f = proc g(int32)\n\tvar b = proc operator +(int32, int32)\n", """
pr[]: null
VAR %0: [int32] int32
VAR %1: @mut [int32] int32
VAR %2: <error>
VAR %3: @mut <error>
VAR %4: null
@0
  %0 = global $g[:int32]: int32
  %1 = alloc
  in %1 store %0
  %2 = error
  %3 = alloc
  in %3 store %2
  %4 = null
  return %4
""", errors=[("ReferenceToBuiltinFunction", "proc operator +(int32, int32)", 0, "refernce to built-in function")])

kat("sizeof_alignof", "generate_ir.rs:790-801, 1227-1228", "type P = record { x int32 }\nsz() size\n\treturn sizeof P + alignof @P\n", """
sz[]: size
VAR %0: size
VAR %1: size
VAR %2: size
@0
  %0 = sizeof P
  %1 = alignof @P
  %2 = add %0, %1
  return %2
""")

kat("float_literals", "generate_ir.rs:550-560, 1222 (f64 Display: 2.0 prints as 2)", "fl() float64\n\treturn 1.5 + 2.\n", """
fl[]: float64
VAR %0: float64
VAR %1: float64
VAR %2: float64
@0
  %0 = 1.5
  %1 = 2
  %2 = add %0, %1
  return %2
""")

kat("string_literal", "generate_ir.rs:571-576, 1225 (Debug of the lossy string)", 'st() string\n\treturn "a\\n"\n', """
st[]: string
VAR %0: string
@0
  %0 = "a\\n"
  return %0
""")

kat("q10_parenthesised_callee", "parse.rs:1103-1110 (CallOrIndex), generate_ir.rs:1053-1063, 1079-1081, 948-960",
    "extern g(a int32) int32\npc() int32\n\treturn (g)(1)\n", """
pc[]: int32
VAR %0: [int32] int32
VAR %1: int32
VAR %2: int32
@0
  %0 = global $g[:int32]: int32
  %1 = 1
  %2 = call %0, 1
  return %2
""", ast_contains='CallOrIndex(Parentheses(Variable("g")), [Integer(1, None)])')

kat("dereference_store_and_load", "generate_ir.rs:1022-1026 (load the pointer, cast its kind), 265-274",
    "inc(p @mut int32) null\n\tp@ = p@ + 1\n", """
inc[:@mut int32]: null
VAR %0: @mut int32
VAR %1: @mut @mut int32
VAR %2: @@mut int32
VAR %3: @mut int32
VAR %4: @@mut int32
VAR %5: @mut int32
VAR %6: @int32
VAR %7: int32
VAR %8: int32
VAR %9: int32
VAR %10: null
@0
  %1 = alloc
  in %1 store %0
  %2 = cast PointerType [@mut @mut int32 to @@mut int32] %1
  %3 = load %2
  %4 = cast PointerType [@mut @mut int32 to @@mut int32] %1
  %5 = load %4
  %6 = cast PointerType [@mut int32 to @int32] %5
  %7 = load %6
  %8 = 1
  %9 = add %7, %8
  in %3 store %9
  %10 = null
  return %10
""")

kat("mutable_reference_argument", "generate_ir.rs:737-740, 857-860", "extern inc(p @mut int32) null\nuse() null\n\tvar x = 1\n\tinc(mut x)\n", """
use[]: null
VAR %0: int32
VAR %1: @mut int32
VAR %2: [@mut int32] null
VAR %3: null
VAR %4: null
@0
  %0 = 1
  %1 = alloc
  in %1 store %0
  %2 = global $inc[:@mut int32]: null
  %3 = call %2, 1
  %4 = null
  return %4
""")

kat("reference_to_rvalue", "generate_ir.rs:970-979, 1082-1085 (the rvalue is generated, then InvalidLvalue)", "rl() null\n\tvar p = ref 5\n", """
rl[]: null
VAR %0: int32
VAR %1: <error>
VAR %2: @mut <error>
VAR %3: null
@0
  %0 = 5
  %1 = error
  %2 = alloc
  in %2 store %1
  %3 = null
  return %3
""", errors=[("InvalidLvalue", "5", 0, "invalid lvalue")])

kat("break_and_continue_outside_loop", "generate_ir.rs:477-479, 486-488", "bo() null\n\tbreak\n\tcontinue\n", """
bo[]: null
VAR %0: null
@0
  %0 = null
  return %0
""", errors=[("BreakOutsideLoop", "break", 0, "break outside loop"), ("ContinueOutsideLoop", "continue", 0, "continue outside loop")])

kat("while_else_break", "generate_ir.rs:315-362 (break jumps to merge, past the else block)",
    "we(a bool) int32\n\twhile a\n\t\tbreak\n\telse\n\t\treturn 1\n\treturn 2\n", """
we[:bool]: int32
VAR %0: bool
VAR %1: @mut bool
VAR %2: @bool
VAR %3: bool
VAR %4: int32
VAR %5: int32
@0
  %1 = alloc
  in %1 store %0
  jump @1
@1
  %2 = cast PointerType [@mut bool to @bool] %1
  %3 = load %2
  branch %3, @2, @4
@2
  jump @5
@3
  jump @1
@4
  %4 = 1
  return %4
  jump @5
@5
  %5 = 2
  return %5
""")

kat("field_store_through_pointer", "generate_ir.rs:1027-1030, 1115-1123", "type P = record { x int32 }\nsx(p @mut P) null\n\tp@.x = 3\n", """
sx[:@mut P]: null
VAR %0: @mut P
VAR %1: @mut @mut P
VAR %2: @@mut P
VAR %3: @mut P
VAR %4: @mut int32
VAR %5: int32
VAR %6: null
@0
  %1 = alloc
  in %1 store %0
  %2 = cast PointerType [@mut @mut P to @@mut P] %1
  %3 = load %2
  %4 = fieldptr %3, x
  %5 = 3
  in %4 store %5
  %6 = null
  return %6
""")

kat("field_of_non_record", "generate_ir.rs:1144-1147 (span = the field expression), 580-582 (the error is still loaded as an lvalue)",
    "fa() null\n\tvar a = 1\n\ta.x\n", """
fa[]: null
VAR %0: int32
VAR %1: @mut int32
VAR %2: @int32
VAR %3: <error>
VAR %4: <error>
VAR %5: null
@0
  %0 = 1
  %1 = alloc
  in %1 store %0
  %2 = cast PointerType [@mut int32 to @int32] %1
  %3 = error
  %4 = load %3
  %5 = null
  return %5
""", errors=[("FieldAccessOnNonRecord", "a.x", 0, "attempt to set or access field of non-record: int32")])

kat("autocast_pointer_to_array", "compile.rs:656-667 (@mut {N} T -> mut {} T is PointerToArray), generate_ir.rs:219-223",
    "extern fill(p {} mut int32) null\npa() null\n\tvar a {4} int32\n\tfill(mut a)\n", """
pa[]: null
VAR %0: @mut {4} int32
VAR %1: @{4} int32
VAR %2: {4} int32
VAR %3: [mut {} int32] null
VAR %4: mut {} int32
VAR %5: null
VAR %6: null
@0
  %0 = alloc
  %1 = cast PointerType [@mut {4} int32 to @{4} int32] %0
  %2 = load %1
  %3 = global $fill[:mut {} int32]: null
  %4 = cast PointerToArray [@mut {4} int32 to mut {} int32] %0
  %5 = call %3, 4
  %6 = null
  return %6
""")

kat("unexpected_type", "generate_ir.rs:225-230", "ut() int32\n\treturn true\n", """
ut[]: int32
VAR %0: bool
VAR %1: <error>
@0
  %0 = true
  %1 = error
  return %1
""", errors=[("UnexpectedType", "true", 0, "invalid type: expected int32, found bool")])

kat("call_of_local_that_shadows_a_function", "generate_ir.rs:1152-1165 (a local with the callee's name: Load, then CallOfNonFunction), 1031-1041 (arguments first)",
    "extern g(a int32) int32\nf() null\n\tvar g = 1\n\tg(2)\n", """
f[]: null
VAR %0: int32
VAR %1: @mut int32
VAR %2: <integer>
VAR %3: int32
VAR %4: <error>
VAR %5: null
@0
  %0 = 1
  %1 = alloc
  in %1 store %0
  %2 = 2
  %3 = load %1
  %4 = error
  %5 = null
  return %5
""", errors=[("CallOfNonFunction", "g(2)", 0, "cannot call value of type int32")])


def main():
    with open(os.path.join(HERE, "kats2.json"), "w") as f:
        json.dump({"_note": "HAND-DERIVED from the reference's Rust before any run through the oracle; see make_kats2.py and oracle/REVIEW.md.", "kats": K}, f, indent=1)
        f.write("\n")
    print(len(K), "cases")


if __name__ == "__main__":
    main()
