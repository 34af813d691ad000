"""Writes tests/golden/kats3.json: hand-derived known answers for TOKENS, AST and SPANS (round 3).

HAND-DERIVED, NOT MACHINE-VERIFIED: the reference (Rust, LLVM 7) cannot be built in this image and ships no tests or golden vectors.
Every expectation below was derived by executing the cited Rust by hand, BEFORE the case was run through the oracle or the HIP
path (oracle/REVIEW.md keeps the first-contact record).  What a span covers is written as the SOURCE TEXT it covers (plus which
occurrence of that text); this script only turns text into byte offsets, so that the offsets are not a second source of slips.
Span rules used (all parse.rs): an expression / type span runs from the first token of the node to `position` = end of the last
CONSUMED token (:829, :966, :1085-1117); a statement span likewise (:1232) — and a consumed Dedent moves `position` to where
lexing stopped, i.e. the start of the next real token (:482-486, :495-501), so a statement that ends with a block reaches up
to the next token; the rotation re-spans the node that moves left to [left.low, right_left.high) (:852-865); the virtual Null of
`not` has the empty span at the keyword (:1032); a `record N {` literal's type node spans `record N` (:1050); a function's
definition span is its signature (:1325).

Run:  python tests/golden/make_kats3.py
"""
import json
import os

HERE = os.path.dirname(os.path.abspath(__file__))


def find(src, text, nth=0):
    p = -1
    for _ in range(nth + 1):
        p = src.index(text, p + 1)
    return p


def span_of(src, spec):
    """spec: text | (text, nth) | ("EMPTY_AT", text, nth): empty span at the start of that text | ("END",): empty span at the end of the source
    | (text, nth, "+", more): the nth occurrence of `text`, extended to the right up to the start of the next occurrence of `more`"""
    if isinstance(spec, str):
        spec = (spec, 0)
    if spec[0] == "EMPTY_AT":
        p = find(src, spec[1], spec[2])
        return p, p
    if spec[0] == "END":
        return len(src), len(src)
    if spec[0] == "IN":       # ("IN", outer, outer_nth, inner[, inner_nth]): `inner` inside that occurrence of `outer`
        base = find(src, spec[1], spec[2])
        q = find(spec[1], spec[3], spec[4] if len(spec) > 4 else 0)
        return base + q, base + q + len(spec[3])
    p = find(src, spec[0], spec[1])
    hi = p + len(spec[0])
    if len(spec) > 2 and spec[2] == "+":
        hi = len(src) if spec[3] == "END" else src.index(spec[3], hi)
    return p, hi


def ast_text(src, defs):
    out = ""
    for debug, dspan, spans in defs:
        lo, hi = span_of(src, dspan)
        out += f"{debug} @ {lo}..{hi}\n"
        for cls, kind, spec in spans:
            a, b = span_of(src, spec)
            out += f"  {cls} {kind} {a} {b}\n"
    return out


def tok_text(src, toks):
    out = ""
    for spec, debug in toks:
        a, b = span_of(src, spec)
        out += f"{a} {b} {debug}\n"
    return out


V = lambda n: f'Variable("{n}")'
I = lambda n: f"Integer({n}, None)"
CASES = []


def case(name, src, defs=None, tokens=None, error=None, cites=""):
    c = {"name": name, "src": src, "cites": cites}
    if defs is not None:
        c["ast"] = ast_text(src, defs)
    if tokens is not None:
        c["tokens"] = tok_text(src, tokens)
    if error is not None:
        c["error"] = error
    CASES.append(c)


def err(src, kind, spec, payload, line):
    """the oracle's / library's first-error line: kind, span low, span length, payload, line number (frontend.rs:213-225)"""
    lo, hi = span_of(src, spec)
    return f"{kind} {lo} {hi - lo}" + (f" {payload}" if payload else "") + f" line {line}\n"


def fn(body_debug, name="f", ret="int32"):
    return f"Function({name}[), [{body_debug}])"


RT = ("T", "Name", "int32")      # the return type of `f() int32` (printed after the argument types, before the body)

# ---- rotation and its re-span (parse.rs:815-869) ------------------------------------------------------------------------------
case("rot1 a*b+c", "f() int32\n\treturn a * b + c\n",
     [(fn(f"Return(Operator(Plus, Operator(Times, {V('a')}, {V('b')}), {V('c')}))"), "f() int32",
       [RT, ("S", "Return", "return a * b + c"), ("E", "Operator", "a * b + c"), ("E", "Operator", "a * b"), ("E", "Variable", "a"),
        ("E", "Variable", "b"), ("E", "Variable", "c")])], cites="parse.rs:837-869: 31 >= 30 rotates; left re-spanned to [a.low, b.high)")
case("rot2 a-b-c", "f() int32\n\treturn a - b - c\n",
     [(fn(f"Return(Operator(Minus, Operator(Minus, {V('a')}, {V('b')}), {V('c')}))"), "f() int32",
       [RT, ("S", "Return", "return a - b - c"), ("E", "Operator", "a - b - c"), ("E", "Operator", "a - b"), ("E", "Variable", "a"),
        ("E", "Variable", "b"), ("E", "Variable", "c")])], cites="equal precedence rotates (>=): left associative")
case("rot3 a*b+c*d", "f() int32\n\treturn a * b + c * d\n",
     [(fn(f"Return(Operator(Plus, Operator(Times, {V('a')}, {V('b')}), Operator(Times, {V('c')}, {V('d')})))"), "f() int32",
       [RT, ("S", "Return", "return a * b + c * d"), ("E", "Operator", "a * b + c * d"), ("E", "Operator", "a * b"), ("E", "Variable", "a"),
        ("E", "Variable", "b"), ("E", "Operator", "c * d"), ("E", "Variable", "c"), ("E", "Variable", "d")])],
     cites="inner b + c*d does not rotate (30 < 31); outer rotates once")
case("rot4 a*b*c+d", "f() int32\n\treturn a * b * c + d\n",
     [(fn(f"Return(Operator(Plus, Operator(Times, Operator(Times, {V('a')}, {V('b')}), {V('c')}), {V('d')}))"), "f() int32",
       [RT, ("S", "Return", "return a * b * c + d"), ("E", "Operator", "a * b * c + d"), ("E", "Operator", "a * b * c"), ("E", "Operator", "a * b"),
        ("E", "Variable", "a"), ("E", "Variable", "b"), ("E", "Variable", "c"), ("E", "Variable", "d")])],
     cites="the recursive correct_precedence(left) at :862 rotates again; the outer insert at :865 then spans a * b * c")
case("rot5 rising precedences", "f() int32\n\treturn b or c and g == h + k\n",
     [(fn(f"Return(Operator(Or, {V('b')}, Operator(And, {V('c')}, Operator(Equals, {V('g')}, Operator(Plus, {V('h')}, {V('k')})))))"), "f() int32",
       [RT, ("S", "Return", "return b or c and g == h + k"), ("E", "Operator", "b or c and g == h + k"), ("E", "Variable", "b"),
        ("E", "Operator", "c and g == h + k"), ("E", "Variable", "c"), ("E", "Operator", "g == h + k"), ("E", "Variable", "g"),
        ("E", "Operator", "h + k"), ("E", "Variable", "h"), ("E", "Variable", "k")])], cites="rising precedences 0 < 1 < 20 < 30: nothing rotates")
case("paren stops rotation", "f() int32\n\treturn a * (b + c)\n",
     [(fn(f"Return(Operator(Times, {V('a')}, Parentheses(Operator(Plus, {V('b')}, {V('c')}))))"), "f() int32",
       [RT, ("S", "Return", "return a * (b + c)"), ("E", "Operator", "a * (b + c)"), ("E", "Variable", "a"), ("E", "Parentheses", "(b + c)"),
        ("E", "Operator", "b + c"), ("E", "Variable", "b"), ("E", "Variable", "c")])], cites="only bare Operator nodes rotate (:839)")
# ---- `not` (parse.rs:1027-1038; Q9) ----------------------------------------------------------------------------------------------
case("not1 a == not b", "f() int32\n\treturn a == not b\n",
     [(fn(f"Return(Operator(Not, Operator(Equals, {V('a')}, Null), {V('b')}))"), "f() int32",
       [RT, ("S", "Return", "return a == not b"), ("E", "Operator", "a == not b"), ("E", "Operator", "a == "), ("E", "Variable", "a"),
        ("E", "Null", ("EMPTY_AT", "not", 0)), ("E", "Variable", "b")])],
     cites="Q9: 20 >= 2 rotates; the moved node spans [a.low, Null.high) = up to the keyword `not`, trailing blank included")
case("not2 not a and b", "f() int32\n\treturn not a and b\n",
     [(fn(f"Return(Operator(And, Operator(Not, Null, {V('a')}), {V('b')}))"), "f() int32",
       [RT, ("S", "Return", "return not a and b"), ("E", "Operator", "not a and b"), ("E", "Operator", "not a"), ("E", "Null", ("EMPTY_AT", "not", 0)),
        ("E", "Variable", "a"), ("E", "Variable", "b")])], cites="2 >= 1 rotates inside the `not` arm (:1035)")
case("not3 prefix then not", "f() int32\n\treturn - not a + b\n",
     [(fn(f"Return(Negate(Operator(Not, Null, Operator(Plus, {V('a')}, {V('b')}))))"), "f() int32",
       [RT, ("S", "Return", "return - not a + b"), ("E", "Negate", "- not a + b"), ("E", "Operator", "not a + b"), ("E", "Null", ("EMPTY_AT", "not", 0)),
        ("E", "Operator", "a + b"), ("E", "Variable", "a"), ("E", "Variable", "b")])],
     cites="parse.rs:993-995 + 1029: the `not` arm calls parse_expression, so the prefix operator swallows the whole rest")
case("not4 not not a", "f() int32\n\treturn not not a\n",
     [(fn(f"Return(Operator(Not, Operator(Not, Null, Null), {V('a')}))"), "f() int32",
       [RT, ("S", "Return", "return not not a"), ("E", "Operator", "not not a"), ("E", "Operator", "not "), ("E", "Null", ("EMPTY_AT", "not", 0)),
        ("E", "Null", ("EMPTY_AT", "not", 1)), ("E", "Variable", "a")])],
     cites="outer Not(Null1, Not(Null2, a)): 2 >= 2 rotates -> Not(Not(Null1, Null2), a); moved node spans [Null1.low, Null2.high)")
# ---- postfix forms (parse.rs:1087-1121) --------------------------------------------------------------------------------------------
case("postfix chain", "f() null\n\tx@.f(1, y):int32\n",
     [(fn(f'Expression(Cast(CallOrIndex(Field("f", Dereference({V("x")})), [{I(1)}, {V("y")}]), Name("int32")))', ret="null"), "f() null",
       [("T", "Null", "null"), ("S", "Expression", "x@.f(1, y):int32"), ("E", "Cast", "x@.f(1, y):int32"), ("E", "CallOrIndex", "x@.f(1, y)"),
        ("E", "Field", "x@.f"), ("E", "Dereference", "x@"), ("E", "Variable", "x"), ("E", "Integer", "1"), ("E", "Variable", "y"),
        ("T", "Name", "int32")])], cites="every postfix node spans from the atom's first token (:1092,:1098,:1111,:1117)")
case("named call, trailing comma", "f() null\n\tg(a, b,)\n",
     [(fn(f'Expression(NamedCall("g", [{V("a")}, {V("b")}]))', ret="null"), "f() null",
       [("T", "Null", "null"), ("S", "Expression", "g(a, b,)"), ("E", "NamedCall", "g(a, b,)"), ("E", "Variable", "a"), ("E", "Variable", "b")])],
     cites="parse.rs:886-891 trailing comma; :1103-1106 the Variable node becomes the call")
case("Q10 (f)(x)", "f() null\n\t(f)(x)\n",
     [(fn(f'Expression(CallOrIndex(Parentheses({V("f")}), [{V("x")}]))', ret="null"), "f() null",
       [("T", "Null", "null"), ("S", "Expression", "(f)(x)"), ("E", "CallOrIndex", "(f)(x)"), ("E", "Parentheses", "(f)"), ("E", "Variable", ("IN", "(f)", 0, "f")),
        ("E", "Variable", "x")])], cites="Q10")
case("index assignment", "f() null\n\ta(1) = b.c\n",
     [(fn(f'Assignment(NamedCall("a", [{I(1)}]), Field("c", {V("b")}))', ret="null"), "f() null",
       [("T", "Null", "null"), ("S", "Assignment", "a(1) = b.c"), ("E", "NamedCall", "a(1)"), ("E", "Integer", "1"), ("E", "Field", "b.c"),
        ("E", "Variable", "b")])])
case("field operator name", "f() null\n\tx.operator +\n",
     [(fn(f'Expression(Field("+", {V("x")}))', ret="null"), "f() null",
       [("T", "Null", "null"), ("S", "Expression", "x.operator +"), ("E", "Field", "x.operator +"), ("E", "Variable", "x")])],
     cites="parse_identifier :776-794: `operator +` is the name `+`")
# ---- prefix forms (parse.rs:993-1046) ------------------------------------------------------------------------------------------------
case("prefix ops", "f() null\n\t-x:int8\n\tref x\n\tmut x.f\n\trefs a\n\tmuts a@\n",
     [(fn(f'Expression(Negate(Cast({V("x")}, Name("int8")))), Expression(Reference({V("x")})), Expression(MutableReference(Field("f", {V("x")}))), '
          f'Expression(ArrayReference({V("a")})), Expression(MutableArrayReference(Dereference({V("a")})))', ret="null"), "f() null",
       [("T", "Null", "null"),
        ("S", "Expression", "-x:int8"), ("E", "Negate", "-x:int8"), ("E", "Cast", "x:int8"), ("E", "Variable", ("IN", "-x:int8", 0, "x")), ("T", "Name", "int8"),
        ("S", "Expression", "ref x"), ("E", "Reference", "ref x"), ("E", "Variable", ("IN", "ref x", 0, "x")),
        ("S", "Expression", "mut x.f"), ("E", "MutableReference", "mut x.f"), ("E", "Field", "x.f"), ("E", "Variable", ("IN", "mut x.f", 0, "x")),
        ("S", "Expression", "refs a"), ("E", "ArrayReference", "refs a"), ("E", "Variable", ("IN", "refs a", 0, "a")),
        ("S", "Expression", "muts a@"), ("E", "MutableArrayReference", "muts a@"), ("E", "Dereference", "a@"), ("E", "Variable", ("IN", "muts a@", 0, "a"))])],
     cites="the operand is parse_atom (:994): it takes the postfix chain, the prefix node wraps the result.  (`-x` is the FIRST statement on purpose: "
           "behind another statement the `-` would be that statement's binary operator — statements are separated by juxtaposition only)")
case("sizeof alignof proc", "f() null\n\tsizeof @int32\n\talignof {} nat8\n\tproc g(int32, bool)\n\tproc h()\n",
     [(fn('Expression(Sizeof(Reference(Name("int32")))), Expression(Alignof(ArrayReference(Name("nat8")))), '
          'Expression(Proc("g", [Name("int32"), Name("bool")])), Expression(Proc("h", []))', ret="null"), "f() null",
       [("T", "Null", "null"),
        ("S", "Expression", "sizeof @int32"), ("E", "Sizeof", "sizeof @int32"), ("T", "Reference", "@int32"), ("T", "Name", ("int32", 0)),
        ("S", "Expression", "alignof {} nat8"), ("E", "Alignof", "alignof {} nat8"), ("T", "ArrayReference", "{} nat8"), ("T", "Name", "nat8"),
        ("S", "Expression", "proc g(int32, bool)"), ("E", "Proc", "proc g(int32, bool)"), ("T", "Name", ("int32", 1)), ("T", "Name", "bool"),
        ("S", "Expression", "proc h()"), ("E", "Proc", "proc h()")])])
case("literals", 'f() null\n\t5i8\n\t1.5f32\n\t"a\\n\\x41\\0"\n\ttrue\n\tfalse\n\tnull\n',
     [(fn('Expression(Integer(5, Some((true, 8)))), Expression(Float(1.5, Some(32))), Expression(String([97, 10, 65, 0])), '
          'Expression(Bool(true)), Expression(Bool(false)), Expression(Null)', ret="null"), "f() null",
       [("T", "Null", ("null", 0)),
        ("S", "Expression", "5i8"), ("E", "Integer", "5i8"), ("S", "Expression", "1.5f32"), ("E", "Float", "1.5f32"),
        ("S", "Expression", '"a\\n\\x41\\0"'), ("E", "String", '"a\\n\\x41\\0"'), ("S", "Expression", "true"), ("E", "Bool", "true"),
        ("S", "Expression", "false"), ("E", "Bool", "false"), ("S", "Expression", ("null", 1)), ("E", "Null", ("null", 1))])])
# ---- record literal (parse.rs:1047-1071) -----------------------------------------------------------------------------------------------
case("record literal over lines", "f() null\n\tvar p = record P {\n\t\tx = 1\n\t\ty = q\n\t}\n",
     [(fn(f'VariableDefinition("p", Record(Name("P"), [("x", {I(1)}), ("y", {V("q")})]))', ret="null"), "f() null",
       [("T", "Null", "null"), ("S", "VariableDefinition", "var p = record P {\n\t\tx = 1\n\t\ty = q\n\t}"),
        ("E", "Record", "record P {\n\t\tx = 1\n\t\ty = q\n\t}"), ("T", "Name", "record P"), ("E", "Integer", "1"), ("E", "Variable", "q")])],
     tokens=[("f", 'Identifier("f")'), ("(", "LeftParenthesis"), (")", "RightParenthesis"), ("null", "Null"), (("EMPTY_AT", "var", 0), "Indent"), ("var", "Var"),
             ("p", 'Identifier("p")'), (("=", 0), "Equals"), ("record", "Record"), ("P", 'Identifier("P")'), ("{", "LeftBrace"), ("x", 'Identifier("x")'),
             (("=", 1), "Equals"), ("1", I(1)), ("y", 'Identifier("y")'), (("=", 2), "Equals"), ("q", 'Identifier("q")'), ("}", "RightBrace"),
             (("END",), "Dedent"), (("END",), "EndOfFile")],
     cites="ignore_dents + 1 inside the braces (:1054): no dent tokens; the type node spans `record P` (:1050)")
# ---- types (parse.rs:899-969) --------------------------------------------------------------------------------------------------------------
case("type forms nested", "type T = @mut {} mut {4} (int32, @nat8) null\n",
     [('Type("T", MutableReference(MutableArrayReference(Array(4, Proc([Name("int32"), Reference(Name("nat8"))], Null)))))',
       "type T = @mut {} mut {4} (int32, @nat8) null",
       [("T", "MutableReference", "@mut {} mut {4} (int32, @nat8) null"), ("T", "MutableArrayReference", "{} mut {4} (int32, @nat8) null"),
        ("T", "Array", "{4} (int32, @nat8) null"), ("T", "Proc", "(int32, @nat8) null"), ("T", "Name", "int32"), ("T", "Reference", "@nat8"),
        ("T", "Name", "nat8"), ("T", "Null", "null")])])
case("record type over lines", "type P = record {\n\tu int32\n\tv @P\n}\nvar g P\n",
     [('Type("P", Record([("u", Name("int32")), ("v", Reference(Name("P")))]))', "type P = record {\n\tu int32\n\tv @P\n}",
       [("T", "Record", "record {\n\tu int32\n\tv @P\n}"), ("T", "Name", "int32"), ("T", "Reference", "@P"), ("T", "Name", ("P", 1))]),
      ('Variable("g", Name("P"), None)', "var g P", [("T", "Name", ("P", 2))])],
     tokens=[("type", "Type"), ("P", 'Identifier("P")'), ("=", "Equals"), ("record", "Record"), ("{", "LeftBrace"), ("u", 'Identifier("u")'),
             ("int32", 'Identifier("int32")'), ("v", 'Identifier("v")'), ("@", "At"), (("P", 1), 'Identifier("P")'), ("}", "RightBrace"), ("var", "Var"),
             ("g", 'Identifier("g")'), (("P", 2), 'Identifier("P")'), (("END",), "EndOfFile")],
     cites="record fields have no separators (:929-938); ignore_dents + 1 inside the braces, indentation is not even measured there")
case("array type forms", "f(a {} nat8, b {} mut nat8, c {3} int32) null\n\treturn null\n",
     [('Function(f[a: ArrayReference(Name("nat8")) b: MutableArrayReference(Name("nat8")) c: Array(3, Name("int32"))), [Return(Null)])',
       "f(a {} nat8, b {} mut nat8, c {3} int32) null",
       [("T", "ArrayReference", "{} nat8"), ("T", "Name", ("nat8", 0)), ("T", "MutableArrayReference", "{} mut nat8"), ("T", "Name", ("nat8", 1)),
        ("T", "Array", "{3} int32"), ("T", "Name", "int32"), ("T", "Null", ("null", 0)), ("S", "Return", "return null"), ("E", "Null", ("null", 1))])],
     cites="parse.rs:943-958: `{}` then an optional `mut`, or `{N}`")
_s = "f(a mut {} nat8) null\n\treturn null\n"
case("mut {} is how a type PRINTS, not how it parses", _s, error=err(_s, "UnexpectedToken", "mut", "Mut", 1),
     cites="TypePrinter writes `mut {} T` (compile.rs:240-243) but parse_type has no Token::Mut arm (:903-962)")
# ---- statements (parse.rs:1126-1235) ----------------------------------------------------------------------------------------------------------
case("var statements", "f() null\n\tvar x = 1\n\tvar y int32\n\tvar z int32 = 7\n",
     [(fn(f'VariableDefinition("x", {I(1)}), Expression(VariableDefinition("y", Name("int32"))), Assignment(VariableDefinition("z", Name("int32")), {I(7)})', ret="null"),
       "f() null",
       [("T", "Null", "null"), ("S", "VariableDefinition", "var x = 1"), ("E", "Integer", "1"), ("S", "Expression", "var y int32"),
        ("E", "VariableDefinition", "var y int32"), ("T", "Name", ("int32", 0)), ("S", "Assignment", "var z int32 = 7"),
        ("E", "VariableDefinition", "var z int32"), ("T", "Name", ("int32", 1)), ("E", "Integer", "7")])], cites="parse.rs:1131-1159")
case("if else spans reach the next token", "f(a bool) null\n\tif a\n\t\tbreak\n\telse\n\t\tcontinue\n\treturn null\n",
     [('Function(f[a: Name("bool")), [If(Variable("a"), [Break], [Continue]), Return(Null)])', "f(a bool) null",
       [("T", "Name", "bool"), ("T", "Null", ("null", 0)), ("S", "If", ("if a", 0, "+", "return")), ("E", "Variable", ("a", 1)), ("S", "Break", "break"),
        ("S", "Continue", "continue"), ("S", "Return", "return null"), ("E", "Null", ("null", 1))])],
     cites="the else block's Dedent is consumed at :1257: position moves to the start of `return`")
case("single statement blocks", "f(a bool) int8\n\tif a return 1 else return 2\n\twhile a break\n",
     [(f'Function(f[a: Name("bool")), [If(Variable("a"), [Return({I(1)})], [Return({I(2)})]), While(Variable("a"), [Break], [])])', "f(a bool) int8",
       [("T", "Name", "bool"), ("T", "Name", "int8"), ("S", "If", "if a return 1 else return 2"), ("E", "Variable", ("a", 1)), ("S", "Return", "return 1"),
        ("E", "Integer", "1"), ("S", "Return", "return 2"), ("E", "Integer", "2"), ("S", "While", "while a break"), ("E", "Variable", ("a", 2)),
        ("S", "Break", ("break", 0))])], cites="parse.rs:1245-1248 single-statement blocks: no dent tokens are consumed, spans end at the last token")
case("for forms", "f(xs T) null\n\tfor i int32 in xs\n\t\tbreak\n\telse\n\t\tcontinue\n\tfor j in xs\n\t\tbreak\n",
     [('Function(f[xs: Name("T")), [For("i", Some(Name("int32")), Variable("xs"), [Break], [Continue]), For("j", None, Variable("xs"), [Break], [])])', "f(xs T) null",
       [("T", "Name", "T"), ("T", "Null", "null"), ("S", "For", ("for i int32 in xs", 0, "+", "for j")), ("T", "Name", "int32"), ("E", "Variable", ("xs", 1)),
        ("S", "Break", ("break", 0)), ("S", "Continue", "continue"), ("S", "For", ("for j in xs", 0, "+", "END")), ("E", "Variable", ("xs", 2)),
        ("S", "Break", ("break", 1))])], cites="parse.rs:1185-1207; the last statement's span reaches the end of the source (two Dedents lexed at EOF)")
case("continuation line", "f() null\n\ta = b\n\t- c\n",
     [(fn(f'Assignment({V("a")}, Operator(Minus, {V("b")}, {V("c")}))', ret="null"), "f() null",
       [("T", "Null", "null"), ("S", "Assignment", "a = b\n\t- c"), ("E", "Variable", "a"), ("E", "Operator", "b\n\t- c"), ("E", "Variable", "b"),
        ("E", "Variable", "c")])],
     tokens=[("f", 'Identifier("f")'), ("(", "LeftParenthesis"), (")", "RightParenthesis"), ("null", "Null"), (("EMPTY_AT", "a", 0), "Indent"), ("a", 'Identifier("a")'),
             ("=", "Equals"), ("b", 'Identifier("b")'), ("-", "Minus"), ("c", 'Identifier("c")'), (("END",), "Dedent"), (("END",), "EndOfFile")],
     cites="SURVEY §7 hard part 2: a newline at unchanged indentation is no token (parse.rs:466-472)")
case("comments and blank lines", "f() null ; c1\n\n\t; only a comment\n\tbreak ; c2\n  \n\t\t; deeper comment\n\tcontinue\n",
     [(fn("Break, Continue", ret="null"), "f() null", [("T", "Null", "null"), ("S", "Break", "break"), ("S", "Continue", "continue")])],
     tokens=[("f", 'Identifier("f")'), ("(", "LeftParenthesis"), (")", "RightParenthesis"), ("null", "Null"), (("EMPTY_AT", "break", 0), "Indent"), ("break", "Break"),
             ("continue", "Continue"), (("END",), "Dedent"), (("END",), "EndOfFile")],
     cites="parse.rs:429-459: a comment restarts skip_whitespace with check_indent = false; blank / comment-only lines never change the indentation")
# ---- definitions (parse.rs:1262-1355) -----------------------------------------------------------------------------------------------------------
case("definition spans", 'extern puts(s {} nat8) int32\nvar g int32 = -5\nconst K nat8 = 200n8\nimport "a.o"\nlibrary "m"\nadd(a int32, b @mut int32) int32\n\treturn a\n',
     [('Extern(puts[s: ArrayReference(Name("nat8"))))', "extern puts(s {} nat8) int32", [("T", "ArrayReference", "{} nat8"), ("T", "Name", ("nat8", 0)), ("T", "Name", ("int32", 0))]),
      (f'Variable("g", Name("int32"), Some(Negate({I(5)})))', "var g int32 = -5", [("T", "Name", ("int32", 1)), ("E", "Negate", "-5"), ("E", "Integer", "5")]),
      ('Constant("K", Name("nat8"), Integer(200, Some((false, 8))))', "const K nat8 = 200n8", [("T", "Name", ("nat8", 1)), ("E", "Integer", "200n8")]),
      ("Import([97, 46, 111])", 'import "a.o"', []), ("Library([109])", 'library "m"', []),
      ('Function(add[a: Name("int32") b: MutableReference(Name("int32"))), [Return(Variable("a"))])', "add(a int32, b @mut int32) int32",
       [("T", "Name", ("int32", 2)), ("T", "MutableReference", "@mut int32"), ("T", "Name", ("int32", 3)), ("T", "Name", ("int32", 4)), ("S", "Return", "return a"),
        ("E", "Variable", ("IN", "return a", 0, "a"))])],
     cites="parse.rs:1268-1329: every span runs from the first token to `position`; a function's stops behind its return type (:1325)")
case("operator names", "operator ==(a T, b T) bool\n\treturn true\noperator not(a bool) bool\n\treturn a\noperator for(a T) T\n\treturn a\n",
     [('Function(=[a: Name("T") b: Name("T")), [Return(Bool(true))])', "operator ==(a T, b T) bool",
       [("T", "Name", ("T", 0)), ("T", "Name", ("T", 1)), ("T", "Name", ("bool", 0)), ("S", "Return", "return true"), ("E", "Bool", "true")]),
      ('Function(not[a: Name("bool")), [Return(Variable("a"))])', "operator not(a bool) bool",
       [("T", "Name", ("bool", 1)), ("T", "Name", ("bool", 2)), ("S", "Return", ("return a", 0)), ("E", "Variable", ("IN", "return a", 0, "a"))]),
      ('Function(for[a: Name("T")), [Return(Variable("a"))])', "operator for(a T) T",
       [("T", "Name", ("T", 2)), ("T", "Name", ("T", 3)), ("S", "Return", ("return a", 1)), ("E", "Variable", ("IN", "return a", 1, "a"))])],
     cites="Q11: operator == defines `=`; parse.rs:776-794")
# ---- errors and their spans --------------------------------------------------------------------------------------------------------------------------
_s = "f() null\n\tg(1,\n"
case("Q8 unclosed paren at EOF", _s, error=err(_s, "UnexpectedToken", ("END",), "EndOfFile", 3),
     tokens=[("f", 'Identifier("f")'), ("(", "LeftParenthesis"), (")", "RightParenthesis"), ("null", "Null"), (("EMPTY_AT", "g", 0), "Indent"), ("g", 'Identifier("g")'),
             (("(", 1), "LeftParenthesis"), ("1", I(1)), (",", "Comma"), (("END",), "EndOfFile")],
     cites="Q8: ignore_dents > 0 discards the EOF dedents; parse_atom :1078-1081 on EndOfFile, span [len, len)")
_s = "f() null\n\treturn 5i7\n"
case("Q2 one bad size digit", _s, error=err(_s, "InvalidNumericSize", "7", "7", 2), cites="parse.rs:614-619: span = the digit")
_s = "f() null\n\treturn 5n77\n"
case("Q2 two bad size digits", _s, error=err(_s, "InvalidNumericSize", "77", "77", 2), cites="parse.rs:610-613: span = the two digits")
_s = "f() null\n\treturn 1.5f3x\n"
case("Q4 float suffix", _s, error=err(_s, "InvalidNumericSize", "1.5f3x", "102", 2),
     cites="parse.rs:565-577: (a, b) = ('3', 'x'): 3 * 10 + (120 - 48) = 102, span = the whole token incl. both swallowed characters")
_s = "f() null\n\tg(1 2)\n"
case("missing comma", _s, error=err(_s, "UnexpectedToken", "2", "Integer(2, None)", 2), cites="parse.rs:892-894: span [token_low, real_position) of the peeked token")
_s = "f() null\n\treturn (1 ]\n"
case("expect reports expected and found", _s, error=err(_s, "ExpectedFoundToken", "]", "RightParenthesis RightSquareBracket", 2), cites="parse.rs:743-748")
_s = "var if int32\n"
case("keyword as name", _s, error=err(_s, "ExpectedIdentifier", "if", "If", 1), cites="parse.rs:796-799")
_s = "f() null\n\treturn\n"
case("return needs an expression", _s, error=err(_s, "UnexpectedToken", ("END",), "Dedent", 3), cites="parse.rs:1164 -> parse_atom on the Dedent token (empty span at the end of the source)")
_s = 'f() null\n\tvar s = "ab\\qcd"\n'
case("bad escape span", _s, error=err(_s, "InvalidEscapeSequence", "q", "", 2), cites="parse.rs:700: span = the escape character")
_s = 'f() null\n\tvar s = "ab\n'
case("unterminated string", _s, error=err(_s, "UnexpectedEndOfFile", ("END",), "", 3), cites="parse.rs:701,712: empty span at the end of the source")
_s = "f() null\n\tvar c = 1 # 2\n"
case("unexpected character", _s, error=err(_s, "UnexpectedCharacter", "#", "35", 2), cites="parse.rs:718; payload = the code point")

if __name__ == "__main__":
    with open(os.path.join(HERE, "kats3.json"), "w") as f:
        json.dump({"note": "hand-derived (parity unpinned by the reference), see make_kats3.py", "cases": CASES}, f, indent=1)
    print(len(CASES), "cases")
