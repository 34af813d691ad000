"""Error rendering: the user-visible wire format of the error tuples the GPU path returns.

Restates, byte for byte, ``frontend::print_span`` (frontend.rs:201-256) and the message lines of ``Compiler::print_error``
(compile.rs:782-917) for (a) all seven ``parse::Error`` kinds, rendered from ``opusfe_error`` + the token arrays, and (b) the
semantic error kinds.  Those whose message needs nothing beyond the span are rendered from (kind, low, len) alone
(``semantic_message``); those that print types, names or function identifiers (UnexpectedType, NoOverloadForFunction, ...,
compile.rs:820-899 with TypePrinter / FunctionIdentifierPrinter, compile.rs:200-320) from the payload of the diagnostic record
(``opusfe_semantic_error``: name, nargs, t[4]) and the unit's type arrays (``semantic_message_of_record``).  A function
identifier prints all its argument types: the record carries four, continuation records the rest (ir_view.sem_error_arg_types).
PARITY UNPINNED: the reference ships no expected outputs; tests/test_render.py holds strings derived by hand from the cited
lines.  Pure host formatting: nothing here runs or replaces the GPU path.
"""
from . import ir_view as V

RED, YELLOW, RESET = "\x1b[31;1m", "\x1b[33m", "\x1b[0m"

# parse.rs:69-157 impl Display for Token, variant order of parse.rs:5-67 (Integer / Float / String / Identifier carry data)
TOKEN_TEXT = ["(", ")", "[", "]", "{", "}", ":", "=", "==", "<", ">", "<=", ">=", "+", "-", "*", "/", "%", "@", "¯", ".", ",", "?",
              None, None, None, None, "<indent>", "<dedent>", "var", "const", "return", "if", "else", "while", "for", "break",
              "continue", "extern", "ref", "mut", "refs", "muts", "null", "false", "true", "type", "record", "is", "in", "and", "or",
              "not", "sizeof", "alignof", "import", "library", "proc", "operator", "<end of file>"]
T_INTEGER, T_FLOAT, T_STRING, T_IDENTIFIER = 23, 24, 25, 26

# compile.rs:124-152 Error variant order -> message without arguments (compile.rs:816-915), or a callable of the span text
SEMANTIC_TEXT = {
    1: lambda span: f"undefined variable: {span}",                 # UndefinedVariable :816
    6: "reference to overloaded function is ambiguous",           # AmbiguousFunctionReference :836
    11: "function might not return",                               # :856
    12: "break outside loop", 13: "continue outside loop",         # :860, :864
    14: "invalid lvalue", 15: "lvalue is immutable",               # :868, :872
    16: "attempt to take array reference to non-array lvalue",     # :876
    18: lambda span: f"undefined type: {span}",                    # UndefinedType :884
    22: "refernce to built-in function",                           # :900 (the reference's spelling)
    23: "attempt to call or index constant",                       # :904
    24: "invalid number of indices",                               # :908
    25: "invalid constant expression",                             # :912
}
SEMANTIC_NAMES = ["ParseError", "UndefinedVariable", "ShadowedName", "UndefinedFunction", "NoOverloadForFunction", "AmbiguousOverload",
                  "AmbiguousFunctionReference", "UnexpectedType", "InvalidType", "InvalidOperandType", "InvalidOperandTypes",
                  "FunctionMightNotReturn", "BreakOutsideLoop", "ContinueOutsideLoop", "InvalidLvalue", "ImmutableLvalue",
                  "ArrayPointerToNonArray", "InvalidCast", "UndefinedType", "FieldAccessOnNonRecord", "FieldDoesNotExist",
                  "CallOfNonFunction", "ReferenceToBuiltinFunction", "AttemptToCallOrIndexConstant", "InvalidNumberOfArgs",
                  "InvalidConstantExpression"]


def print_span(filename: str, source: bytes, low: int, length: int) -> str:
    """frontend.rs:201-256 for the source the span lies in (the reference finds it by pointer distance, :204-208)."""
    high = low + length
    out = [f"{YELLOW}{filename}:{RESET}\n"]
    line_low, line_number = 0, 1
    nl = source.rfind(b"\n", 0, low)                    # :213-225: lines before `low`
    if nl >= 0:
        line_low = nl + 1
        line_number += source.count(b"\n", 0, low)
    end = source.find(b"\n", high)                      # :226-232: to the end of the line `high` is on
    line_high = len(source) if end < 0 else end
    out.append(f"{YELLOW}{line_number:>4}:{RESET} ")
    i = line_low
    for c in source[line_low:line_high].decode("utf-8", errors="replace"):
        if i == low:                                    # :237-241
            out.append(RED)
        elif i == high:
            out.append(RESET)
        if c == "\t":
            out.append("    ")
        elif c == "\n":
            line_number += 1
            out.append(f"\n{YELLOW}{line_number:>4}:{RESET} ")
            if low <= i < high:
                out.append(RED)
        else:
            out.append(c)
        i += len(c.encode("utf-8"))
    out.append(f"{RESET}\n")
    return "".join(out)


def token_text(view, ti) -> str:
    """parse.rs:69-157 Display of token `ti` of an IrView-like object (tokens, val64, strings, source)"""
    t = view.tokens[ti]
    kind, aux = int(t["kind"]), int(t["aux"])
    if kind == T_INTEGER:
        value = view.token_u64(ti)
        if not aux & 0x8000:                            # no suffix
            return str(value)
        signed, size = bool(aux & 0x100), aux & 0xFF
        return f"{value}{'o' if signed else 's'}" if size == 0 else f"{value}{'i' if signed else 'n'}{size}"
    if kind == T_FLOAT:
        import numpy as np
        d = float(np.array([int(view.val64[int(t["payload"])])], dtype=np.uint64).view(np.float64)[0])
        return V.rust_f64(d) + (f"f{aux}" if aux else "")
    if kind == T_STRING:
        return view.token_string(ti).decode("utf-8", errors="replace")
    if kind == T_IDENTIFIER:
        low = int(t["low"])
        return view.source[low:low + int(t["len"])].decode("utf-8", errors="replace")
    return TOKEN_TEXT[kind]


def parse_error_message(err, view) -> str:
    """the line after "Error: " for a parse::Error (compile.rs:788-815); `err` is the opusfe_error record"""
    k = int(err.kind)
    if k == 0:
        return f"unexpected character: {chr(int(err.payload))}"
    if k == 1:
        return "unexpected end of file"
    if k == 2:
        return "invalid escape sequence"
    found = token_text(view, int(err.tok_index)) if k in (3, 4, 5) else None
    if k == 3:
        return f"unexpected token: {found}"
    if k == 4:
        return f"unexpected token: expected {TOKEN_TEXT[int(err.expected)]}, found {found}"
    if k == 5:
        return f"unexpected token: expected name, found {found}"
    return f"invalid numeric size: {int(err.payload)}"


class _LexView:
    """token-level arrays of a unit that stopped at a syntax error (no IR)"""

    def __init__(self, unit):
        self.tokens, self.val64 = unit.array(V.A_TOKENS), unit.array(V.A_VAL64)
        self.strings, self.source = unit.array(V.A_STRINGS).tobytes(), unit.array(V.A_SOURCE).tobytes()

    token_u64 = V.IrView.token_u64
    token_string = V.IrView.token_string


def render_parse_error(unit, filename: str) -> str:
    """what report_error(ParseError(e)) writes to stderr (compile.rs:786 + :788-815 + print_span) for a unit with status SYNTAX"""
    err = unit.error()
    view = _LexView(unit)
    return f"{RED}Error:{RESET} {parse_error_message(err, view)}\n" + print_span(filename, view.source, int(err.low), int(err.len))


def semantic_message(kind: int, span_text: str):
    m = SEMANTIC_TEXT.get(kind)
    if m is None:
        return None
    return m(span_text) if callable(m) else m


def _record_name(rec, view) -> str:
    """text of the record's `name`: an identifier, or Operator::symbol (parse.rs:224-243) of an operator token / builtin symbol"""
    name = int(rec["name"])
    if name & 0x80000000:
        kind = name & 0xFF
        return "=" if kind == 8 else TOKEN_TEXT[kind]
    return view.token_name(name)


def semantic_message_of_record(rec, view, arg_types=None):
    """message line of compile.rs:816-915 for one opusfe_semantic_error record; `view` = Unit.ir_view() (types, tokens, source);
    `arg_types` = view.sem_error_arg_types[i] for record i (every argument type of a function identifier: the record itself holds
    four, the rest come from its continuation records).  None when a function identifier has more than four argument types and
    `arg_types` was not given."""
    kind, low, length = int(rec["kind"]), int(rec["low"]), int(rec["len"])
    plain = semantic_message(kind, view.source[low:low + length].decode("utf-8", errors="replace"))
    if plain is not None:
        return plain
    ty = lambda k: view.type_text(int(rec["t"][k]))
    if kind == 2:                                                    # ShadowedName :820
        return f"shadowed variable: {_record_name(rec, view)}"
    if kind in (3, 4, 5):                                            # UndefinedFunction / NoOverloadForFunction / AmbiguousOverload :824-835
        n = int(rec["nargs"])
        if arg_types is None:
            if n > 4:
                return None
            arg_types = [int(rec["t"][k]) for k in range(n)]
        if len(arg_types) != n:
            return None
        ident = f"{_record_name(rec, view)}[" + " ".join(":" + view.type_text(t) for t in arg_types) + "]"      # FunctionIdentifierPrinter, compile.rs:302-319
        return ("undefined function: ", "no matching overload for function: ", "call to overloaded function is ambiguous: ")[kind - 3] + ident
    if kind == 7:                                                    # UnexpectedType :840
        return f"invalid type: expected {ty(0)}, found {ty(1)}"
    if kind == 8:                                                    # InvalidType :844
        return f"invalid type: {ty(0)}"
    if kind == 9:                                                    # InvalidOperandType :848
        return f"invalid operand type: {ty(0)}"
    if kind == 10:                                                   # InvalidOperandTypes :852
        return f"invalid operand types: {ty(0)} {_record_name(rec, view)} {ty(1)}"
    if kind == 17:                                                   # InvalidCast :880
        return f"invalid cast: {ty(0)} to {ty(1)}"
    if kind == 19:                                                   # FieldAccessOnNonRecord :888
        return f"attempt to set or access field of non-record: {ty(0)}"
    if kind == 20:                                                   # FieldDoesNotExist :892
        return f"type {ty(0)} has no field named '{_record_name(rec, view)}'"
    if kind == 21:                                                   # CallOfNonFunction :896
        return f"cannot call value of type {ty(0)}"
    return None


def render_unit_errors(unit, filename: str) -> str:
    """everything the reference writes to stderr for this unit's errors (the parse error, else every semantic diagnostic in report
    order): what the oracle's dump 6 restates independently (oracle_dump.cpp dump_rendered_errors)"""
    if unit.status == 1:
        return render_parse_error(unit, filename)
    view = unit.ir_view()
    return "".join(render_semantic_record(rec, view, filename, arg_types=view.sem_error_arg_types[i]) for i, rec in enumerate(view.sem_errors))


def render_semantic_record(rec, view, filename: str, source: bytes = None, low: int = None, arg_types=None) -> str:
    """stderr text of one diagnostic record: message + print_span.  `source` / `low` override the unit's own source and position
    (a multi-file program renders the span inside the file it belongs to)."""
    msg = semantic_message_of_record(rec, view, arg_types)
    if msg is None:
        msg = f"<{SEMANTIC_NAMES[int(rec['kind'])]}>"
    src = view.source if source is None else source
    return f"{RED}Error:{RESET} {msg}\n" + print_span(filename, src, int(rec["low"]) if low is None else low, int(rec["len"]))


def render_semantic(kind: int, filename: str, source: bytes, low: int, length: int) -> str:
    """stderr text of one semantic diagnostic; kinds whose message prints types / function identifiers fall back to
    "<KindName>" as the message (documented gap, see the module docstring)"""
    msg = semantic_message(kind, source[low:low + length].decode("utf-8", errors="replace"))
    if msg is None:
        msg = f"<{SEMANTIC_NAMES[kind]}>"
    return f"{RED}Error:{RESET} {msg}\n" + print_span(filename, source, low, length)
