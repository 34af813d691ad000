"""Semantic mutation of generated Opus sources — TEST INFRASTRUCTURE.

The generator (tools/) only writes well-typed programs, so the error paths of IR generation (the 25 semantic error kinds,
`Error` instructions, the constructs on which the reference panics) are reached only by hand cases.  `mutants()` perturbs
generated programs at token level (identifiers, types, literals, operators, lines) so that most stay syntactically valid but
ill-typed; the oracle decides what each one must produce and the device code (emulated on the host, or on the GPU) has to
agree bit for bit."""
import random
import re

import oracle_lib as O

KEYWORDS = {"var", "const", "return", "if", "else", "while", "for", "break", "continue", "extern", "ref", "mut", "refs", "muts",
            "null", "false", "true", "type", "record", "is", "in", "and", "or", "not", "sizeof", "alignof", "import", "library",
            "proc", "operator"}
TYPES = ["int8", "int16", "int32", "int64", "nat8", "nat16", "nat32", "nat64", "bool", "float32", "float64", "string", "size",
         "offset"]
TYPE_SUBST = TYPES + ["@int32", "@mut int64", "{} nat8", "mut {} int32", "nosuch", "(int32) int32", "{3} int32", "null",
                      "@@bool", "record { x int32 }"]
LITERAL_SUBST = ["1.5", '"s"', "null", "true", "false", "0", "300", "5i8", "7n64", "2.0f32", "(1)", "-1", "sizeof int32"]
OPERATORS = ["+", "-", "*", "/", "%", "==", "<", ">", "<=", ">=", "and", "or", "is"]
PREFIXES = ["ref ", "mut ", "refs ", "muts ", "-", "not ", "sizeof ", "proc "]
SUFFIXES = ["@", ".x", ".length", ".data", "(0)", "()", ":int64", ":@int32", ":bool", "(1, 2)", ":float64", ":{} nat8"]
LINES = ["break", "continue", "return", "return null", "return 1", "var zz = zz", "var zz int32 = 1.5", "zz = 1", "1 = 2",
         "while 1\n{i}\tbreak", "if null\n{i}\treturn", "for q in 5\n{i}\tbreak", "var zz = ref 5", "var zz = (ref zz)@",
         "var zz {2} int32\n{i}zz(0, 1) = 2", "var zz = record nosuch { a = 1 }", "var zz = proc nosuch(int32)"]

IDENT = re.compile(rb"[a-z_][a-z0-9_]*")
INTEGER = re.compile(rb"(?<![a-z0-9_.])[0-9]+(?![a-z0-9_.])")
OPERATOR = re.compile(rb" (\+|-|\*|/|%|==|<=|>=|<|>|and|or) ")


def _occurrences(src, rx, keep=lambda m: True):
    return [m for m in rx.finditer(src) if keep(m)]


def mutate(src: bytes, rnd: random.Random) -> bytes:
    for _ in range(rnd.randint(1, 4)):
        kind = rnd.randrange(10)
        if kind == 0:      # an identifier becomes another identifier of the file (wrong type, wrong arity, undefined ...)
            ids = _occurrences(src, IDENT, lambda m: m.group().decode() not in KEYWORDS and m.group().decode() not in TYPES)
            if len(ids) >= 2:
                a, b = rnd.choice(ids), rnd.choice(ids)
                src = src[:a.start()] + b.group() + src[a.end():]
        elif kind == 1:    # a type becomes another type
            tys = _occurrences(src, IDENT, lambda m: m.group().decode() in TYPES)
            if tys:
                a = rnd.choice(tys)
                src = src[:a.start()] + rnd.choice(TYPE_SUBST).encode() + src[a.end():]
        elif kind == 2:    # an integer literal becomes something else
            ints = _occurrences(src, INTEGER)
            if ints:
                a = rnd.choice(ints)
                src = src[:a.start()] + rnd.choice(LITERAL_SUBST).encode() + src[a.end():]
        elif kind == 3:    # an operator becomes another
            ops = _occurrences(src, OPERATOR)
            if ops:
                a = rnd.choice(ops)
                src = src[:a.start(1)] + rnd.choice(OPERATORS).encode() + src[a.end(1):]
        elif kind in (4, 5):   # a line is deleted / duplicated
            lines = src.split(b"\n")
            i = rnd.randrange(len(lines))
            lines[i:i + 1] = [] if kind == 4 else [lines[i], lines[i]]
            src = b"\n".join(lines)
        elif kind == 6:    # a statement is inserted at the indentation of its neighbour
            lines = src.split(b"\n")
            body = [i for i, ln in enumerate(lines) if ln.startswith(b"\t")]
            if body:
                i = rnd.choice(body)
                indent = lines[i][:len(lines[i]) - len(lines[i].lstrip(b"\t"))].decode()
                lines.insert(i, (indent + rnd.choice(LINES).replace("{i}", indent)).encode())
                src = b"\n".join(lines)
        elif kind == 7:    # a prefix operator in front of an identifier inside a body
            ids = _occurrences(src, IDENT, lambda m: m.group().decode() not in KEYWORDS and m.start() > 0 and src[m.start() - 1:m.start()] in (b" ", b"("))
            if ids:
                a = rnd.choice(ids)
                src = src[:a.start()] + rnd.choice(PREFIXES).encode() + src[a.start():]
        elif kind == 8:    # a postfix operator behind an identifier
            ids = _occurrences(src, IDENT, lambda m: m.group().decode() not in KEYWORDS and m.group().decode() not in TYPES)
            if ids:
                a = rnd.choice(ids)
                src = src[:a.end()] + rnd.choice(SUFFIXES).encode() + src[a.end():]
        else:              # `return <expr>` loses its value
            rets = [m for m in re.finditer(rb"return [^\n]+", src)]
            if rets:
                a = rnd.choice(rets)
                src = src[:a.start()] + b"return" + src[a.end():]
    return src


def mutants(n, seed0=0, lo=600, hi=5000):
    """n mutated sources (deterministic)"""
    out = []
    rnd = random.Random(1234 + seed0)
    for k in range(n):
        base = O.generate(40_000 + seed0 + k, rnd.randint(lo, hi))
        out.append(mutate(base, rnd))
    return out


BYTES = b"\t\t\t\n\n\n  ()[]{}:=<>+-*/%@~.,?\"\"\\\\;;#$!&|^`'0123456789abcdefxyz_ABZ\r\x00\x7f"


# lexical errors behind which the reference resumes in the middle of a string body or swallows arbitrary characters
# (bad escapes, invalid float suffixes): dropped by the parser at some positions (Q6/Q7), the resync passes of api.hip
LEXICAL_TRAPS = [b'"ab\\q', b'\\q', b'"\\xZ1', b'1.5f";', b'"\\', b'"', b'\\x4', b'"\\xG', b'2.5f(7', b'"\\"', b'\\\\', b'3.5f\n',
                 b'"a\\q"', b'"\\q\\w\\e', b'1.0f ;',
                 # non-ASCII in code: Unicode white space is skipped, any other character is an UnexpectedCharacter error of
                 # its own length; right behind an identifier it is UNSUPPORTED (is_alphanumeric tables unpinned)
                 "\u00a0".encode(), "\u3000 ".encode(), "\u2028".encode(), " \u2192 ".encode(), "\u00e9".encode(), " \U0001F600".encode(),
                 "\u00a0\u00a0x".encode(), "\u0085".encode(),
                 # invalid float suffixes that swallow non-ASCII characters (code-point arithmetic, parse.rs:572,577)
                 "1.5f\u00e97".encode(), "2.5f\u2192".encode(), "1.5f\u00a0 ".encode(), "3.5f(\U0001F600".encode(), "4.5f\u2192\u2192x".encode()]


def trap_mutants(n, seed0=0, lo=700, hi=2500):
    """n sources with 1-4 LEXICAL_TRAPS inserted (half of them at line starts, where the parser drops an Err token)"""
    out = []
    rnd = random.Random(4242 + seed0)
    for k in range(n):
        s = bytearray(O.generate(70_000 + seed0 + k, rnd.randint(lo, hi)))
        for _ in range(rnd.randint(1, 4)):
            i = rnd.randrange(len(s))
            if rnd.random() < 0.5:
                j = s.find(b"\n", i)
                i = j + 1 if j >= 0 else i
            s[i:i] = rnd.choice(LEXICAL_TRAPS)
        out.append(bytes(s))
    return out


def byte_mutants(n, seed0=0, lo=700, hi=3000):
    """n sources with a few random byte edits (replace / insert / delete / swap of lines' indentation): lexer contexts,
    brackets, dents, continuation lines, escapes"""
    out = []
    rnd = random.Random(777 + seed0)
    for k in range(n):
        s = bytearray(O.generate(60_000 + seed0 + k, rnd.randint(lo, hi)))
        for _ in range(rnd.randint(1, 5)):
            if not s:
                break
            i = rnd.randrange(len(s))
            op = rnd.randrange(4)
            if op == 0:
                s[i] = rnd.choice(BYTES)
            elif op == 1:
                s.insert(i, rnd.choice(BYTES))
            elif op == 2:
                del s[i]
            else:
                j = s.find(b"\n", i)
                if j >= 0:
                    s[j + 1:j + 1] = b"\t" * rnd.randint(1, 3) if rnd.random() < 0.7 else b"  "
        out.append(bytes(s))
    return out


def swallowed_lexical_error(src: bytes) -> bool:
    """the reference lexed an Err token and went on (Q6/Q7: after a bad escape it resumes inside the string body as code):
    the product's documented OPUSFE_UNSUPPORTED, the only accepted difference"""
    lines = O.dump(src, O.TOKENS).splitlines()
    return any(" Err(" in ln for ln in lines[:-1])


def expectation(src: bytes):
    """what the reference does with `src`, per the oracle: ("syntax", error line, unsupported_allowed) | ("unsupported",) |
    ("semantic", diagnostics) — finalize errors, no IR | ("ir", IR text, diagnostics)"""
    err = O.dump(src, O.ERROR)
    if err == "Unsupported\n":
        return ("unsupported",)
    if err != "Ok\n":
        # swallowed lexical errors (Q6/Q7) are followed exactly since the resync passes (api.hip body_parse): no allowance
        return ("syntax", err, False)
    if O.time_run(src)[1].status == 2:
        return ("unsupported",)
    ir, sem = O.dump(src, O.IR), O.dump(src, O.SEM)
    if ir == "" and sem != "":
        return ("semantic", sem)
    return ("ir", ir, sem)
