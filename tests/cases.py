"""Shared hand-written inputs for the parity tests (GPU: tests/test_*_gpu.py; host emulation: tests/test_emu_cpu.py)."""

LEX_CASES = ["5s", "5i8", "5n", "5i80", "5in", "7i12", "1.5f32", "2.", "1..2", "a\\b", '"a\\x41\\n"', "x<=y", "; c\n\tfoo",
             "123abc", "", " ", "\n", "\t", "\tfoo", " \tfoo", "\t; c\n\tfoo", "a\n\tb\n\t\tc\nd", "a\n\tb\n\t\tc", "==== <== >== =<=",
             "5.foo", "5i7", "5i", "12n16x", "0.5f64+1", "x\n\n\n\ty\n ; c\n\tz\n", '"unterminated', '"bad \\q esc" x', '"\\x4"',
             "a#b", "#", "foo(\n\tbar,\n\tbaz)\n", "a;b\nc", '"a;b" ; "c"\nd', "a\r\n\tb\r\n", "\x0b\x0cfoo",
             "18446744073709551616 4294967296 4294967295", "1.25 0.1 123456.789 3.0f32",
             "x" * 200, "ab " * 100, '"' + "s" * 300 + '" tail', "; " + "c" * 300 + "\nafter", "1" * 70,
             '"' + "u" * 500, 'x = "abc\\', '"\\x', '"\\xZZ" y', "a" * 63 + '"' + "b" * 200, "\t" * 70 + "x\n" + "\t" * 3 + "y",
             # float literals outside the Clinger fast path: correctly rounded like str::parse::<f64> (ties to even)
             "0.30000000000000004 9007199254740993.0 9007199254740992.5 4503599627370497.5 123456789012345678901234567890.5",
             "0.000000000000000000000000000001 1.0000000000000000000000000000000000000001 3.14159265358979323846264338327950288",
             "1.50000000000000000000000000 000.000 179769313486231570000000000000000000000.0 0.1000000000000000055511151231257827"]
# more than 300 significant digits: OPUSFE_UNSUPPORTED, never a wrong value
TOO_LONG_FLOAT = "var x = 1." + "1" * 301 + "\n"

# parser inputs: valid programs first, then one construct per syntax error
PARSE_CASES = [
    "", "\n", "; only a comment\n", "var x int32\n", "var x int32 = 5\nconst y nat8 = 7\n", "type T = record { a int32 b @mut T }\n",
    "type P = (int32, @bool) null\ntype A = {4} int32\ntype S = {} nat8\ntype M = {} mut nat8\n",
    "extern puts(s {} nat8) int32\nimport \"a/b\"\nlibrary \"c\"\n",
    "f() null\n\tx = 1\n", "f() int32 return 1\n", "f() null if a b\n", "f() null\n\tif a\n\t\tb\n\telse\n\t\tc\n",
    "f() null\n\twhile true\n\t\tbreak\n\telse\n\t\tcontinue\n",
    "f() null\n\tfor i in r\n\t\tx\n\tfor j int32 in r\n\t\ty\n\telse\n\t\tz\n",
    "f() null\n\tvar a = 1\n\tvar b int32\n\tvar c int32 = 2\n\t(var d int32) = 3\n",
    "f() null\n\ta = b\n\t- c\n", "f() null\n\tx = a == not b\n\ty = not a or b and c\n\tz = - not q\n",
    "f() null\n\tx = 1 + 2 * 3 - 4 / 5 % 6 < 7 and 8 >= 9 or p is q\n",
    "f() null\n\tx = a * b + c * d == e\n\ty = (a + b) * c\n",
    "f() null\n\tx = a.b.c@.d(1, 2)(3):int32@\n\ty = f()\n\tz = (f)(x,)\n\tw = g(\n\t\t1,\n\t\t2\n\t)\n",
    "f() null\n\tx = record P { a = 1 b = q(2) }\n\ty = sizeof @int32 + alignof {2} nat8\n\tz = proc f(int32, bool)\n",
    "f() null\n\tx = ref y\n\tx = mut y\n\tx = refs y\n\tx = muts y\n\tx = -y\n\tx = --y@\n",
    "operator +(a P, b P) P\n\treturn a\noperator ==(a P, b P) bool return true\noperator not(a P) bool return false\n"
    "operator for(a P) P return a\noperator continue(a @mut P, x @mut int32) bool return false\noperator <=(a P, b P) bool return true\n",
    "f(a int32, b @mut {} nat8,) int32\n\treturn a\n",
    "f() null\n\tx = \"str\\n\" + 1.5f32 + 2. + 7i64 + 5s\n",
    # errors
    "f() int32\n\treturn #\n", "#\nf() int32\n\treturn 1\n", "f() int32\n\tif true\n\t\treturn 1\n\t#\n\treturn 2\n",
    "f( int32\n", "f() null\n\tx = (1 + 2\n", "f() null\n\tx = g(1 2)\n", "f() null\n\tx = 5i7\n", "f() null\n\t\tx = 1\n",
    "f() null\n\tvar if = 1\n", "f() null\n\tx = \"abc\n", "operator and(a P) P\n", "operator var(a P) P\n", "type X int32\n",
    "const c int32 5\n", "import x\n", "f() null\n\treturn\n", "f() null\n\tx = [1]\n", "f() null\n\tx = record P { 1 }\n",
    "f() null\n\tx = record P { a 1 }\n", "type T = record { a }\n", "type T = {x} int32\n", "type T = {3 int32\n", "var\n", "f",
    "f() null\n\tx = 1\n\t\ty = 2\n", "f() null\nx\n", "else\n", "f() null\n\tx\nelse\n\ty\n", "f() null if a\n\tb\nelse\n\tc\ng() null\n\td\n",
    "f() null\n\tx = 1\n#\n", "f() null\n\tx = 1\n#", "f() null\n\tx = 1\n# #\ng() null\n\ty\n", "f() null\n\tx = 1\n#\ng() null\n\ty\n",
    "f() null\n\tfor in x\n\t\ty\n", "f() null\n\tfor i x\n\t\ty\n", "f() null\n\tx = a.1\n", "f() null\n\tx = a:5\n",
    "f() null\n\tx = proc 5()\n", "f() null\n\tx = proc f[int32]\n", "f() null\n\tx = (a\n\t\t+ b)\n\ty = z\n",
    "f() null\n\tx = g(\na,\nb)\n\ty = 1\n", "var x @ @mut @ {} {} mut {3} (int32) null\n",
    "f() null\n\tx = 1\n\textern g() null\n", "var a int32 extern g() null\ntype T = int32 const c int32 = 1\n", "f() null x\nextern g() null\n",
    "f() null\n\tx = (1 +\nextern g() null\n", "f() null\n\tif a\n\t\tb\n\ttype T = int32\n", "var a int32 =\ntype T = int32\n", "import \"a\" library \"b\" extern f() null\n",
    "f() null\n\tx = 1 +\n", "f() null\n\tx = 1 + )\n", "f() null\n\tx = \"bad \\q esc\"\n", "f() null\n\tif a\n\t\tb\n\t\"bad \\q esc\" c\n\td\n",
]

# IR generation inputs (all parse): well-typed programs exercising every lowering path, then ill-typed ones whose IR
# contains Error instructions and whose semantic diagnostics are compared too.
IR_CASES = [
    "main() int32\n\treturn 1 + 2 * 3\n",
    "f() null\n\tvar s = \"hi\\n\"\n\tvar d = s.data\n\tvar l = s.length\n",
    "var g int32 = 5\nconst c int64 = -7\nf() int64\n\tg = g + 1\n\treturn c\n",
    "f(p @mut int32, q @int32) int32\n\tp@ = q@ + 1\n\treturn p@\n",
    "f(a {} nat8, b mut {} nat8) nat8\n\tb(1) = a(0)\n\treturn b(2)\n",
    "f() int32\n\tvar arr {4} int32\n\tarr(0) = 1\n\tvar p = ref arr\n\tvar q {} int32 = p\n\treturn arr(1) + q(0)\n",
    "g(x int32) int32 return x\nf() int32\n\tvar h = proc g(int32)\n\tvar k = g\n\treturn h(1) + k(2) + g(3)\n",
    "type P = record { x int32 y float64 }\nf() float64\n\tvar p = record P { x = 1 y = 2.5 }\n\tp.x = 3\n\tvar q = mut p\n\tq@.y = 1.0\n\treturn p.y\n",
    "f(a int32, b int64) bool\n\treturn a:int64 < b and not (a == 3) or a >= 2\n",
    "f(a @int32, b @int32) bool return a is b\n",
    "f(a int8) int8 return -a\nh() int32 return -5\nk() int64\n\tvar z int64 = -9\n\treturn z\n",
    "f() size return sizeof int32 + alignof @int64\n",
    "f(a int32) int32\n\twhile true\n\t\tif a > 3\n\t\t\tbreak\n\t\ta = a + 1\n\t\tcontinue\n\treturn a\n",
    "f(a int32) int32\n\twhile a < 3\n\t\ta = a + 1\n\telse\n\t\treturn 0\n\treturn a\n",
    "f(a bool) null\n\tif a\n\t\treturn null\n",
    "f(a bool) int32\n\tif a\n\t\treturn 1\n\telse\n\t\treturn 2\n",
    "f(a bool) int32\n\twhile true\n\t\ta = a\n",
    "type R = record { lo int32 hi int32 }\ntype I = record { cur int32 hi int32 }\noperator for(r R) I\n\treturn record I { cur = r.lo hi = r.hi }\n"
    "operator continue(it @mut I, x @mut int32) bool\n\tif it@.cur < it@.hi\n\t\tx@ = it@.cur\n\t\tit@.cur = it@.cur + 1\n\t\treturn true\n\treturn false\n"
    "f(r R) int32\n\tvar s = 0\n\tfor i in r\n\t\ts = s + i\n\tfor j int32 in r\n\t\tif j > 2\n\t\t\tbreak\n\telse\n\t\ts = 0\n\treturn s\n",
    "type V = record { x int32 }\noperator +(a V, b V) V return a\noperator ==(a V, b V) bool return true\noperator not(a V) bool return false\n"
    "f(a V, b V) bool\n\tvar c = a + b\n\treturn c == a and not b\n",
    "f(a int32) int32\n\tvar a = 3\n\treturn a\n",
    "f() int32\n\tvar x nat8 = 200\n\tvar y = x + 5\n\tvar z int64 = 1 + 2\n\treturn 1\n",
    "f(p @mut @mut int32) int32 return p@@\n",
    "type T = @@@@@@@@int32\nf(t T) int32 return t@@@@@@@@\n",
    "f() int32\n\tvar f2 (int32) int32\n\treturn f2(1)\n",
    "extern puts(s {} nat8) int32\nf() int32\n\treturn puts(\"x\".data)\n",
    "f(x int32) int32\n\treturn ((x)) + (1)\n",
    "f() float32 return 1.5f32\ng() float64 return 2.5 * 2.0\n",
    "f(a int32) int32\n\tvar b = a\n\tif a > 0\n\t\tvar c = b\n\t\tb = c\n\tvar c = 1\n\treturn b + c\n",
    "f(a nat8, b nat8) nat8 return a % b / 2n8\n",
    "type A = B\ntype B = int32\nf(a A, b B) int32 return b\n",
    "f(s string) size return s.length\n",
    "f(p mut {} int32) @mut int32\n\treturn mut p(3)\n",
    "f(r @mut {3} int32) int32\n\treturn r@(1)\n",
    "f() int32\n\tvar i = 0\n\twhile i < 10 and true\n\t\ti = i + 1\n\t\tbreak\n\treturn i\n",
    # ill-typed / erroneous programs (semantic errors; IR still generated)
    "f() int32\n\treturn x\n", "f() int32\n\tvar a = 1\n\tvar a = 2\n\treturn a\n", "f() int32\n\treturn g(1)\n",
    "g(a bool) int32 return 1\nf() int32\n\treturn g(1)\n", "f() int32\n\treturn true\n", "f() int32\n\tvar x = 1\n", "f() null\n\tbreak\n\tcontinue\n",
    "f() null\n\t1 = 2\n", "f(a @int32) null\n\ta@ = 1\n", "f(a int32) null\n\tvar b = a:bool\n", "f(a int32) null\n\tvar b = a.x\n",
    "type P = record { x int32 }\nf(p P) null\n\tvar b = p.y\n", "f(a int32) null\n\tvar b = a(1)\n", "f() null\n\tvar b = proc +(int32, int32)\n",
    "const c int32 = 1\nf() null\n\tvar b = c(1)\n", "f(a {} int32) null\n\tvar b = a(1, 2)\n", "f(a float64) float64 return -a\n",
    "f(a int32, b bool) null\n\tvar c = a and b\n\tvar d = b and a\n", "f(a int32, b bool) null\n\tvar c = a is b\n", "f() null\n\tvar a Undefined\n",
    "g(a int32) int32 return a\ng(a int64) int32 return 1\nf() null\n\tvar a = g\n\tvar b = g(1)\n", "f() null\n\tvar a = ref 1\n",
    "f(a int32) null\n\tvar b = refs a\n",
    # finalize errors: no IR
    "f(a Undefined) null\n\treturn null\n", "var g int32 = 1 + 2\n", "var g bool = 1\n", "type A = Undefined\n",
]


def big_function(n_locals=150, n_ifs=140, chain=90, depth=70):
    """One function that overflows every LDS-resident table of the lowering machine (scope table 64, variable types 320,
    basic blocks 96, own types 64, frame stack 32) and of the parser (frame stack 48, node ring 256): the spill paths to the
    sparse global regions must give the same IR as the oracle."""
    L = ["big(a int32, b int64) int32"]
    for i in range(n_locals):
        L.append(f"\tvar v{i} = a + {i}")
    for i in range(n_ifs):
        L.append(f"\tif v{i % n_locals} < v{(i * 7 + 3) % n_locals}")
        L.append(f"\t\tv{(i * 5) % n_locals} = v{(i * 11) % n_locals} * {i + 1} - a")
        L.append("\telse")
        L.append(f"\t\tv{(i * 3) % n_locals} = v{i % n_locals} % {i + 2}")
    L.append("\tvar s = " + " + ".join(f"v{i % n_locals} * {i}" for i in range(chain)))
    L.append("\tvar d = " + "(" * depth + "a" + " + 1)" * depth)
    L.append("\twhile s < d")
    for i in range(40):
        L.append(f"\t\tvar w{i} = mut v{i}")
        L.append(f"\t\tw{i}@ = w{i}@ + s")
    L.append("\t\ts = s + 1")
    L.append("\treturn s + d")
    return "\n".join(L) + "\n"


def many_overloads(n=40):
    """one operator symbol and one function name with n definitions each: the name-table slot is ordered by the heapsort path
    of k_sem_table_sort (> 16 entries) and lookup_function walks all of them in definition order (compile.rs:1083-1139)"""
    L = []
    for i in range(n):
        L.append(f"type R{i} = record {{ a{i} int32 }}")
    for i in reversed(range(n)):
        L.append(f"operator +(x R{i}, y R{i}) R{i}\n\treturn x")
        L.append(f"pick(x R{i}) int32\n\treturn {i}")
    L.append(f"f(p R7, q R7, r R{n - 1}) int32\n\tvar s = p + q\n\tvar t = r + r\n\treturn pick(s) + pick(t) + pick(r)")
    return "\n".join(L) + "\n"


IR_CASES.append(many_overloads())
# pointer identity (generate_ir.rs:616-634) inside conditions and under `not`: lowered by the node-parallel path
IS_CASE = ("same(a @int32, b @int32, c @mut int32) int32\n\tvar n = 0\n\tif a is b\n\t\tn = n + 1\n\twhile not (a is b) and n < 3\n\t\tn = n + 1\n"
           "\tvar q = mut n\n\tif q is c\n\t\treturn 1\n\treturn n\n")
IR_CASES.append(IS_CASE)
# `proc` expressions (generate_ir.rs:586-602) and calls through variables of function type (local, argument, global; :948-960:
# arguments without expected type or autocast) — both lowered by the node-parallel path
PROC_CASE = ("extern g(a int32) int32\nextern g(a nat8) int32\npr(x int32) int32\n\tvar f = proc g(int32)\n\tvar h = proc pr(int32)\n\tvar k = proc g(nat8)\n"
             "\tif x > 2\n\t\treturn x\n\treturn 0\n")
FNPTR_CASE = ("extern g(a int32, b nat8) int32\nextern z() null\nvar gp (int32, nat8) int32\ncall(p (int32, nat8) int32) int32\n\tvar fp = g\n\tvar zz = z\n\tzz()\n"
              "\tvar x = fp(2, 3n8)\n\tif x > 0\n\t\treturn p(x, 4n8) + gp(1, 5n8)\n\treturn fp(x + 1, 6n8)\n")
IR_CASES += [PROC_CASE, FNPTR_CASE]

# finalize_definition errors (compile.rs:1002-1065): reported in definition order, no IR is generated (frontend.rs:145)
FINALIZE_ERROR_CASES = [
    "f(a nosuch, b int32, c other) missing\n\treturn 1\n",
    "var g nosuch\nf() int32\n\treturn 1\n",
    "var g int32 = 1 + 2\n",
    "var g float64 = 1\nconst c bool = 2\nconst d int8 = -3\n",
    "var g nosuch = 1.5\n",
    "var g nosuch = 3\n",
    "type T = record { a nosuch b @other c int32 }\nf(t T) int32 return 1\n",
    "extern e(p (nosuch, int32) other, q {4} third) fourth\n",
    "type A = B\ntype B = int32\nvar x A = 1\nvar y B = 2\n",
    "var a int32 = x\nf(p nosuch) int32\n\treturn q\nvar b other = -1\ntype U = @@missing\nconst k int64 = \"s\"\n",
    "f() int32\n\tvar x nosuch\n\treturn 1\nvar g nosuch2\n",
    "type P = record { x int32 }\nvar p P = 1\nvar q @int32 = 0\nvar r string = 1\n",
]

# duplicate `type` names: every definition of a name writes the one placeholder type_resolution_map holds for it
# (compile.rs:968-981, 1055-1062), in definition order; what a global between two of them sees is the earlier one's content
DUPLICATE_TYPE_CASES = [
    "type T = int32\nvar x T = 1\ntype T = float64\nvar y T = 2\n",
    "type T = int32\nf(a T) T\n\treturn a\ntype T = float64\ng(a T) T\n\treturn a\n",
    "type T = int32\ntype U = T\ntype T = bool\ntype V = T\nvar a U = 1\nvar b V = 1\nvar c T = 1\n",
    "var w T = 1\ntype T = nat8\nvar x T = 1\ntype T = T\nvar y T = 1\ntype T = @T\nvar z T = 1\nf(t T) T return t\n",
    "type A = B\ntype B = int32\ntype A = B\nvar p A = 1\ntype B = bool\ntype A = B\nvar q A = 1\n",
    "type A = B\ntype B = int32\ntype A = B\nf(p A, q B) A\n\tvar r A = q\n\treturn r\n",
]
IR_CASES += DUPLICATE_TYPE_CASES

# non-ASCII in code (parse.rs:432,474 char::is_whitespace; :718 UnexpectedCharacter), placed so that the characters straddle
# the 64-byte chunks of the GPU lexer; at top level the error token is dropped (Q7) and parsing goes on
NON_ASCII_CASES = ([" " * k + ch + "\nf() null\n\treturn\n" for k in range(57, 67) for ch in ("\u2192", "\u00a0", "\U0001F600", "\u3000\u00e9")]
                   + ["f() int32\n\treturn 1\u00a0+\u20282\n", "f() int32\n\treturn x\u00a0\n", "f() int32\n\treturn x\u00e9\n",
                      "f() int32\n\treturn 1 \u2192 2\n", "f() int32\n\tvar in\u00e9t = 1\n", "; \u00e9 \u2192\nf() null\n\tvar s = \"\u00e9\u2192\"\n"])
PARSE_CASES += NON_ASCII_CASES

# invalid float suffixes over multi-byte characters, straddling the lexer's 64-byte chunks; dropped at top level (Q7)
NON_ASCII_SUFFIX_CASES = [" " * k + lit + "\nf() null\n\treturn\n" for k in range(52, 64) for lit in ("1.5f\u00e9\u2192", "2.5f\U0001F600\U0001F600", "3.5f(\u2192", "4.5f\u00e97x")] \
                         + ["f() null\n\tx = 1.5f\u00e97\n", "f() null\n\treturn\n2.5f\u2192"]
PARSE_CASES += NON_ASCII_SUFFIX_CASES
