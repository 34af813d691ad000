"""Shared hand-written inputs for the parity tests (GPU: tests/test_*_gpu.py; host emulation: tests/test_emu_cpu.py)."""

LEX_CASES = ["5s", "5i8", "5n", "5i80", "5in", "7i12", "1.5f32", "2.", "1..2", "a\\b", '"a\\x41\\n"', "x<=y", "; c\n\tfoo",
             "123abc", "", " ", "\n", "\t", "\tfoo", " \tfoo", "\t; c\n\tfoo", "a\n\tb\n\t\tc\nd", "a\n\tb\n\t\tc", "==== <== >== =<=",
             "5.foo", "5i7", "5i", "12n16x", "0.5f64+1", "x\n\n\n\ty\n ; c\n\tz\n", '"unterminated', '"bad \\q esc" x', '"\\x4"',
             "a#b", "#", "foo(\n\tbar,\n\tbaz)\n", "a;b\nc", '"a;b" ; "c"\nd', "a\r\n\tb\r\n", "\x0b\x0cfoo",
             "18446744073709551616 4294967296 4294967295", "1.25 0.1 123456.789 3.0f32",
             "x" * 200, "ab " * 100, '"' + "s" * 300 + '" tail', "; " + "c" * 300 + "\nafter", "1" * 70,
             '"' + "u" * 500, 'x = "abc\\', '"\\x', '"\\xZZ" y', "a" * 63 + '"' + "b" * 200, "\t" * 70 + "x\n" + "\t" * 3 + "y"]

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
    "f() null\n\tx = 1 +\n", "f() null\n\tx = 1 + )\n", "f() null\n\tx = \"bad \\q esc\"\n", "f() null\n\tif a\n\t\tb\n\t\"bad \\q esc\" c\n\td\n",
]
