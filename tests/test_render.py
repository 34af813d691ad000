"""Error rendering (opus_amd/render.py; frontend.rs:201-256 print_span, compile.rs:782-917 print_error).
PARITY UNPINNED: the reference holds no expected outputs; every expected string below was derived by hand from the cited lines.
The print_span cases run on the CPU (pure formatting); the parse-error cases need the GPU (the error tuple and the token arrays
come from the HIP path)."""
import pytest

from opus_amd import render as R

Y, RED, Z = "\x1b[33m", "\x1b[31;1m", "\x1b[0m"


def test_print_span_single_line_with_tab():
    src = b"f() int32\n\treturn x\n"
    assert R.print_span("a.opus", src, 18, 1) == f"{Y}a.opus:{Z}\n{Y}   2:{Z}     return {RED}x{Z}\n"


def test_print_span_ends_mid_line_and_spans_lines():
    assert R.print_span("f", b"hello world\n", 0, 5) == f"{Y}f:{Z}\n{Y}   1:{Z} {RED}hello{Z} world{Z}\n"
    # a span over a line break: the line number column is re-printed and the colour re-opened (frontend.rs:244-250)
    assert R.print_span("f", b"ab\ncd\nef\n", 1, 4) == f"{Y}f:{Z}\n{Y}   1:{Z} a{RED}b\n{Y}   2:{Z} {RED}cd{Z}\n"


def test_print_span_empty_span_at_end_of_file_and_utf8():
    src = b'f() null\n\tvar s = "ab'
    assert R.print_span("f", src, len(src), 0) == f'{Y}f:{Z}\n{Y}   2:{Z}     var s = "ab{Z}\n'
    # byte offsets, not character offsets: the span starts after a 2-byte character
    src = "; é\nxy z\n".encode()
    assert R.print_span("f", src, 8, 1) == f"{Y}f:{Z}\n{Y}   2:{Z} xy {RED}z{Z}\n"
    src = "é = q\n".encode()
    assert R.print_span("f", src, 5, 1) == f"{Y}f:{Z}\n{Y}   1:{Z} é = {RED}q{Z}\n"


def test_semantic_messages_without_payload():
    assert R.semantic_message(1, "nosuch") == "undefined variable: nosuch"
    assert R.semantic_message(18, "T") == "undefined type: T"
    assert R.semantic_message(22, "") == "refernce to built-in function"      # the reference's spelling, compile.rs:901
    assert R.semantic_message(7, "x") is None                                   # UnexpectedType prints types: needs the record (semantic_message_of_record)
    src = b"f() int32\n\treturn nosuch\n"
    assert R.render_semantic(1, "m.opus", src, 18, 6) == (f"{RED}Error:{Z} undefined variable: nosuch\n{Y}m.opus:{Z}\n"
                                                          f"{Y}   2:{Z}     return {RED}nosuch{Z}\n")


PARSE_ERRORS = [
    (b"f() int32\n\treturn )\n", f"unexpected token: )\n{Y}t.opus:{Z}\n{Y}   2:{Z}     return {RED}){Z}\n"),
    (b"f() int32\n\treturn 1 # 2\n", f"unexpected character: #\n{Y}t.opus:{Z}\n{Y}   2:{Z}     return 1 {RED}#{Z} 2{Z}\n"),
    (b"var 5 int32\n", f"unexpected token: expected name, found 5\n{Y}t.opus:{Z}\n{Y}   1:{Z} var {RED}5{Z} int32{Z}\n"),
    (b"f() int32\n\treturn (1 200n8\n", f"unexpected token: expected ), found 200n8\n{Y}t.opus:{Z}\n{Y}   2:{Z}     return (1 {RED}200n8{Z}\n"),
    (b'f() null\n\tvar s = "ab', f'unexpected end of file\n{Y}t.opus:{Z}\n{Y}   2:{Z}     var s = "ab{Z}\n'),
    (b"f() int32\n\treturn 5i7\n", f"invalid numeric size: 7\n{Y}t.opus:{Z}\n{Y}   2:{Z}     return 5i{RED}7{Z}\n"),
    (b'f() null\n\tvar s = "ab\\q"\n', f'invalid escape sequence\n{Y}t.opus:{Z}\n{Y}   2:{Z}     var s = "ab\\{RED}q{Z}"{Z}\n'),
    (b"f( int32\n", f"unexpected token: <end of file>\n{Y}t.opus:{Z}\n{Y}   2:{Z} {Z}\n"),
]


@pytest.mark.gpu
@pytest.mark.parametrize("src,want", PARSE_ERRORS, ids=[f"perr{i}" for i in range(len(PARSE_ERRORS))])
def test_parse_error_rendering(fe, src, want):
    u = fe.get_definitions(src)
    assert u.status == 1
    assert R.render_parse_error(u, "t.opus") == f"{RED}Error:{Z} " + want
