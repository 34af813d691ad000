"""CPU: the host logic of the import closure (opus_amd/program.py; frontend.rs:57-113) — queue order, canonical-path
de-duplication, `library` collection, offsets of the joined sources — with a stand-in for the GPU context (the stand-in only
fabricates the definition-root arrays the loop reads; lexing / parsing / joining proper are covered by tests/test_program_gpu.py)."""
import os
import re

import numpy as np
import pytest

from opus_amd import ir_view as V
from opus_amd import program as P


class FakeUnit:
    def __init__(self, src, status=0):
        self.src, self.status, self.closed = src, status, False
        self.dirs = [(5 if m.group(1) == b"import" else 6, m.group(2)) for m in re.finditer(rb'^(import|library) "([^"]*)"', src, re.M)]

    def array(self, which):
        n = len(self.dirs)
        if which == V.A_DEFINITIONS:
            return np.arange(n, dtype=np.uint32)
        if which == V.A_NODES:
            a = np.zeros(n, dtype=V.DTYPES[V.A_NODES])
            for i, (k, _) in enumerate(self.dirs):
                a[i]["kind"], a[i]["payload"] = k, i
            return a
        if which == V.A_TOKENS:
            a = np.zeros(n, dtype=V.DTYPES[V.A_TOKENS])
            off = 0
            for i, (_, t) in enumerate(self.dirs):
                a[i]["payload"] = off
                off += 4 + ((len(t) + 3) & ~3)
            return a
        if which == V.A_STRINGS:
            out = b""
            for _, t in self.dirs:
                out += len(t).to_bytes(4, "little") + t + b"\0" * (-len(t) % 4)
            return np.frombuffer(out, dtype=np.uint8)
        raise AssertionError(which)

    def join(self, other, split_pos):
        assert split_pos == 0
        pad = -len(self.src) % 64
        return FakeUnit(self.src + b"\n" * pad + other.src)

    def close(self):
        self.closed = True


class FakeCtx:
    def __init__(self, bad=()):
        self.bad, self.parsed = set(bad), []

    def get_definitions(self, src):
        self.parsed.append(src)
        return FakeUnit(src, 1 if src in self.bad else 0)


FILES = {
    "r/main.opus": b'import "lib/b.opus"\nlibrary "m"\nimport "c.opus"\nf() null\n\treturn\n',
    "r/lib/b.opus": b'import "../c.opus"\nimport "d.opus"\nlibrary "pthread"\n',
    "r/c.opus": b'g() null\n\treturn\nimport "main.opus"\n',
    "r/lib/d.opus": b"h() null\n\treturn\n",
}


def read(p):
    try:
        return FILES[os.path.normpath(p)]
    except KeyError:
        raise FileNotFoundError(p)


def test_queue_order_dedupe_and_linking():
    ctx = FakeCtx()
    prog = P.load_program(ctx, "r/main.opus", read=read)
    assert [os.path.normpath(p) for p in prog.source_files] == ["r/main.opus", "r/lib/b.opus", "r/c.opus", "r/lib/d.opus"]
    assert ctx.parsed == [FILES[p] for p in ("r/main.opus", "r/lib/b.opus", "r/c.opus", "r/lib/d.opus")]      # each file parsed once
    assert prog.linking == ["m", "pthread"]
    assert len(prog.filenames) == 6                      # one entry per push, visited or not (frontend.rs:67,103)
    joined = prog.unit.src
    for i, p in enumerate(prog.source_files):
        assert prog.offsets[i] % 64 == 0
        assert joined[prog.offsets[i]:prog.offsets[i] + prog.lengths[i]] == FILES[os.path.normpath(p)]
        assert prog.locate(prog.offsets[i]) == (i, 0) and prog.locate(prog.offsets[i] + prog.lengths[i] - 1) == (i, prog.lengths[i] - 1)


def test_unreadable_file_and_syntax_error_end_the_run():
    files = dict(FILES)
    files["r/main.opus"] = b'import "nope.opus"\n'
    with pytest.raises(P.ProgramError) as e:
        P.load_program(FakeCtx(), "r/main.opus", read=lambda p: files[os.path.normpath(p)] if os.path.normpath(p) in files else (_ for _ in ()).throw(FileNotFoundError(p)))
    assert "could not open file" in str(e.value)
    with pytest.raises(P.ProgramError) as e:
        P.load_program(FakeCtx(bad=[FILES["r/c.opus"]]), "r/main.opus", read=read)
    assert os.path.normpath(e.value.filename) == "r/c.opus" and e.value.unit.status == 1


def test_file_that_is_not_utf8_ends_the_run_like_an_unreadable_one():
    """fs::read_to_string (frontend.rs:85-90) fails on invalid UTF-8: the file must never reach the lexer"""
    files = dict(FILES)
    files["r/lib/d.opus"] = b"h() null\n\tvar s = \"\xff\"\n"
    ctx = FakeCtx()
    with pytest.raises(P.ProgramError) as e:
        P.load_program(ctx, "r/main.opus", read=lambda p: files[os.path.normpath(p)])
    assert "could not open file" in str(e.value) and os.path.normpath(e.value.filename) == "r/lib/d.opus"
    assert files["r/lib/d.opus"] not in ctx.parsed
