"""GPU: import closure (frontend.rs:57-117) — every file parsed as its own unit, units joined on the GPU in
the reference's queue order, lowered as one program.  Expected IR: the oracle on the files concatenated in that order (all
files end in a newline, so the concatenation has the same definition list; IR text carries no positions)."""
import os

import pytest

import oracle_lib as O
from opus_amd import program as P

pytestmark = pytest.mark.gpu


def split_defs(src, parts):
    """cut a generated source at top-level definition starts into `parts` pieces"""
    starts = [0] + [i + 1 for i in range(len(src) - 1) if src[i] == 10 and src[i + 1] not in (9, 10, 32, 59)]
    cuts = [starts[len(starts) * k // parts] for k in range(parts)] + [len(src)]
    return [src[cuts[k]:cuts[k + 1]] for k in range(parts)]


def make_tree(seed=71, size=60_000):
    a, b, c, d = split_defs(O.generate(seed, size), 4)
    files = {
        "root/main.opus": b'import "lib/b.opus"\nlibrary "m"\nimport "c.opus"\n' + a,
        "root/lib/b.opus": b'import "../c.opus"\nimport "d.opus"\nlibrary "pthread"\n' + b,     # c again: visited, skipped
        "root/c.opus": c + b'import "main.opus"\n',                                                 # cycle back to the root
        "root/lib/d.opus": d,
    }
    order = ["root/main.opus", "root/lib/b.opus", "root/c.opus", "root/lib/d.opus"]          # breadth first, duplicates dropped
    return files, order


def test_import_closure_in_memory(fe):
    files, order = make_tree()
    prog = P.load_program(fe, "root/main.opus", read=lambda p: files[os.path.normpath(p)])
    assert [os.path.normpath(p) for p in prog.source_files] == order
    assert prog.linking == ["m", "pthread"]
    # compiler.filenames keeps one entry per push, visited or not (frontend.rs:67,103)
    assert [os.path.normpath(p) for p in prog.filenames] == ["root/main.opus", "root/lib/b.opus", "root/c.opus", "root/c.opus",
                                                             "root/lib/d.opus", "root/main.opus"]
    whole = b"".join(files[p] for p in order)
    assert O.dump(whole, O.ERROR) == "Ok\n"
    assert prog.compile_source() == 0
    assert prog.unit.dump_ir() == O.dump(whole, O.IR)
    assert prog.unit.ir_view().text() == O.dump(whole, O.IR)
    c = prog.unit.counts()
    assert c["definitions"] == sum(O.dump(files[p], O.AST).count(" @ ") for p in order)
    # positions map back to (file, offset): every file starts on a 64-byte boundary of the joined source
    src = prog.unit.array(0).tobytes()
    for i, p in enumerate(order):
        assert prog.offsets[i] % 64 == 0
        assert src[prog.offsets[i]:prog.offsets[i] + prog.lengths[i]] == files[p]
        assert prog.locate(prog.offsets[i] + 5) == (i, 5)


def test_import_closure_on_disk_with_cross_file_calls(fe, tmp_path):
    (tmp_path / "sub").mkdir()
    (tmp_path / "main.opus").write_bytes(b'import "sub/util.opus"\nmain() int32\n\treturn twice(21) + limit\n')
    (tmp_path / "sub" / "util.opus").write_bytes(b'import "../main.opus"\nconst limit int32 = 7\ntwice(x int32) int32\n\treturn x * 2\n')
    prog = P.load_program(fe, str(tmp_path / "main.opus"))
    assert len(prog.source_files) == 2
    assert prog.compile_source() == 0
    whole = (tmp_path / "main.opus").read_bytes() + (tmp_path / "sub" / "util.opus").read_bytes()
    assert prog.unit.dump_ir() == O.dump(whole, O.IR)
    assert "call" in prog.unit.dump_ir()


def test_missing_file_and_syntax_error_end_the_run(fe, tmp_path):
    (tmp_path / "main.opus").write_bytes(b'import "nope.opus"\nf() int32\n\treturn 1\n')
    with pytest.raises(P.ProgramError) as e:
        P.load_program(fe, str(tmp_path / "main.opus"))
    assert "could not open file" in str(e.value) and e.value.filename.endswith("nope.opus")
    (tmp_path / "nope.opus").write_bytes(b"f() int32\n\treturn (1\n")
    with pytest.raises(P.ProgramError) as e:
        P.load_program(fe, str(tmp_path / "main.opus"))
    assert e.value.filename.endswith("nope.opus") and e.value.unit.status == 1
    assert e.value.unit.dump_error() == O.dump(b"f() int32\n\treturn (1\n", O.ERROR)


def test_semantic_error_positions_map_back_to_their_file(fe):
    files = {"m.opus": b'import "u.opus"\nf() int32\n\treturn 1\n', "u.opus": b"g() int32\n\treturn nosuch\n"}
    prog = P.load_program(fe, "m.opus", read=lambda p: files[os.path.normpath(p)])
    assert prog.compile_source() == 0
    errs = prog.unit.array(15)
    assert len(errs) == 1
    i, off = prog.locate(int(errs[0]["low"]))
    assert prog.source_files[i] == "u.opus" and files["u.opus"][off:off + int(errs[0]["len"])] == b"nosuch"


def test_errors_of_a_program_are_rendered_in_their_file(fe, tmp_path):
    from opus_amd import render as R
    Y, RED, Z = "\x1b[33m", "\x1b[31;1m", "\x1b[0m"
    (tmp_path / "m.opus").write_bytes(b'import "u.opus"\nf() int32\n\treturn 1\n')
    (tmp_path / "u.opus").write_bytes(b"g() int32\n\treturn nosuch\n")
    prog = P.load_program(fe, str(tmp_path / "m.opus"))
    assert prog.compile_source() == 0
    name = str(tmp_path / "u.opus")
    assert prog.render_semantic_errors() == (f"{RED}Error:{Z} undefined variable: nosuch\n{Y}{name}:{Z}\n"
                                             f"{Y}   2:{Z}     return {RED}nosuch{Z}\n")
    (tmp_path / "u.opus").write_bytes(b"g() int32\n\treturn )\n")
    with pytest.raises(P.ProgramError) as e:
        P.load_program(fe, str(tmp_path / "m.opus"))
    assert P.render_program_error(e.value) == (f"{RED}Error:{Z} unexpected token: )\n{Y}{name}:{Z}\n{Y}   2:{Z}     return {RED}){Z}\n")
