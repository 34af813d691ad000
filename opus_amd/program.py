"""Import closure of a root file: the step before the hot path.

Mirrors frontend.rs:57-117.  The reference pops files from a queue (breadth first), skips a file whose canonical path it has
already visited (:77-83), reads it (:85-92; an unreadable file ends the run), parses it with ``get_definitions`` (:96; a syntax
error ends the run, :130-133), and lets ``preload_definition`` (compile.rs:968-981) turn every ``import "<path>"`` into a new
queue entry relative to the importing file's directory (:99-105) and every ``library "<name>"`` into a linker argument
(:106-108).  All definitions of all files, in that order, then go through ``compile_source`` (:117) as one program.

Here every file is parsed on the GPU as a unit of its own (``opusfe_parse``) and the units are joined on the GPU in queue order
(``opusfe_unit_join``: arrays concatenated and re-based, nothing is parsed twice), which gives ``opusfe_lower`` exactly the
definition list the reference builds.  Positions in the joined unit are offsets into the concatenation of the sources (each
file starts on a 64-byte boundary); ``Program.locate`` maps them back to (file, offset).  Host work here is file IO and the
queue; lexing, parsing, joining and lowering are the HIP path (no CPU fallback).
"""
import os

from . import frontend as F
from . import ir_view as V

D_IMPORT, D_LIBRARY = 5, 6          # defs.h NodeKind (parse.rs:333-350 Definition::Import / Library)


class ProgramError(RuntimeError):
    """what ends the reference's run before compile_source: an unreadable file or a syntax error (with the file it is in)"""

    def __init__(self, msg, filename=None, unit=None):
        super().__init__(msg)
        self.filename, self.unit = filename, unit


class Program:
    """Result of the import closure: the joined unit plus the reference's `filenames`, `sources` and `linking` lists."""

    def __init__(self, unit, filenames, source_files, offsets, lengths, linking):
        self.unit = unit
        self.filenames = filenames          # compiler.filenames: one entry per queue push (frontend.rs:67,103), duplicates kept
        self.source_files = source_files    # the files actually read, in queue order (compiler.sources, frontend.rs:92)
        self.offsets, self.lengths = offsets, lengths
        self.linking = linking              # options.linking (frontend.rs:107)

    def locate(self, pos):
        """(index into source_files, offset inside that file) of a position of the joined unit"""
        for i in range(len(self.offsets) - 1, -1, -1):
            if pos >= self.offsets[i]:
                return i, pos - self.offsets[i]
        raise ValueError(pos)

    def compile_source(self):
        return self.unit.compile_source()

    def render_semantic_errors(self, read=None):
        """stderr text of the unit's semantic diagnostics in report order (compile.rs:782-917 + frontend.rs:201-256), each
        located in its own file.  The reference indexes `filenames` by the position of the source in `sources`
        (frontend.rs:204-209), so after a duplicate import the printed name is the one pushed at that index, as there."""
        from . import render as R
        if read is None:
            def read(path):
                with open(path, "rb") as f:
                    return f.read()
        out = []
        view = self.unit.ir_view()
        for e in view.sem_errors:
            i, off = self.locate(int(e["low"]))
            out.append(R.render_semantic_record(e, view, self.filenames[i], read(self.source_files[i]), off))
        return "".join(out)


def render_program_error(err):
    """stderr text of the ProgramError that ended load_program: the unreadable-file line (frontend.rs:88) or the parse error
    of the file it was found in (frontend.rs:130-133)"""
    from . import render as R
    if err.unit is None:
        return str(err) + "\n"
    if err.unit.status != F.SYNTAX:
        return f"{err.filename}: {F._NAMES[err.unit.status]}\n"
    return R.render_parse_error(err.unit, err.filename)


def _directives(unit):
    """(kind, bytes) of the Import / Library definitions of a parsed unit, in definition order"""
    nodes, defs = unit.array(V.A_NODES), unit.array(V.A_DEFINITIONS)
    roots = nodes[defs] if len(defs) else nodes[:0]
    picks = [i for i in range(len(roots)) if int(roots[i]["kind"]) in (D_IMPORT, D_LIBRARY)]
    if not picks:
        return []
    tokens, strings = unit.array(V.A_TOKENS), unit.array(V.A_STRINGS).tobytes()
    out = []
    for i in picks:
        p = int(tokens[int(roots[i]["payload"])]["payload"])
        n = int.from_bytes(strings[p:p + 4], "little")
        out.append((int(roots[i]["kind"]), strings[p + 4:p + 4 + n]))
    return out


def load_program(ctx, root, read=None):
    """frontend.rs:57-113 for the root file `root`; returns a Program whose unit is parsed (not yet lowered).
    `read(path) -> bytes` replaces the file system in tests."""
    if read is None:
        def read(path):
            with open(path, "rb") as f:
                return f.read()
        canonical = lambda p: os.path.realpath(p) if os.path.exists(p) else None      # Path::canonicalize fails if missing
    else:
        canonical = lambda p: os.path.normpath(p)
    queue, filenames = [root], [root]
    visited = set()
    acc, acc_len = None, 0
    source_files, offsets, lengths, linking = [], [], [], []
    while queue:
        path = queue.pop(0)
        c = canonical(path)
        if c is not None:
            if c in visited:
                continue
            visited.add(c)
        try:
            src = read(path)
        except OSError:
            raise ProgramError(f"error: could not open file {path!r}", path)          # frontend.rs:86-90
        try:
            src.decode("utf-8")      # fs::read_to_string (frontend.rs:85) fails on invalid UTF-8: same message, same exit
        except UnicodeDecodeError:
            raise ProgramError(f"error: could not open file {path!r}", path)
        unit = ctx.get_definitions(src)
        if unit.status != F.OK:                                                          # frontend.rs:130-133 (or UNSUPPORTED)
            raise ProgramError(f"{path}: {F._NAMES[unit.status]}", path, unit)
        directory = os.path.dirname(path)
        for kind, text in _directives(unit):
            if kind == D_IMPORT:
                child = os.path.join(directory, text.decode("utf-8", errors="replace"))  # String::from_utf8_lossy
                filenames.append(child)
                queue.append(child)
            else:
                linking.append(text.decode("utf-8", errors="replace"))
        if acc is None:
            acc, offset = unit, 0
        else:
            offset = (acc_len + 63) & ~63                  # opusfe_unit_join: the second source starts on a 64-byte boundary
            joined = acc.join(unit, 0)                     # acc's definitions, then all of the new file's
            acc.close()
            unit.close()
            acc = joined
        acc_len = offset + len(src)
        source_files.append(path)
        offsets.append(offset)
        lengths.append(len(src))
    return Program(acc, filenames, source_files, offsets, lengths, linking)
