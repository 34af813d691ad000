"""ctypes binding of libopusfe.so mirroring the reference's frontend.rs seam.

``Context.get_definitions(source)``  <-> frontend.rs:120  get_definitions(compiler, source)
``Unit.compile_source()``            <-> frontend.rs:140  compile_source (finalize + IrGenerator per function)
``Unit.dump_ir()``                   <-> the text `opus -Z` prints (generate_ir.rs:1201-1283)

There is no CPU fallback: if the HIP library is missing or no GPU is present the calls raise.
"""
import ctypes as C
import os

OK, SYNTAX, UNSUPPORTED, ERR_HIP, ERR_OOM, ERR_ARG, ERR_INTERNAL, SEMANTIC = range(8)
_NAMES = ["OK", "SYNTAX", "UNSUPPORTED", "ERR_HIP", "ERR_OOM", "ERR_ARG", "ERR_INTERNAL", "SEMANTIC"]


class OpusfeError(RuntimeError):
    def __init__(self, code, msg=""):
        super().__init__(f"opusfe: {_NAMES[code] if 0 <= code < len(_NAMES) else code}: {msg}")
        self.code = code


class Token(C.Structure):
    _fields_ = [("kind", C.c_uint8), ("flags", C.c_uint8), ("aux", C.c_uint16), ("low", C.c_uint32),
                ("len", C.c_uint32), ("payload", C.c_uint32)]


class Error(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("tok_kind", C.c_uint32), ("low", C.c_uint64), ("len", C.c_uint32),
                ("expected", C.c_uint32), ("payload", C.c_uint64), ("line", C.c_uint64), ("tok_index", C.c_uint32),
                ("_pad", C.c_uint32)]


class Counts(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("source_bytes", "tokens", "nodes", "definitions", "functions", "instrs",
                                          "vars", "bbs", "types")]


def library_path():
    if os.environ.get("OPUSFE_LIB"):          # e.g. the -DOPUSFE_LOWER_PROF build used by tools/ for cycle accounting
        return os.environ["OPUSFE_LIB"]
    return os.path.join(os.path.dirname(os.path.abspath(__file__)), "_build", "libopusfe.so")


_lib = None


def lib():
    global _lib
    if _lib is None:
        path = library_path()
        if not os.path.exists(path):
            raise OpusfeError(ERR_HIP, f"{path} is missing: run `python -c 'import __graft_entry__ as g; g.build()'`")
        L = C.CDLL(path)
        vp, u64, i = C.c_void_p, C.c_uint64, C.c_int
        L.opusfe_create.argtypes = [C.POINTER(vp), i]
        L.opusfe_destroy.argtypes = [vp]
        L.opusfe_last_message.argtypes = [vp]; L.opusfe_last_message.restype = C.c_char_p
        L.opusfe_stream.argtypes = [vp]; L.opusfe_stream.restype = vp
        for name in ("opusfe_parse", "opusfe_parse_device"):
            getattr(L, name).argtypes = [vp, vp, u64, C.POINTER(vp)]
        for name in ("opusfe_lex", "opusfe_lex_device"):
            getattr(L, name).argtypes = [vp, vp, u64, i, C.POINTER(vp)]
        L.opusfe_parse_batch.argtypes = [vp, C.c_uint32, C.POINTER(C.c_char_p), C.POINTER(u64), i, C.POINTER(vp), C.POINTER(i)]
        L.opusfe_parse_group.argtypes = [vp, C.c_uint32, C.POINTER(C.c_char_p), C.POINTER(u64), i, C.POINTER(vp)]
        L.opusfe_parse_group_device.argtypes = [vp, vp, u64, C.c_uint32, C.POINTER(u64), C.POINTER(u64), i, C.POINTER(vp)]
        L.opusfe_group_file_status.argtypes = [vp, C.c_uint32]
        L.opusfe_group_file_error.argtypes = [vp, C.c_uint32, C.POINTER(Error)]
        L.opusfe_group_unit.argtypes = [vp]; L.opusfe_group_unit.restype = vp
        L.opusfe_group_file_unit.argtypes = [vp, C.c_uint32]; L.opusfe_group_file_unit.restype = vp
        L.opusfe_group_file_ranges.argtypes = [vp, C.c_uint32, C.POINTER(u64)]
        L.opusfe_group_stats.argtypes = [vp, C.POINTER(u64), C.POINTER(u64), C.POINTER(u64)]
        L.opusfe_group_dump.argtypes = [vp, C.c_uint32, i, C.POINTER(vp), C.POINTER(u64)]
        L.opusfe_free_group.argtypes = [vp]
        L.opusfe_semantic_error_count.argtypes = [vp, C.POINTER(u64)]
        u32 = C.c_uint32
        L.opusfe_shard_ctx.argtypes = [vp, vp, u64, C.POINTER(u32)]
        L.opusfe_shard_ctx_static.argtypes = [vp, vp, u64, vp]
        L.opusfe_shard_lex.argtypes = [vp, vp, u64, u32, C.POINTER(vp), C.POINTER(u64), C.POINTER(u64)]
        L.opusfe_shard_net_depth.argtypes = [vp, u64, C.POINTER(C.c_int64)]
        L.opusfe_shard_boundary.argtypes = [vp, C.c_int64, C.POINTER(u64)]
        L.opusfe_shard_carry.argtypes = [vp, vp, u64, C.POINTER(u64), C.POINTER(vp)]
        L.opusfe_unit_end_state.argtypes = [vp, C.POINTER(u32), C.POINTER(C.c_int32)]
        L.opusfe_shard_free.argtypes = [vp]
        L.opusfe_unit_headers.argtypes = [vp, C.POINTER(vp), C.POINTER(u64)]
        L.opusfe_unit_join.argtypes = [vp, vp, vp, u64, C.POINTER(vp)]
        L.opusfe_unit_from_arrays.argtypes = [vp, vp, u64, vp, u64, vp, u64, vp, u64, vp, u64, vp, u64, C.POINTER(vp)]
        L.opusfe_lower.argtypes = [vp, vp]
        L.opusfe_unit_status.argtypes = [vp]
        L.opusfe_lowering_stats.argtypes = [vp, C.POINTER(u64), C.POINTER(u64)]
        L.opusfe_parsing_stats.argtypes = [vp, C.POINTER(u64), C.POINTER(u64)]
        L.opusfe_set_parser_mode.argtypes = [vp, i]
        L.opusfe_memory_held.argtypes = [vp, C.POINTER(u64)]
        L.opusfe_memory_stats.argtypes = [vp, C.POINTER(u64)]
        L.opusfe_unit_resync_passes.argtypes = [vp, C.POINTER(u32)]
        L.opusfe_get_counts.argtypes = [vp, C.POINTER(Counts)]
        L.opusfe_get_error.argtypes = [vp, C.POINTER(Error)]
        L.opusfe_copy_tokens.argtypes = [vp, vp, u64, C.POINTER(u64)]
        L.opusfe_array.argtypes = [vp, i, C.POINTER(vp), C.POINTER(u64), C.POINTER(u32)]
        L.opusfe_copy_array.argtypes = [vp, i, vp, u64, C.POINTER(u64)]
        L.opusfe_dump.argtypes = [vp, i, C.POINTER(vp), C.POINTER(u64)]
        L.opusfe_free_text.argtypes = [vp]
        L.opusfe_free_unit.argtypes = [vp]
        L.opusfe_set_profiling.argtypes = [vp, i]
        L.opusfe_kernel_count.argtypes = [vp]
        L.opusfe_kernel_info.argtypes = [vp, i, C.POINTER(C.c_char_p), C.POINTER(C.c_double), C.POINTER(u64)]
        _lib = L
    return _lib


class Unit:
    """Result of get_definitions for one source (tokens + AST; IR after compile_source)."""

    def __init__(self, ctx, handle, status):
        self.ctx, self.handle, self.status = ctx, handle, status

    def close(self):
        if self.handle:
            if self.ctx.handle:          # a destroyed context has already released everything
                lib().opusfe_free_unit(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def counts(self):
        c = Counts()
        lib().opusfe_get_counts(self.handle, C.byref(c))
        return {n: getattr(c, n) for n, _ in Counts._fields_}

    def error(self):
        e = Error()
        if lib().opusfe_get_error(self.handle, C.byref(e)) != OK:
            return None
        return e

    def _dump(self, what):
        out, n = C.c_void_p(), C.c_uint64()
        rc = lib().opusfe_dump(self.handle, what, C.byref(out), C.byref(n))
        if rc != OK:
            raise OpusfeError(rc, self.ctx.message())
        try:
            return C.string_at(out, n.value).decode("utf-8", errors="surrogateescape")
        finally:
            lib().opusfe_free_text(out)

    def dump_tokens(self):
        return self._dump(0)

    def dump_ast(self):
        return self._dump(1)

    def dump_ir(self):
        return self._dump(2)

    def dump_error(self):
        return self._dump(3)

    def resync_passes(self):
        """how many extra lex + parse passes the unit needed for swallowed lexical errors (0 for well-formed input)"""
        n = C.c_uint32()
        lib().opusfe_unit_resync_passes(self.handle, C.byref(n))
        return n.value

    def end_state(self):
        """(lexer context, bracket depth) behind the unit's last byte, as the reference's lexer has them after every resync pass"""
        st, d = C.c_uint32(), C.c_int32()
        lib().opusfe_unit_end_state(self.handle, C.byref(st), C.byref(d))
        return st.value, d.value

    def compile_source(self):
        """frontend.rs:140-187, IR part.  Returns the status (OK / SEMANTIC / UNSUPPORTED)."""
        rc = lib().opusfe_lower(self.ctx.handle, self.handle)
        if rc not in (OK, SEMANTIC, UNSUPPORTED):
            raise OpusfeError(rc, self.ctx.message())
        self.status = rc
        return rc

    def parsing_stats(self):
        """(top-level segments parsed by the data-parallel parser, segments parsed by the sequential machine)"""
        a, b = C.c_uint64(), C.c_uint64()
        lib().opusfe_parsing_stats(self.handle, C.byref(a), C.byref(b))
        return a.value, b.value

    def lowering_stats(self):
        """(functions lowered by the node-parallel kernels, functions lowered by the sequential machine)"""
        a, b = C.c_uint64(), C.c_uint64()
        lib().opusfe_lowering_stats(self.handle, C.byref(a), C.byref(b))
        return a.value, b.value

    def join(self, hdr, split_pos):
        """self (+) hdr (opusfe_unit_join): a new parsed Unit with hdr's definitions that start before byte `split_pos` of
        hdr's source, then self's, then the rest of hdr's."""
        out = C.c_void_p()
        rc = lib().opusfe_unit_join(self.ctx.handle, self.handle, hdr.handle, int(split_pos), C.byref(out))
        if rc != OK:
            raise OpusfeError(rc, self.ctx.message())
        return Unit(self.ctx, out, OK)

    def headers_device(self):
        """(device pointer, length) of the definition-header text other shards need (opusfe_unit_headers)."""
        p, n = C.c_void_p(), C.c_uint64()
        rc = lib().opusfe_unit_headers(self.handle, C.byref(p), C.byref(n))
        if rc != OK:
            raise OpusfeError(rc, self.ctx.message())
        return p.value or 0, n.value

    def array(self, which):
        """numpy copy of one flat result array (opusfe_copy_array; ids and record layouts in opus_amd/ir_view.py)"""
        import numpy as np
        from .ir_view import DTYPES
        n = C.c_uint64()
        if lib().opusfe_copy_array(self.handle, which, None, 0, C.byref(n)) != OK:
            raise OpusfeError(ERR_ARG, f"no array {which}")
        arr = np.empty(n.value, dtype=DTYPES[which])
        rc = lib().opusfe_copy_array(self.handle, which, arr.ctypes.data, n.value, C.byref(n))
        if rc != OK:
            raise OpusfeError(rc, self.ctx.message())
        return arr

    def array_device(self, which):
        """(device pointer, element count, element size) of one flat result array (opusfe_array)"""
        p, n, es = C.c_void_p(), C.c_uint64(), C.c_uint32()
        if lib().opusfe_array(self.handle, which, C.byref(p), C.byref(n), C.byref(es)) != OK:
            raise OpusfeError(ERR_ARG, f"no array {which}")
        return p.value or 0, n.value, es.value

    def ir_view(self):
        """backend hand-off view over the arrays (opus_amd/ir_view.py)"""
        from .ir_view import IrView
        return IrView(self)

    def tokens(self):
        n = C.c_uint64()
        lib().opusfe_copy_tokens(self.handle, None, 0, C.byref(n))
        arr = (Token * n.value)()
        rc = lib().opusfe_copy_tokens(self.handle, arr, n.value, C.byref(n))
        if rc != OK:
            raise OpusfeError(rc, self.ctx.message())
        return arr


class _BorrowedUnit(Unit):
    """a unit owned by a Group (the packed unit, or the unit of a file that took the single-file path): never freed from here"""

    def close(self):
        self.handle = None


class Group:
    """Result of Context.compile_group: a corpus of independent files answered by one packed pass (opusfe_parse_group)."""

    def __init__(self, ctx, handle, n_files):
        self.ctx, self.handle, self.n_files = ctx, handle, n_files

    def close(self):
        if self.handle:
            if self.ctx.handle:
                lib().opusfe_free_group(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __len__(self):
        return self.n_files

    def status(self, f):
        return lib().opusfe_group_file_status(self.handle, f)

    def statuses(self):
        return [self.status(f) for f in range(self.n_files)]

    def error(self, f):
        e = Error()
        if lib().opusfe_group_file_error(self.handle, f, C.byref(e)) != OK:
            return None
        return e

    def stats(self):
        """(files answered by the packed pass, files that took the single-file path, files with a syntax error)"""
        a, b, c = C.c_uint64(), C.c_uint64(), C.c_uint64()
        lib().opusfe_group_stats(self.handle, C.byref(a), C.byref(b), C.byref(c))
        return a.value, b.value, c.value

    def ranges(self, f):
        r = (C.c_uint64 * 10)()
        if lib().opusfe_group_file_ranges(self.handle, f, r) != OK:
            raise OpusfeError(ERR_ARG, "opusfe_group_file_ranges")
        return [int(x) for x in r]

    def unit(self):
        return _BorrowedUnit(self.ctx, C.c_void_p(lib().opusfe_group_unit(self.handle)), OK)

    def file_unit(self, f):
        h = lib().opusfe_group_file_unit(self.handle, f)
        return _BorrowedUnit(self.ctx, C.c_void_p(h), self.status(f)) if h else None

    def _dump(self, f, what):
        out, n = C.c_void_p(), C.c_uint64()
        rc = lib().opusfe_group_dump(self.handle, f, what, C.byref(out), C.byref(n))
        if rc != OK:
            raise OpusfeError(rc, self.ctx.message())
        try:
            return C.string_at(out, n.value).decode("utf-8", errors="surrogateescape")
        finally:
            lib().opusfe_free_text(out)

    def dump_tokens(self, f):
        return self._dump(f, 0)

    def dump_ast(self, f):
        return self._dump(f, 1)

    def dump_ir(self, f):
        return self._dump(f, 2)

    def dump_error(self, f):
        return self._dump(f, 3)

    def dump_semantic_errors(self, f):
        return self._dump(f, 4)


def pack_group(sources):
    """host-side packing for opusfe_parse_group_device: (packed bytes, starts, lens) — file f at a 64-byte aligned start,
    '\n' in between (at least one)"""
    starts, total = [], 0
    for s in sources:
        starts.append(total)
        total = (total + len(s) + 1 + 63) // 64 * 64
    buf = bytearray(b"\n" * (total + 64))
    for st, s in zip(starts, sources):
        buf[st:st + len(s)] = s
    return bytes(buf), starts, [len(s) for s in sources], total


class Context:
    def __init__(self, device=0):
        h = C.c_void_p()
        rc = lib().opusfe_create(C.byref(h), device)
        if rc != OK:
            raise OpusfeError(rc, "opusfe_create failed (no MI355X visible?)")
        self.handle = h

    def close(self):
        if self.handle:
            lib().opusfe_destroy(self.handle)
            self.handle = None

    def message(self):
        return (lib().opusfe_last_message(self.handle) or b"").decode()

    def stream(self):
        return lib().opusfe_stream(self.handle)

    def memory_held(self):
        """device bytes this context holds (live results + cached blocks): high-water mark of its HBM footprint"""
        n = C.c_uint64()
        lib().opusfe_memory_held(self.handle, C.byref(n))
        return n.value

    def memory_stats(self):
        """device bytes by kind: live results, cached blocks, the temporaries' arena, high-water mark"""
        st = (C.c_uint64 * 4)()
        lib().opusfe_memory_stats(self.handle, st)
        return {"live": st[0], "cached": st[1], "arena": st[2], "peak": st[3]}

    def _finish(self, rc, out):
        if rc not in (OK, SYNTAX, UNSUPPORTED):
            if out.value:
                lib().opusfe_free_unit(out)
            raise OpusfeError(rc, self.message())
        return Unit(self, out, rc)

    def get_definitions(self, source: bytes):
        """frontend.rs:120 — lex + parse one source held in host memory."""
        out = C.c_void_p()
        buf = C.c_char_p(source)
        return self._finish(lib().opusfe_parse(self.handle, C.cast(buf, C.c_void_p), len(source), C.byref(out)), out)

    def get_definitions_device(self, d_ptr: int, length: int):
        out = C.c_void_p()
        return self._finish(lib().opusfe_parse_device(self.handle, C.c_void_p(d_ptr), length, C.byref(out)), out)

    def unit_from_arrays(self, arrays):
        """opusfe_unit_from_arrays: `arrays` = six (device pointer, element count) pairs in the order source, tokens, val64,
        strings, nodes, definitions (what Unit.array_device gives for those arrays)"""
        out = C.c_void_p()
        args = []
        for p, n in arrays:
            args += [C.c_void_p(p), int(n)]
        return self._finish(lib().opusfe_unit_from_arrays(self.handle, *args, C.byref(out)), out)

    def compile_corpus(self, sources, lower=True):
        """Corpus of independent files (one reference run per file): list of Units, each with its own status."""
        n = len(sources)
        srcs = (C.c_char_p * n)(*sources)
        lens = (C.c_uint64 * n)(*[len(s) for s in sources])
        units = (C.c_void_p * n)()
        statuses = (C.c_int * n)()
        rc = lib().opusfe_parse_batch(self.handle, n, srcs, lens, int(lower), units, statuses)
        out = [Unit(self, C.c_void_p(units[k]), statuses[k]) for k in range(n)]
        if rc != OK:
            raise OpusfeError(rc, self.message())
        return out

    def compile_group(self, sources, lower=True):
        """Corpus of independent files as ONE packed pass (opusfe_parse_group): a Group with a status per file."""
        n = len(sources)
        srcs = (C.c_char_p * n)(*sources)
        lens = (C.c_uint64 * n)(*[len(s) for s in sources])
        out = C.c_void_p()
        rc = lib().opusfe_parse_group(self.handle, n, srcs, lens, int(lower), C.byref(out))
        if rc != OK:
            raise OpusfeError(rc, self.message())
        return Group(self, out, n)

    def compile_group_device(self, d_ptr, packed_len, starts, lens, lower=True):
        """opusfe_parse_group_device: the packed buffer (pack_group's layout) is already in HBM"""
        n = len(starts)
        st = (C.c_uint64 * n)(*starts)
        ln = (C.c_uint64 * n)(*lens)
        out = C.c_void_p()
        rc = lib().opusfe_parse_group_device(self.handle, C.c_void_p(d_ptr), packed_len, n, st, ln, int(lower), C.byref(out))
        if rc != OK:
            raise OpusfeError(rc, self.message())
        return Group(self, out, n)

    def lex(self, source: bytes, force_dents=False):
        out = C.c_void_p()
        buf = C.c_char_p(source)
        return self._finish(lib().opusfe_lex(self.handle, C.cast(buf, C.c_void_p), len(source), int(force_dents), C.byref(out)), out)

    def lex_device(self, d_ptr: int, length: int, force_dents=False):
        out = C.c_void_p()
        return self._finish(lib().opusfe_lex_device(self.handle, C.c_void_p(d_ptr), length, int(force_dents), C.byref(out)), out)

    # ---- shard carries (one huge source over several GPUs) ----
    def shard_ctx(self, d_ptr, length):
        fn = C.c_uint32()
        rc = lib().opusfe_shard_ctx(self.handle, C.c_void_p(d_ptr), length, C.byref(fn))
        if rc != OK:
            raise OpusfeError(rc, self.message())
        return fn.value

    def shard_ctx_static(self, d_ptr, length, d_out):
        """opusfe_shard_ctx_static: no allocation, result written to the device word `d_out` on this context's stream"""
        rc = lib().opusfe_shard_ctx_static(self.handle, C.c_void_p(d_ptr), length, C.c_void_p(d_out))
        if rc != OK:
            raise OpusfeError(rc, self.message())

    def shard_lex(self, d_ptr, length, start_state):
        h, n, tail = C.c_void_p(), C.c_uint64(), C.c_uint64()
        rc = lib().opusfe_shard_lex(self.handle, C.c_void_p(d_ptr), length, start_state, C.byref(h), C.byref(n), C.byref(tail))
        if rc != OK:
            raise OpusfeError(rc, self.message())
        return h, n.value, tail.value

    def shard_net_depth(self, shard, prev_kinds):
        net = C.c_int64()
        rc = lib().opusfe_shard_net_depth(shard, prev_kinds, C.byref(net))
        if rc != OK:
            raise OpusfeError(rc, self.message())
        return net.value

    def shard_boundary(self, shard, depth_in):
        pos = C.c_uint64()
        rc = lib().opusfe_shard_boundary(shard, depth_in, C.byref(pos))
        if rc != OK:
            raise OpusfeError(rc, self.message())
        return None if pos.value == 0xFFFFFFFFFFFFFFFF else pos.value

    def shard_carry(self, d_ptr, length):
        """opusfe_shard_carry: (shard handle, the eight words of the fixed-size carry) computed for the common entry (context N, depth 0)"""
        h = C.c_void_p()
        carry = (C.c_uint64 * 8)()
        rc = lib().opusfe_shard_carry(self.handle, C.c_void_p(d_ptr), length, carry, C.byref(h))
        if rc != OK:
            raise OpusfeError(rc, self.message())
        return h, [int(x) for x in carry]

    def shard_free(self, shard):
        lib().opusfe_shard_free(shard)

    def set_parser_mode(self, mode):
        """0: data-parallel parser with the sequential machine as fallback (default); 1: the sequential machine alone"""
        rc = lib().opusfe_set_parser_mode(self.handle, int(mode))
        if rc != OK:
            raise OpusfeError(rc, "opusfe_set_parser_mode")

    def set_profiling(self, on=True):
        lib().opusfe_set_profiling(self.handle, int(on))

    def kernel_stats(self):
        out = []
        for k in range(lib().opusfe_kernel_count(self.handle)):
            name, ms, n = C.c_char_p(), C.c_double(), C.c_uint64()
            lib().opusfe_kernel_info(self.handle, k, C.byref(name), C.byref(ms), C.byref(n))
            out.append((name.value.decode(), ms.value, n.value))
        return out
