"""ctypes access to the CPU oracle (oracle/_build/liboracle.so) and the generator — TEST INFRASTRUCTURE."""
import ctypes as C
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_SO = os.path.join(ROOT, "oracle", "_build", "liboracle.so")
GEN_SO = os.path.join(ROOT, "tools", "_build", "libopusgen.so")

TOKENS, AST, IR, ERROR, SEM, LEX, RENDERED = 0, 1, 2, 3, 4, 5, 6


class Counts(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("tokens", "nodes", "instrs", "vars", "definitions", "functions")] + [("status", C.c_int32)]


def _ensure(path, directory):
    if not os.path.exists(path):
        subprocess.check_call(["make", "-C", directory, "-j8"], stdout=subprocess.DEVNULL)
    return C.CDLL(path)


_o = _g = None


def oracle():
    global _o
    if _o is None:
        _o = _ensure(ORACLE_SO, os.path.join(ROOT, "oracle"))
        _o.oracle_dump.argtypes = [C.c_char_p, C.c_uint64, C.c_int, C.POINTER(C.c_uint64)]
        _o.oracle_dump.restype = C.c_void_p
        _o.oracle_free.argtypes = [C.c_void_p]
        _o.oracle_time.argtypes = [C.c_char_p, C.c_uint64, C.c_int, C.POINTER(Counts)]
        _o.oracle_time.restype = C.c_double
    return _o


def gen():
    global _g
    if _g is None:
        _g = _ensure(GEN_SO, os.path.join(ROOT, "tools"))
        _g.opusgen_generate.argtypes = [C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint64, C.c_void_p, C.c_uint64, C.c_int]
        _g.opusgen_generate.restype = C.c_uint64
        _g.opusgen_generate_range.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint32, C.c_void_p, C.c_uint64, C.c_int]
        _g.opusgen_generate_range.restype = C.c_uint64
        _g.opusgen_generate_lines.argtypes = [C.c_uint64, C.c_uint64, C.c_void_p, C.c_uint64]
        _g.opusgen_generate_lines.restype = C.c_uint64
        _g.opusgen_inject.argtypes = [C.c_char_p, C.c_uint64, C.c_uint64, C.c_uint32, C.c_void_p, C.c_uint64]
        _g.opusgen_inject.restype = C.c_uint64
    return _g


def dump(src: bytes, what: int) -> str:
    n = C.c_uint64()
    p = oracle().oracle_dump(src, len(src), what, C.byref(n))
    try:
        return C.string_at(p, n.value).decode("utf-8", errors="surrogateescape")
    finally:
        oracle().oracle_free(p)


def time_run(src: bytes, lower=True):
    c = Counts()
    t = oracle().oracle_time(src, len(src), int(lower), C.byref(c))
    return t, c


def generate(seed: int, nbytes: int, flags=0, block_bytes=0, threads=8) -> bytes:
    buf = C.create_string_buffer(nbytes)
    n = gen().opusgen_generate(seed, nbytes, flags, block_bytes, buf, nbytes, threads)
    assert n == nbytes, "generator failed"
    return buf.raw[:n]


def generate_range(seed: int, first_block: int, n_blocks: int, block_bytes: int, flags=0, threads=8) -> bytes:
    """blocks [first_block, first_block + n_blocks) of generate(seed, ..., block_bytes=block_bytes)"""
    cap = n_blocks * block_bytes
    buf = C.create_string_buffer(cap)
    n = gen().opusgen_generate_range(seed, first_block, n_blocks, block_bytes, flags, buf, cap, threads)
    assert n == cap, "generator failed"
    return buf.raw[:n]


def generate_lines(seed: int, lines: int) -> bytes:
    cap = 64 << 20
    buf = C.create_string_buffer(cap)
    n = gen().opusgen_generate_lines(seed, lines, buf, cap)
    assert n > 0
    return buf.raw[:n]


def inject(src: bytes, seed: int, kind: int) -> bytes:
    cap = len(src) + 64
    buf = C.create_string_buffer(cap)
    n = gen().opusgen_inject(src, len(src), seed, kind, buf, cap)
    return buf.raw[:n] if n else b""
