"""opus_amd — MI355X-native front end (lex -> parse -> IR) for the Opus language of ftxqxd/opus.

The product is the hipcc-built C-ABI library ``libopusfe.so`` (include/opusfe.h).  This package is the
thin Python host mirror used by the tests and by bench.py; it never falls back to a CPU path.
"""
from .frontend import Context, Unit, OpusfeError, OK, SYNTAX, UNSUPPORTED, library_path  # noqa: F401
from .program import load_program, Program, ProgramError  # noqa: F401,E402
