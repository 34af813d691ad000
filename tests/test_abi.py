"""CPU: the C-ABI library loads and exports every symbol include/opusfe.h declares; record layouts match."""
import ctypes as C
import os
import re

import opus_amd
from opus_amd import frontend as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    text = open(os.path.join(ROOT, "include", "opusfe.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(opusfe_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = C.CDLL(opus_amd.library_path())
    names = declared_functions()
    assert len(names) >= 19
    for n in names:
        assert hasattr(lib, n), f"{n} is declared in include/opusfe.h but not exported"


def test_record_sizes_match_the_header():
    assert C.sizeof(F.Token) == 16
    assert C.sizeof(F.Error) == 48
    assert C.sizeof(F.Counts) == 9 * 8
    # flat result arrays (opusfe_array): numpy record layouts of opus_amd/ir_view.py against the sizes in the header
    from opus_amd import ir_view as V
    import numpy as np
    sizes = {V.A_TOKENS: 16, V.A_NODES: 24, V.A_INSTRS: 24, V.A_VARIABLES: 8, V.A_TYPES: 16, V.A_SIGNATURES: 24, V.A_GLOBALS: 16,
             V.A_FUNCTIONS: 64, V.A_SEM_ERRORS: 36, V.A_VAL64: 8, V.A_DEFINITIONS: 4}
    for which, size in sizes.items():
        assert np.dtype(V.DTYPES[which]).itemsize == size, which
    assert len(V.DTYPES) == 17


def test_hand_off_text_helpers_follow_rust_formatting():
    from opus_amd import ir_view as V
    assert V.rust_f64(1.0) == "1" and V.rust_f64(0.1) == "0.1" and V.rust_f64(1e21) == "1000000000000000000000"
    assert V.rust_f64(1.5e-7) == "0.00000015" and V.rust_f64(float("inf")) == "inf"
    assert V.rust_debug_str(b'a"\n\x01\\') == '"a\\"\\n\\u{1}\\\\"'


def test_no_gpu_means_loud_failure_not_a_cpu_path():
    import torch
    if torch.cuda.is_available():
        return
    try:
        opus_amd.Context(0)
    except opus_amd.OpusfeError as e:
        assert e.code == F.ERR_HIP
    else:
        raise AssertionError("opusfe_create succeeded without a GPU")


def test_integration_doc_binds_every_entry_point():
    """INTEGRATION.md shows the reference-side binding: one extern declaration per function of include/opusfe.h"""
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    missing = [n for n in declared_functions() if f"pub fn {n}(" not in text]
    assert not missing, missing


def test_static_library_links_into_a_plain_c_program(tmp_path):
    """libopusfe.a: a C program that includes only include/opusfe.h links against the archive with the HIP
    runtime and the C++ runtime, and every declared entry point is defined in it"""
    import shutil
    import subprocess
    static = os.path.join(os.path.dirname(opus_amd.library_path()), "libopusfe.a")
    assert os.path.exists(static), "build() writes libopusfe.a next to libopusfe.so"
    nm = subprocess.run(["nm", "--defined-only", static], capture_output=True, text=True, check=True).stdout
    defined = set(re.findall(r" T (opusfe_[a-z0-9_]+)$", nm, flags=re.M))
    assert set(declared_functions()) <= defined
    cc = shutil.which("gcc")
    src = tmp_path / "main.c"
    src.write_text('#include "opusfe.h"\n#include <stdio.h>\nint main(void) { opusfe_ctx* c = 0; int rc = opusfe_create(&c, 0);\n'
                   ' printf("rc=%d\\n", rc); if (rc == OPUSFE_OK) opusfe_destroy(c); return 0; }\n')
    exe = tmp_path / "main"
    subprocess.run([cc, "-I", os.path.join(ROOT, "include"), str(src), static, "-L/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib", "-lamdhip64", "-lstdc++",
                    "-lpthread", "-lm", "-ldl", "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and re.match(r"rc=-?\d+", out.stdout)      # OPUSFE_ERR_HIP without a GPU, OPUSFE_OK with one: never a crash
