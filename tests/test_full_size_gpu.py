"""GPU: the full-size configuration (BASELINE configs[1]: one source of 1 GB) through size-independent properties — the oracle
cannot run a gigabyte inside a test.  At this size the byte offsets of the instruction array pass 4 GiB and node / token indices
pass 2^27, so the 32-bit edges of the layout are touched."""
import numpy as np
import pytest

import oracle_lib as O
from opus_amd import ir_view as V

pytestmark = pytest.mark.gpu

SIZE = 1_000_000_000
BLOCK = 1 << 26


def test_one_gigabyte_properties():
    import opus_amd
    src = O.generate(3, SIZE, flags=1, threads=16)
    ctx = opus_amd.Context(0)       # a context of its own: its cached blocks go away with it
    try:
        u = ctx.get_definitions(src)
        assert u.status == 0
        c = u.counts()
        # (1) parse-level additivity: the definitions are independent, so the unit cut at top-level definition starts and parsed
        #     piece by piece gives the same definitions, nodes and tokens (each piece adds one EndOfFile token of its own)
        defs = u.array(V.A_DEFINITIONS)
        roots = u.array(V.A_NODES)["low"][defs]
        assert (np.diff(roots.astype(np.int64)) > 0).all()
        cuts = [0]
        for k in range(1, 4):
            cuts.append(int(roots[np.searchsorted(roots, k * SIZE // 4)]))
        cuts.append(len(src))
        tot = {"tokens": 0, "nodes": 0, "definitions": 0}
        for a, b in zip(cuts[:-1], cuts[1:]):
            w = ctx.get_definitions(src[a:b])
            assert w.status == 0
            for k in tot:
                tot[k] += w.counts()[k]
            w.close()
        assert tot["definitions"] == c["definitions"] and tot["nodes"] == c["nodes"] and tot["tokens"] - 3 == c["tokens"]
        # (2) the first 64 MB block (whole definitions) against the oracle: same counts as bench.py's parity check, here as a test
        head = src[:int(roots[np.searchsorted(roots, BLOCK)])]
        t, oc = O.time_run(head)
        w = ctx.get_definitions(head)
        assert w.compile_source() == 0
        wc = w.counts()
        assert oc.status == 0 and (oc.tokens, oc.definitions, oc.instrs, oc.vars) == (wc["tokens"], wc["definitions"], wc["instrs"], wc["vars"])
        w.close()
        # (3) lowering the gigabyte: totals are consistent with the per-function headers, which tile the dense arrays
        assert u.compile_source() == 0
        c = u.counts()
        assert c["instrs"] * 24 > 1 << 32, "the instruction array must pass 4 GiB for this test to mean anything"
        f = u.array(V.A_FUNCTIONS)
        body = f[f["n_bbs"] > 0]
        for name, base, total in (("n_vars", "var_base", c["vars"]), ("n_instrs", "instr_base", c["instrs"]), ("n_bbs", "bb_base", c["bbs"])):
            o = np.argsort(body[base], kind="stable")
            b, n = body[base][o].astype(np.int64), body[name][o].astype(np.int64)
            assert b[0] == 0 and (b[1:] == (b + n)[:-1]).all() and b[-1] + n[-1] == total, name
        off = u.array(V.A_BB_OFFSETS).astype(np.int64)
        assert off[0] == 0 and off[-1] == c["instrs"] and (np.diff(off) >= 0).all()
        par, seq = u.lowering_stats()
        assert par + seq == len(body) and par > 0.99 * len(body)
        # (4) idempotence: lowering again gives the same totals
        assert u.compile_source() == 0 and u.counts() == c
        # (5) the boundary reports the same sizes for the device arrays
        ins = u.array_device(V.A_INSTRS)
        assert ins[1] == c["instrs"] and ins[2] == 24
    finally:
        ctx.close()


def test_four_gigabyte_unit_fits_one_gpu():
    """The 2-GPU point of BASELINE config 4 is 4 GB per rank, the edge of the u32 positions (4 GiB - 64 KiB): one unit of
    3.96e9 bytes through lex + parse + lower on ONE GPU (tools/big_unit.py, a process of its own so that its HBM goes away with
    it) — the per-function headers tile the dense arrays, the last 64 MB block parsed alone has the kinds and spans of the unit's
    last nodes (positions beyond 2^31, node indices beyond 2^29), lowering twice gives the same totals, and the context's
    high-water mark stays under 33 B of HBM per source byte (round 3: the flat lowering keeps 54 B of per-node state in HBM while it runs)."""
    import os
    import re
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-u", os.path.join(root, "tools", "big_unit.py"), "4e9"], capture_output=True, text=True, timeout=900, cwd=root)
    assert out.returncode == 0 and out.stdout.rstrip().endswith("OK"), (out.stdout[-1500:], out.stderr[-1500:])
    per_byte = float(re.search(r"bytes per source byte \(peak\) ([0-9.]+)", out.stdout).group(1))
    assert per_byte < 33.0, per_byte
