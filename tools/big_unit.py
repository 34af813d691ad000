"""one unit of SIZE bytes (default 4e9: the 2-GPU point of BASELINE config 4 is 4 GB per rank) through lex + parse + lower on one
GPU: size-independent properties (the per-function headers tile the dense arrays, lowering twice gives the same totals) and the
HBM the context holds.  usage: big_unit.py [SIZE]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import opus_amd
from opus_amd import ir_view as V
import oracle_lib as O

size = int(float(sys.argv[1])) if len(sys.argv) > 1 else 4_000_000_000
t0 = time.time()
blk = 1 << 26
nblk = size // blk
src = O.generate_range(3, 0, nblk, blk, flags=1, threads=16)
print("generated", len(src), "bytes in", round(time.time() - t0, 1), "s", flush=True)
d = torch.frombuffer(bytearray(src), dtype=torch.uint8).cuda()
del src
ctx = opus_amd.Context(0)
t0 = time.perf_counter()
u = ctx.get_definitions_device(d.data_ptr(), d.numel())
torch.cuda.synchronize()
print("parse status", u.status, round((time.perf_counter() - t0) * 1e3, 1), "ms", ctx.message(), flush=True)
assert u.status == 0
t0 = time.perf_counter()
rc = u.compile_source()
print("lower status", rc, round((time.perf_counter() - t0) * 1e3, 1), "ms", ctx.message(), flush=True)
assert rc == 0
c = u.counts()
print(c, flush=True)
print("memory", ctx.memory_stats(), "bytes per source byte (peak)", round(ctx.memory_stats()["peak"] / d.numel(), 2), flush=True)
f = u.array(V.A_FUNCTIONS)
body = f[f["n_bbs"] > 0]
for name, base, total in (("n_vars", "var_base", c["vars"]), ("n_instrs", "instr_base", c["instrs"]), ("n_bbs", "bb_base", c["bbs"])):
    o = np.argsort(body[base], kind="stable")
    b, n = body[base][o].astype(np.int64), body[name][o].astype(np.int64)
    assert b[0] == 0 and (b[1:] == (b + n)[:-1]).all() and b[-1] + n[-1] == total, name
off = u.array(V.A_BB_OFFSETS).astype(np.int64)
assert off[0] == 0 and off[-1] == c["instrs"] and (np.diff(off) >= 0).all()
del off
par, seq = u.lowering_stats()
assert par + seq == len(body) and par > 0.99 * len(body)
# the last 64 MB block alone must give what the unit holds for it (positions beyond 2^31, indices beyond 2^29)
defs = u.array(V.A_DEFINITIONS)
roots = u.array(V.A_NODES)["low"][defs].astype(np.int64)
assert (np.diff(roots) > 0).all()
k0 = int(np.searchsorted(roots, d.numel() - blk))
tail_lo = int(roots[k0])
w = ctx.get_definitions_device(d.data_ptr() + tail_lo, d.numel() - tail_lo)
assert w.status == 0
wc = w.counts()
# (parse level only: the generator's prelude — Range, `operator for` — lives in block 0, so the tail alone does not type-check)
assert wc["definitions"] == len(defs) - k0, (wc["definitions"], len(defs) - k0)
assert wc["nodes"] == c["nodes"] - int(defs[k0]), (wc["nodes"], c["nodes"] - int(defs[k0]))
wn, un = w.array(V.A_NODES), u.array(V.A_NODES)
ut = un[int(defs[k0]):]
assert (wn["kind"] == ut["kind"]).all() and (wn["low"].astype(np.int64) + tail_lo == ut["low"]).all() and (wn["high"].astype(np.int64) + tail_lo == ut["high"]).all()
del wn, un, ut
print("tail block:", wc["definitions"], "definitions,", wc["nodes"], "nodes: kinds and spans identical to the unit's last", wc["nodes"], "nodes", flush=True)
w.close()
assert u.compile_source() == 0 and u.counts() == c
print("OK", flush=True)
u.close(); ctx.close()
