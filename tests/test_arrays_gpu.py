"""GPU: the flat result arrays of the boundary (opusfe_array / opusfe_copy_array) and the backend hand-off view
built on them (opus_amd/ir_view.py; backend/mod.rs:6-10, llvm.rs:355-420 read exactly these fields)."""
import json
import os

import numpy as np
import pytest

import oracle_lib as O
from opus_amd import ir_view as V

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
with open(os.path.join(HERE, "golden", "kats.json")) as f:
    KATS = json.load(f)


def test_arrays_of_a_kat_render_the_hand_derived_ir(fe):
    for k in ("kat1", "kat2", "kat3"):
        u = fe.get_definitions(KATS[k]["src"].encode())
        assert u.compile_source() == 0
        assert u.ir_view().text() == KATS[k]["ir"]


def test_array_counts_agree_with_counts_and_structure(fe):
    src = O.generate(31, 300_000)
    u = fe.get_definitions(src)
    assert u.compile_source() == 0
    c, v = u.counts(), u.ir_view()
    assert len(v.source) == len(src) and v.source == src
    assert len(v.tokens) == c["tokens"] and len(v.nodes) == c["nodes"] and len(v.definitions) == c["definitions"]
    assert len(v.functions) == c["functions"] == len(v.signatures)
    assert len(v.variables) == c["vars"] and len(v.instrs) == c["instrs"] and len(v.bb_offsets) == c["bbs"] + 1
    assert len(v.types) == c["types"] and len(v.node_types) == c["nodes"]
    # bb_offsets is a monotone partition of the instruction array; the per-function bases tile the global arrays
    off = v.bb_offsets.astype(np.int64)
    assert off[0] == 0 and off[-1] == len(v.instrs) and (np.diff(off) >= 0).all()
    f = v.functions
    # (the order of the functions inside the arrays is the order in which they were lowered, not definition order: sort by base)
    for name, base, total in (("n_vars", "var_base", len(v.variables)), ("n_instrs", "instr_base", len(v.instrs)), ("n_bbs", "bb_base", c["bbs"])):
        body = f[f["n_bbs"] > 0]
        o = np.argsort(body[base], kind="stable")
        b, n = body[base][o].astype(np.int64), body[name][o].astype(np.int64)
        assert b[0] == 0 and (b[1:] == (b + n)[:-1]).all() and b[-1] + n[-1] == total, name
    # blocks and instructions are placed together: sorted by block base the instruction bases ascend too
    o = np.argsort(f["bb_base"], kind="stable")
    assert (np.diff(f["instr_base"][o].astype(np.int64)) >= 0).all()
    # every operand of every instruction is a variable of its own function; every jump target one of its blocks
    for i in range(0, len(v), max(1, len(v) // 50)):
        fv = v.function(i)
        nv, nb = len(fv.variables), len(fv.bbs)
        for bb in fv.bbs:
            assert len(bb) > 0
            for ins in bb:
                op = int(ins["opcode"])
                if op in (V.I_LOAD, V.I_NOT, V.I_NEGATE, V.I_CAST):
                    assert int(ins["dst"]) < nv and int(ins["a"]) < nv
                if op == V.I_JUMP:
                    assert int(ins["c"]) < nb
                if op == V.I_BRANCH:
                    assert int(ins["a"]) < nv and int(ins["b"]) < nb and int(ins["c"]) < nb
                if op == V.I_CALL:
                    assert all(int(x) < nv for x in fv.call_args(ins))
    # definitions are in file order and each root node's span starts where its definition does
    lows = v.nodes["low"][v.definitions]
    assert (np.diff(lows.astype(np.int64)) > 0).all()
    assert v.text() == O.dump(src, O.IR)


def test_get_lvalue_type_is_the_pointee(fe):
    src = b"f(x int32) int32\n\tvar y int32 = x\n\ty = y + 1\n\treturn y\n"
    u = fe.get_definitions(src)
    assert u.compile_source() == 0
    fv = u.ir_view().function(0)
    allocs = [int(i["dst"]) for bb in fv.bbs for i in bb if int(i["opcode"]) == V.I_ALLOCATE]
    assert allocs, "locals are allocated lvalues (generate_ir.rs:100-116)"
    for a in allocs:
        assert u.ir_view().type_text(fv.get_lvalue_type(a)) == "int32"
    with pytest.raises(ValueError):
        loads = [int(i["dst"]) for bb in fv.bbs for i in bb if int(i["opcode"]) == V.I_LOAD]
        fv.get_lvalue_type(loads[0])


def test_device_pointers_and_states(fe):
    u = fe.get_definitions(b"f() int32\n\treturn 1\n")
    # before lowering: front-half arrays only
    assert u.array_device(V.A_TOKENS)[1] == u.counts()["tokens"] and u.array_device(V.A_TOKENS)[2] == 16
    assert u.array_device(V.A_NODES)[2] == 24 and u.array_device(V.A_INSTRS)[1] == 0 and u.array_device(V.A_INSTRS)[0] == 0
    assert u.compile_source() == 0
    p, n, es = u.array_device(V.A_INSTRS)
    assert p != 0 and n == u.counts()["instrs"] and es == 24
    assert u.array_device(V.A_VARIABLES)[2] == 8
    # a unit whose finalize reported errors has no IR arrays (frontend.rs:145)
    bad = fe.get_definitions(b"var g nosuchtype\nf() int32\n\treturn 1\n")
    assert bad.compile_source() == 7
    assert bad.array_device(V.A_INSTRS)[1] == 0 and len(bad.array(V.A_FUNCTIONS)) == 0
    assert len(bad.array(V.A_NODES)) == bad.counts()["nodes"]


def test_semantic_error_array_matches_the_dump(fe):
    src = b"f() int32\n\treturn nosuch\n\ng() int32\n\tbreak\n\treturn 2\n"
    u = fe.get_definitions(src)
    assert u.compile_source() == 0
    errs = u.array(V.A_SEM_ERRORS)
    want = O.dump(src, O.SEM).splitlines()
    assert len(errs) == len(want) > 0
    for e, line in zip(errs, want):
        _, low, ln = line.split(" ")
        assert int(e["low"]) == int(low) and int(e["len"]) == int(ln)


def test_unit_from_arrays_round_trip(fe):
    """opusfe_unit_from_arrays is the inverse of opusfe_array for the six parse-side arrays: the rebuilt unit lowers to the same
    IR and joins like the original"""
    src = O.generate(32, 200_000)
    u = fe.get_definitions(src)
    assert u.status == 0
    arrs = [u.array_device(w)[:2] for w in (V.A_SOURCE, V.A_TOKENS, V.A_VAL64, V.A_STRINGS, V.A_NODES, V.A_DEFINITIONS)]
    v = fe.unit_from_arrays(arrs)
    assert v.status == 0 and v.counts()["nodes"] == u.counts()["nodes"] and v.counts()["definitions"] == u.counts()["definitions"]
    assert v.dump_ast() == u.dump_ast() == O.dump(src, O.AST)
    assert v.compile_source() == 0 and v.dump_ir() == O.dump(src, O.IR)
    with pytest.raises(Exception):
        fe.unit_from_arrays([(0, 5)] + arrs[1:])       # a count without a pointer is an argument error


def test_lowering_stats_and_memory_held(fe):
    """opusfe_lowering_stats / opusfe_memory_held: every function with a body is lowered by exactly one of the two paths, the
    node-parallel kernels take (nearly) all of a generated source, and the context's device footprint is reported"""
    src = O.generate(33, 400_000)
    u = fe.get_definitions(src)
    assert u.lowering_stats() == (0, 0)                 # nothing lowered yet
    assert u.compile_source() == 0
    par, seq = u.lowering_stats()
    f = u.array(V.A_FUNCTIONS)
    assert par + seq == int((f["n_bbs"] > 0).sum()) > 0
    assert par >= 0.95 * (par + seq)
    held = fe.memory_held()       # (live results + blocks cached by whatever ran on this context before: only a lower bound here)
    assert held >= len(src) + 16 * u.counts()["tokens"] + 24 * u.counts()["nodes"] + 24 * u.counts()["instrs"]
    import opus_amd
    fresh = opus_amd.Context(0)
    try:
        w = fresh.get_definitions(src)
        assert w.compile_source() == 0
        assert len(src) < fresh.memory_held() < 400 * len(src)
        w.close()
    finally:
        fresh.close()
    # a function beyond the node-parallel kernels' size limit goes to the sequential machine, with the same IR
    big = b"f(x int32) int32\n\tvar y int32 = x\n" + b"".join(b"\ty = y * %d + x\n" % (i % 97) for i in range(600)) + b"\treturn y\n"
    v = fe.get_definitions(big)
    assert v.compile_source() == 0 and v.lowering_stats() == (0, 1)
    assert v.dump_ir() == O.dump(big, O.IR)


def test_flat_lowering_lays_functions_out_in_definition_order_and_reproducibly(fe):
    """opusfe.h LAYOUT: the functions of the flat lowering tile one range of every dense array in definition order (a scan of their
    sizes, no atomics), so two lowerings of one unit give byte-identical arrays when nothing takes the sequential machine"""
    src = O.generate(41, 600_000)
    u = fe.get_definitions(src)
    assert u.compile_source() == 0
    par, seq = u.lowering_stats()
    assert seq == 0 and par > 100
    which = (V.A_FUNCTIONS, V.A_VARIABLES, V.A_BB_OFFSETS, V.A_INSTRS, V.A_CALL_ARGS, V.A_TYPES)
    first = [u.array(w).tobytes() for w in which]
    f = u.array(V.A_FUNCTIONS)
    body = f[f["n_bbs"] > 0]
    for base, n in (("var_base", "n_vars"), ("instr_base", "n_instrs"), ("bb_base", "n_bbs"), ("type_base", "n_types"), ("call_arg_base", "n_call_args")):
        b, k = body[base].astype(np.int64), body[n].astype(np.int64)
        assert b[0] == 0 and (b[1:] == (b + k)[:-1]).all(), base      # definition order: every function starts where the one before it ends
    assert u.compile_source() == 0
    assert [u.array(w).tobytes() for w in which] == first
    w = fe.get_definitions(src)
    assert w.compile_source() == 0
    assert [w.array(x).tobytes() for x in which] == first


def test_parent_array(fe):
    """OPUSFE_A_PARENTS: the upward link of every node, built on request; the inverse of the first_child / next chains"""
    src = O.generate(34, 300_000)
    u = fe.get_definitions(src)
    assert u.status == 0
    nodes, defs, par = u.array(V.A_NODES), u.array(V.A_DEFINITIONS), u.array(V.A_PARENTS)
    assert len(par) == len(nodes) == u.counts()["nodes"]
    roots = np.zeros(len(nodes), dtype=bool)
    roots[defs] = True
    assert ((par == V.NONE) == roots).all()                       # exactly the definition roots have no parent
    # rebuild the parent array on the host from the child chains
    want = np.full(len(nodes), V.NONE, dtype=np.uint32)
    fc, nx = nodes["first_child"], nodes["next"]
    for p in range(len(nodes)):
        c = int(fc[p])
        while c != V.NONE:
            want[c] = p
            c = int(nx[c])
    assert (par == want).all()
    assert u.array_device(V.A_PARENTS)[0] == u.array_device(V.A_PARENTS)[0] != 0       # built once, same device array
    # a unit that did not parse has none
    bad = fe.get_definitions(b"f( int32\n")
    assert bad.status != 0 and len(bad.array(V.A_PARENTS)) == 0
