"""prints what the packed pass decided for every file of a small group (flags, status, message) and the first difference from
the oracle's per-file answers.  usage: group_debug.py [mutants SEED N | files...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import opus_amd
import oracle_lib as O


def first_diff(tag, got, want):
    if got == want:
        return False
    gl, wl = got.split("\n"), want.split("\n")
    for k in range(min(len(gl), len(wl))):
        if gl[k] != wl[k]:
            print("  ", tag, "first difference at line", k, "\n    got ", gl[max(0, k - 3):k + 3], "\n    want", wl[max(0, k - 3):k + 3])
            break
    print("  ", tag, "lines", len(gl), len(wl))
    return True


def main():
    ctx = opus_amd.Context(0)
    files = [O.generate(11, 4000), O.generate(12, 6000)]
    verbose = True
    if len(sys.argv) > 1 and sys.argv[1] == "mutants":
        import mutate as M
        seed, n = int(sys.argv[2]), int(sys.argv[3])
        files = M.mutants(n, seed0=seed) + M.byte_mutants(n, seed0=seed)
        verbose = False
    elif len(sys.argv) > 1:
        files = [open(p, "rb").read() for p in sys.argv[1:]]
    g = ctx.compile_group(files)
    print("message:", ctx.message())
    print("stats:", g.stats())
    u = g.unit()
    print("unit status", u.status, u.counts())
    nbad = 0
    for f, src in enumerate(files):
        r = g.ranges(f)
        if verbose:
            print(f, "status", g.status(f), "flags", hex(r[9]), "ranges", r[:9])
        want_err = O.dump(src, O.ERROR)
        st = g.status(f)
        bad = False
        if want_err == "Unsupported\n":
            bad = st != 2
        elif want_err != "Ok\n":
            bad = st != 1 or first_diff("error", g.dump_error(f), want_err)
        else:
            want_ir, want_sem = O.dump(src, O.IR), O.dump(src, O.SEM)
            if want_ir == "" and want_sem != "":
                bad = st != 7 or first_diff("sem", g.dump_semantic_errors(f), want_sem)
            elif st != 0:
                bad = True
            else:
                bad = first_diff("tokens", g.dump_tokens(f), O.dump(src, O.TOKENS)) or first_diff("ast", g.dump_ast(f), O.dump(src, O.AST)) or \
                    first_diff("ir", g.dump_ir(f), want_ir) or first_diff("sem", g.dump_semantic_errors(f), want_sem)
        if bad:
            nbad += 1
            print("FILE", f, "status", st, "flags", hex(r[9]), "want_err", want_err.strip(), "single" if g.file_unit(f) else "packed")
            if nbad <= 3:
                open(os.path.join(ROOT, "gpurun_out", f"bad_{f}.opus"), "wb").write(src)
            if nbad > 10:
                break
    print("bad files:", nbad)
    g.close(); ctx.close()


main()
