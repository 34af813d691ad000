#!/usr/bin/env python3
"""CPU: semantic mutants (tests/mutate.py) through the host emulation of the device code (tests/emu, ASan/UBSan) against the
oracle.  usage: python tools/fuzz_semantic_emu.py [--n 2000] [--seed0 0] [--lo 600] [--hi 5000]"""
import argparse
import collections
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=2000)
    ap.add_argument("--seed0", type=int, default=0)
    ap.add_argument("--lo", type=int, default=600)
    ap.add_argument("--hi", type=int, default=5000)
    ap.add_argument("--bytes", action="store_true", help="byte-level mutants (lexer / parser paths) instead of semantic ones")
    args = ap.parse_args()
    import emu_lib as E
    import mutate as M
    bad, kinds = 0, collections.Counter()
    for lo in range(0, args.n, 500):
        cases = (M.byte_mutants if args.bytes else M.mutants)(min(500, args.n - lo), args.seed0 + lo, args.lo, args.hi)
        want = [M.expectation(c) for c in cases]
        errs, irs, sems = E.dump_many(cases, 3), E.dump_many(cases, 2), E.dump_many(cases, 4)
        for k, (c, w, (_, err), (st, ir), (_, sem)) in enumerate(zip(cases, want, errs, irs, sems)):
            kinds[w[0]] += 1
            if w[0] == "unsupported":
                ok = st == 2
            elif w[0] == "syntax":
                ok = (st == 1 and err == w[1]) or (w[2] and st == 2)
            elif w[0] == "semantic":
                ok = st == 7 and sem == w[1]
            else:
                ok = st == 0 and ir == w[1] and sem == w[2]
            if not ok:
                bad += 1
                path = f"/tmp/fuzz_sem_bad_{args.seed0 + lo + k}.opus"
                open(path, "wb").write(c)
                print(f"MISMATCH {args.seed0 + lo + k}: expected {w[0]}, emu status {st} -> {path}", flush=True)
        print(f"{lo + len(cases)} done, {bad} mismatches, {dict(kinds)}", flush=True)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
