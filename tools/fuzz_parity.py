#!/usr/bin/env python3
"""Differential fuzzing on the GPU box: many small generated sources (some with injected syntax errors) through
opusfe_parse_batch, every result compared with the CPU oracle (IR text, semantic diagnostics, first-error line).
usage: python tools/fuzz_parity.py [--seeds 300] [--first 1000]        exit status 1 on the first mismatch"""
import argparse
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", type=int, default=300)
    ap.add_argument("--first", type=int, default=1000)
    ap.add_argument("--mutants", type=int, default=0, help="also this many ill-typed mutants (tests/mutate.py)")
    args = ap.parse_args()
    import opus_amd
    import oracle_lib as O
    rnd = random.Random(99)
    srcs = []
    for k in range(args.seeds):
        s = O.generate(args.first + k, rnd.randint(2000, 40000), flags=0)
        if k % 5 == 3:
            s = O.inject(s, args.first + k, k % 3)
        if k % 11 == 7:
            s = s[:rnd.randint(1, len(s))]
        srcs.append(s)
    if args.mutants:
        import mutate as M
        srcs += M.mutants(args.mutants, seed0=args.first) + M.byte_mutants(args.mutants, seed0=args.first) + M.trap_mutants(args.mutants, seed0=args.first)
    ctx = opus_amd.Context(0)
    def valid_utf8(b):
        try:
            b.decode("utf-8")
            return True
        except UnicodeDecodeError:
            return False
    # the reference reads its sources with fs::read_to_string (frontend.rs:85): a fault injected into the middle of a multi-byte
    # character is not an input of the path (opus_amd/program.py rejects it like an unreadable file)
    n_all = len(srcs)
    srcs = [s for s in srcs if valid_utf8(s)]
    if len(srcs) != n_all:
        print(f"fuzz: {n_all - len(srcs)} sources that are not UTF-8 left out", flush=True)
    units = ctx.compile_corpus(srcs, lower=True)
    bad = 0
    for k, (s, u) in enumerate(zip(srcs, units)):
        import mutate as M
        w = M.expectation(s)
        want_err = O.dump(s, O.ERROR)
        if w[0] == "unsupported":
            ok = u.status == 2
        elif w[0] == "syntax":
            ok = (u.status == 1 and u.dump_error() == w[1]) or (w[2] and u.status == 2)
        elif w[0] == "semantic":
            ok = u.status == 7 and u._dump(4) == w[1]
        else:
            ok = u.status == 0 and u.dump_ir() == w[1] and u._dump(4) == w[2]
        if not ok:
            bad += 1
            path = os.path.join(ROOT, "gpurun_out", f"fuzz_bad_{args.first + k}.opus")
            os.makedirs(os.path.dirname(path), exist_ok=True)
            open(path, "wb").write(s)
            print(f"MISMATCH seed {args.first + k}: status {u.status}, oracle error line {want_err.strip()!r} -> {path}", flush=True)
        u.close()
    print(f"fuzz: {len(srcs)} sources, {bad} mismatches", flush=True)
    ctx.close()
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
