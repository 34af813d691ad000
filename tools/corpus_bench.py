#!/usr/bin/env python3
"""Throughput of opusfe_parse_batch on a corpus of independent small files (BASELINE configs[4] shape, scaled by --files).
usage: python tools/corpus_bench.py [--files 2000] [--min 1024] [--max 65536] [--no-lower]"""
import argparse
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--files", type=int, default=2000)
    ap.add_argument("--min", type=int, default=1024)
    ap.add_argument("--max", type=int, default=65536)
    ap.add_argument("--no-lower", action="store_true")
    args = ap.parse_args()
    import opus_amd
    import oracle_lib as O
    rnd = random.Random(5)
    t0 = time.time()
    big = O.generate(11, max(4 << 20, args.max * 8), flags=0, threads=8)
    # files = whole-definition slices of one generated source (cut at column-0 lines)
    starts = [0] + [i + 1 for i in range(len(big) - 1) if big[i] == 10 and big[i + 1] not in (9, 10, 32, 59)]
    files = []
    for k in range(args.files):
        want = rnd.randint(args.min, args.max)
        a = rnd.randrange(len(starts) - 1)
        b = a
        while b + 1 < len(starts) and starts[b + 1] - starts[a] < want:
            b += 1
        s = big[:starts[1]] * 0 + big[starts[a]:starts[b + 1] if b + 1 < len(starts) else len(big)]
        if k % 100 == 7:
            s = s[:len(s) // 2] + b"\n)) broken ((\n" + s[len(s) // 2:]     # 1 % with a syntax error
        files.append(s)
    total = sum(len(f) for f in files)
    print(f"corpus: {len(files)} files, {total / 1e6:.1f} MB, generated in {time.time() - t0:.1f} s", flush=True)
    ctx = opus_amd.Context(0)
    units = ctx.compile_corpus(files[:64], lower=not args.no_lower)     # warm-up
    for u in units:
        u.close()
    t0 = time.perf_counter()
    units = ctx.compile_corpus(files, lower=not args.no_lower)
    dt = time.perf_counter() - t0
    st = {}
    for u in units:
        st[u.status] = st.get(u.status, 0) + 1
        u.close()
    print(f"batch: {dt:.3f} s  {total / 1e6 / dt:.1f} MB/s  {len(files) / dt:.0f} files/s  statuses {st}", flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
