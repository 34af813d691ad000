#!/usr/bin/env python3
"""Per-file latency of the single-source pipeline on small files (GPU box): where the time of BASELINE config 5 goes.
usage: python tools/perfile_bench.py        prints ms per file and the kernel time shares for three file-size classes"""
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import opus_amd
    import oracle_lib as O
    ctx = opus_amd.Context(0)
    rnd = random.Random(5)
    for lo, hi in ((1024, 2048), (8192, 16384), (49152, 65536)):
        files = [O.generate(9000 + k, rnd.randint(lo, hi)) for k in range(200)]
        tot = sum(map(len, files))
        for f in files[:20]:
            u = ctx.get_definitions(f); u.compile_source(); u.close()
        ctx.set_profiling(False)
        t0 = time.perf_counter()
        for f in files:
            u = ctx.get_definitions(f); u.compile_source(); u.close()
        dt = time.perf_counter() - t0
        ctx.set_profiling(True)
        for f in files:
            u = ctx.get_definitions(f); u.compile_source(); u.close()
        ks = sorted(ctx.kernel_stats(), key=lambda k: -k[1])
        ksum = sum(k[1] for k in ks)
        top = ", ".join(f"{k[0]} {k[1] / len(files):.3f}" for k in ks[:4])
        print(f"files of {lo}-{hi} B: {dt / len(files) * 1e3:.2f} ms per file ({tot / 1e6 / dt:.1f} MB/s), kernels {ksum / len(files):.2f} ms per file: {top}", flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
