import sys, time, random
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import opus_amd, oracle_lib as O
rnd = random.Random(5)
files = [O.generate(9000 + k, rnd.randint(1024, 65536)) for k in range(400)]
tot = sum(map(len, files))
ctx = opus_amd.Context(0)
for keep in (False, True, False):
    for u in [ctx.get_definitions(f) for f in files[:20]]: u.compile_source(); u.close()
    held = []
    t0 = time.perf_counter(); tp = 0.0
    for f in files:
        a = time.perf_counter()
        u = ctx.get_definitions(f)
        tp += time.perf_counter() - a
        u.compile_source()
        if keep: held.append(u)
        else: u.close()
    dt = time.perf_counter() - t0
    print(f"keep={keep}: {dt/len(files)*1e3:.2f} ms/file (parse {tp/len(files)*1e3:.2f}), {tot/1e6/dt:.1f} MB/s", flush=True)
    for u in held: u.close()
ctx.set_profiling(True)
for f in files[:100]:
    u = ctx.get_definitions(f); u.compile_source(); u.close()
ks = ctx.kernel_stats()
tot_ms = sum(k[1] for k in ks) if ks and not isinstance(ks, dict) else None
print(ks if tot_ms is None else sorted(ks, key=lambda k: -k[1])[:12], tot_ms)
