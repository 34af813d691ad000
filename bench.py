#!/usr/bin/env python3
"""bench.py — source MB/s of the opusfe hot path (lex + parse + IR generation) on N MI355X.

Contract (driver):  python bench.py --gpus N --steps K --warmup W     (N>1: launched by torch.distributed.run)
One "step" = one pass of the hot path over one synthetic source that is already resident in HBM:
  opusfe_parse_device (lexer kernels + parser kernels) followed by opusfe_lower (tables + IR kernels),
  all result arrays left complete in HBM.  Result buffers are released between steps (the library's caching
  allocator keeps the blocks, so steady-state steps do not call hipMalloc).

Workload (config.workload):
  N=1 ("c3"): BASELINE.json configs[2] — ONE synthetic source of --bytes (default 1e9) produced by the seeded
        generator G (tools/gen), with the three depth-1000 regions injected.
  N>1 ("c4", default): BASELINE.json configs[3] — ONE logical source of N x --bytes cut at ARBITRARY byte positions
        (rank r holds bytes [r*B+12345, (r+1)*B+12345): cuts fall inside lines, strings and blocks); the ranks
        exchange ONE all-gather of 64-byte carries (lexer context function, token tails, bracket depth, cut), the head and
        prefix halos point to point, and the definition headers by all-gather (opus_amd/shard.py); each lowers the definitions it owns.  Weak scaling:
        `value` = N x --bytes / max-over-ranks time.  `--workload c3` at N>1 gives independent sources per rank
        (no collective at all).

  "c5" (--workload c5): BASELINE.json configs[4] — a corpus of independent files (sizes log-uniform in [1 KB, 64 KB],
        G(seed=5e6+i), every 100th file with one injected syntax fault), dealt to the ranks by cumulative bytes, NO collective;
        --files sets the corpus size (default 6000 files, about 90 MB, so that the default run finishes in minutes; the
        configuration names 100000).  The files are host buffers handed to opusfe_parse_group (ONE packed pass for the whole
        list: one launch per stage, not per file): the rate includes packing them and the H2D copy.

Extra objects on the JSON line:
  roofline     — the dominant kernel (largest share of HIP-event time, measured by the library on its own
                 stream with hipEventRecord around every launch while the timed steps run)
  cpu_baseline — the C++ CPU restatement (oracle, "port") timed on rank 0's host cores on a bounded sample
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E, /opt/skills/guides/MI355X_MICROARCH.md

# kernel name -> (stage, which compulsory bytes it is priced against)
#   lex   kernels read the source once and write the token records:  S + 16 T
#   parse kernels read tokens and write node records:                16 T + 24 N
#   lower kernels read nodes and write IR records:                   24 N + 24 I + 8 V
STAGE_OF = {"lex": ("lex_", "tok_", "scan_cnt3", "scan_depth", "scan_indent", "scan_tokoff", "scan_seg"),
            "parse": ("parse_", "ast_"), "lower": ("sem_", "lower_", "ir_")}


def lib_sha1():
    """identity of the library a profile was taken with: roofline.traffic is only reported for the very same build"""
    import hashlib
    import opus_amd
    try:
        return hashlib.sha1(open(opus_amd.library_path(), "rb").read()).hexdigest()
    except OSError:
        return None


def stage_of(name):
    for st, prefixes in STAGE_OF.items():
        if name.startswith(prefixes):
            return st
    return "other"


def extra_c2(ctx, opus_amd, O, torch):
    """BASELINE configs[1]: G(seed=2), 10^6 B, resident in HBM: latency of one pass, and the three dumps against the oracle's"""
    import hashlib
    src = O.generate(2, 1_000_000, flags=0)
    d = torch.empty(len(src) + 64, dtype=torch.uint8, device="cuda")
    d[:len(src)].copy_(torch.frombuffer(bytearray(src), dtype=torch.uint8))
    torch.cuda.synchronize()

    def one(keep=False):
        u = ctx.get_definitions_device(d.data_ptr(), len(src))
        if u.status == opus_amd.OK:
            u.compile_source()
        if keep:
            return u
        u.close()
    for _ in range(3):
        one()
    n = 20
    t0 = time.perf_counter()
    for _ in range(n):
        one()
    dt = (time.perf_counter() - t0) / n
    u = one(keep=True)
    sha = lambda t: hashlib.sha1(t.encode("utf-8", errors="surrogateescape")).hexdigest()
    same = u.status == opus_amd.OK and all(sha(g) == sha(O.dump(src, w)) for g, w in ((u.dump_tokens(), O.TOKENS), (u.dump_ast(), O.AST), (u.dump_ir(), O.IR)))
    u.close()
    tcpu, _ = O.time_run(src, lower=True)
    return {"workload": "C2: G(seed=2), 1000000 B, one source resident in HBM", "us_per_pass": round(dt * 1e6, 1), "mbs": round(len(src) / 1e6 / dt, 2),
            "dumps_equal_oracle": bool(same), "compared": ["tokens", "ast", "ir"], "cpu_port_mbs": round(len(src) / 1e6 / tcpu, 2), "cpu_cores": 1}


def extra_c5(ctx, opus_amd, O, n_files):
    """BASELINE configs[4] at reduced corpus size: n_files files of 1-64 KB, every 100th with one injected fault, host buffers"""
    import hashlib
    import math
    import random
    from concurrent.futures import ThreadPoolExecutor
    rnd = random.Random(5)
    sizes = [int(math.exp(rnd.uniform(math.log(1024), math.log(65536)))) for _ in range(n_files)]
    cores = max(1, min(16, os.cpu_count() or 1))

    def make_file(i):
        s = O.generate(5_000_000 + i, sizes[i], flags=0, threads=1)
        if i % 100 == 7:
            s = O.inject(s, i, (i // 100) % 10) or s
        return s
    with ThreadPoolExecutor(max_workers=cores) as ex:
        corpus = list(ex.map(make_file, range(n_files)))
    total = sum(len(f) for f in corpus)
    run_corpus(ctx, opus_amd, corpus)["close"]()          # warm-up (allocator)
    t0 = time.perf_counter()
    res = run_corpus(ctx, opus_amd, corpus)
    dt = time.perf_counter() - t0
    # CPU: one oracle worker per hardware thread on a bounded prefix of the same corpus
    acc, sample = 0, []
    for f in corpus:
        if acc >= 6e6 * cores * 1.5:
            break
        sample.append(f); acc += len(f)
    t0 = time.perf_counter()
    with ThreadPoolExecutor(max_workers=cores) as ex:
        list(ex.map(lambda f: O.time_run(f, lower=True)[1].status, sample))
    tc = time.perf_counter() - t0
    # parity on a sample: every faulty file among the first 1000 and every 10th clean one
    ok, nbad, checked = True, 0, 0
    for i, f in enumerate(corpus[:1000]):
        if i % 100 != 7 and i % 10 != 3:
            continue
        checked += 1
        want = O.dump(f, O.ERROR)
        nbad += want != "Ok\n"
        got_err, got_ir = res["error_text"](i), None
        if got_err != want:
            ok = False
        elif want == "Ok\n":
            got_ir = res["ir_text"](i)
            if hashlib.sha1(got_ir.encode()).digest() != hashlib.sha1(O.dump(f, O.IR).encode()).digest():
                ok = False
    res["close"]()
    return {"workload": f"C5: {n_files} files of 1-64 KB ({total} B), 1 % with one injected syntax fault, host buffers (H2D included)",
            "path": res["path"], "mbs": round(total / 1e6 / dt, 2), "files_per_s": round(n_files / dt, 1), "files_with_syntax_error": res["n_syntax"],
            "files_on_packed_path": res["n_packed"], "files_on_single_path": res["n_single"],
            "cpu_port_mbs": round(acc / 1e6 / tc, 2), "cpu_cores": cores, "cpu_sample_files": len(sample),
            "sample_parity": {"files_checked": checked, "with_syntax_error": nbad, "error_tuples_and_ir_text_match": bool(ok)}}


def run_corpus(ctx, opus_amd, corpus):
    """one pass of the corpus path (opusfe_parse_group: the whole list as one packed pass); returns accessors for the parity sample"""
    g = ctx.compile_group(corpus, lower=True)
    statuses = g.statuses()
    for st in statuses:
        if st not in (opus_amd.OK, opus_amd.SYNTAX):
            raise SystemExit(f"corpus file failed: status {st}: {ctx.message()}")
    n_packed, n_single, n_syntax = g.stats()
    return {"path": "opusfe_parse_group: all files packed into one unit, one launch per stage; irregular files alone through the single-source pipeline",
            "n_syntax": n_syntax, "n_packed": n_packed, "n_single": n_single,
            "error_text": lambda i: g.dump_error(i), "ir_text": lambda i: g.dump_ir(i), "close": g.close}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--bytes", type=float, default=1e9, help="source bytes per GPU")
    ap.add_argument("--seed", type=int, default=3)
    ap.add_argument("--stage", default="ir", choices=["lex", "parse", "ir"], help="deepest stage to run (default: whole path)")
    ap.add_argument("--files", type=int, default=6000, help="c5: number of files of the corpus (whole job)")
    ap.add_argument("--workload", default=None, choices=["c3", "c4", "c5"],
                    help="c3: one source per GPU (default at N=1; at N>1 independent shards, no collective); "
                         "c4: ONE logical source of N x bytes cut at arbitrary byte positions, carries exchanged by all-gather (default at N>1)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"], help="gloo lets several ranks share one GPU (rehearsal only)")
    ap.add_argument("--cpu-sample-bytes", type=float, default=48e6)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the c2 / c5 legs of the default N=1 run")
    ap.add_argument("--extra-files", type=int, default=6000, help="files of the c5 leg of the default N=1 run")
    ap.add_argument("--gen-flags", type=int, default=1, help="generator flags (bit 0: inject the three depth-1000 regions of BASELINE configs[2]); diagnostics only")
    args = ap.parse_args()
    import torch
    import opus_amd
    import oracle_lib as O   # generator (and, on rank 0 at N=1, the oracle as the timed CPU baseline)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...")
    dist = None
    local_rank = local_rank % max(1, torch.cuda.device_count())
    torch.cuda.set_device(local_rank)
    if world > 1:
        import torch.distributed as dist
        import datetime
        tmo = datetime.timedelta(seconds=300)      # a rank that fails must not leave the others waiting for ever
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank), timeout=tmo)
        else:
            dist.init_process_group(backend="gloo", timeout=tmo)
    workload = args.workload or ("c3" if world == 1 else "c4")

    nbytes = int(args.bytes)
    threads = min(16, os.cpu_count() or 8)
    t0 = time.time()
    # N ranks generate at the same time: share the host's cores (os.cpu_count() is the whole machine on the GPU boxes)
    gen_threads = max(2, min(16, (os.cpu_count() or 16) // world)) if world > 1 else threads
    corpus = None
    if workload == "c5":
        import math, random
        from concurrent.futures import ThreadPoolExecutor
        from opus_amd import sharding as _sh5
        rnd = random.Random(5)
        sizes_all = [int(math.exp(rnd.uniform(math.log(1024), math.log(65536)))) for _ in range(args.files)]
        mine_idx = _sh5.shard_by_bytes(sizes_all, rank, world)

        def make_file(i):
            s = O.generate(5_000_000 + i, sizes_all[i], flags=0, threads=1)
            if i % 100 == 7:                       # 1 % of the files carry one injected fault
                bad = O.inject(s, i, (i // 100) % 10)
                s = bad or s
            return s
        with ThreadPoolExecutor(max_workers=gen_threads) as ex:
            corpus = list(ex.map(make_file, mine_idx))
        src = b""
        nbytes = sum(len(f) for f in corpus)
    elif workload == "c3":
        src = O.generate(args.seed + 1000 * rank, nbytes, flags=args.gen_flags, threads=gen_threads)
    else:
        # rank r holds bytes [r*B + shift, (r+1)*B + shift) of ONE generated source of world*B bytes (rank 0 from byte 0):
        # the cuts fall in the middle of lines, strings and blocks on purpose
        nblk = 16
        blk = nbytes // nblk
        nbytes = blk * nblk
        shift = 12345 if world > 1 else 0
        mine = O.generate_range(args.seed, rank * nblk, nblk, blk, flags=1, threads=gen_threads)
        nxt = O.generate_range(args.seed, (rank + 1) * nblk, 1, blk, flags=1, threads=1)[:shift] if rank + 1 < world else b""
        src = (mine if rank == 0 else mine[shift:]) + nxt
    gen_s = time.time() - t0
    n_local = len(src)
    d_src = torch.empty(n_local + 64, dtype=torch.uint8, device="cuda")
    if n_local:
        d_src[:n_local].copy_(torch.frombuffer(bytearray(src), dtype=torch.uint8))
    torch.cuda.synchronize()

    ctx = opus_amd.Context(local_rank)
    gather = None
    if workload == "c4" and world > 1:
        from opus_amd import shard, sharding
        shard_times = {"comm_s": 0.0, "header_reparse_s": 0.0, "header_bytes": 0, "fast": 0, "slow": 0}

        class TimedComm(sharding.DistComm):      # (host time of the exchange steps, waiting for the slowest rank included)
            def _timed(self, f, *a):
                t = time.perf_counter()
                r = f(*a)
                shard_times["comm_s"] += time.perf_counter() - t
                return r

            def gather(self, name, values):
                return self._timed(super().gather, name, values)

            def gather_sized(self, name, values, sizes):
                return self._timed(super().gather_sized, name, values, sizes)

            def p2p(self, name, sends, recvs, ranks):
                return self._timed(super().p2p, name, sends, recvs, ranks)
        gather = TimedComm(dist, torch.device("cuda", local_rank))

    def step_c4():
        st = shard.RankState(ctx, d_src[:n_local], rank, world)
        shard.plan([st], gather)
        shard_times["fast" if st.speculation_held else "slow"] += 1
        res = shard.compile_sharded([st], gather, lower=(args.stage == "ir"))[0]
        if res["status"] != opus_amd.OK:
            raise SystemExit(f"rank {rank}: sharded hot path failed: status {res['status']} (first bad rank {res['first_bad_rank']}): {ctx.message()}")
        c = res["unit"].counts()
        own = res["own_counts"]
        shard_times["header_reparse_s"] += res["header_reparse_s"]
        shard_times["header_bytes"] = res["header_bytes"]
        for k in ("source_bytes", "tokens", "nodes"):
            c[k] = own[k]                  # tokens / nodes of the bytes this rank owns (the unit also holds the other ranks' headers)
        c["definitions_own"] = own["definitions"]
        res["unit"].close()
        return c

    lowering = {}

    def step_c5():
        g = ctx.compile_group(corpus, lower=(args.stage == "ir"))
        nbad = 0
        for st in g.statuses():
            if st not in (opus_amd.OK, opus_amd.SYNTAX):
                raise SystemExit(f"rank {rank}: corpus file failed: status {st}: {ctx.message()}")
            nbad += st == opus_amd.SYNTAX
        n_packed, n_single, _ = g.stats()
        tot = dict(g.unit().counts())          # the packed unit (files on the single-file path are not in these totals)
        tot["files"], tot["files_with_syntax_error"], tot["files_on_packed_path"], tot["files_on_single_path"] = len(corpus), nbad, n_packed, n_single
        g.close()
        return tot

    def step():
        if corpus is not None:
            return step_c5()
        if gather is not None:
            return step_c4()
        if args.stage == "lex":
            u = ctx.lex_device(d_src.data_ptr(), n_local)
        else:
            u = ctx.get_definitions_device(d_src.data_ptr(), n_local)
        if u.status != opus_amd.OK:
            raise SystemExit(f"rank {rank}: hot path failed: status {u.status}: {ctx.message()} {u.dump_error() if u.status == 1 else ''}")
        if args.stage == "ir":
            rc = u.compile_source()
            if rc != opus_amd.OK:
                raise SystemExit(f"rank {rank}: lowering failed: status {rc}: {ctx.message()}")
            lowering["node_parallel"], lowering["sequential_machine"] = u.lowering_stats()
        c = u.counts()
        u.close()
        return c

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    counts = None
    for _ in range(args.warmup):
        counts = step()
    ctx.set_profiling(True)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        counts = step()
    barrier()
    dt = time.perf_counter() - t0
    kstats = ctx.kernel_stats()
    ctx.set_profiling(False)
    # k_flat_pre is ONE kernel launched once per size class with differently sized LDS: one entry with that many launches per step.
    # The lowering is a chain of passes over the same nodes (lower_par.h: k_flat_pre, k_sched_*, k_flat_count<class> per round,
    # k_flat_scan, k_flat_rv, k_flat_emit<class>); none of them lowers anything alone, so the stage is ALSO priced as one unit,
    # "lower_flat": the sum of its passes' HIP-event times, one "launch" per step, against the stage's algorithmic bytes.
    pre = [k for k in kstats if k[0].startswith("lower_pre_")]
    if pre and not os.environ.get("OPUSFE_BENCH_RAW_KERNELS"):      # (diagnostics: keep the class launches apart)
        kstats = [k for k in kstats if not k[0].startswith("lower_pre_")] + [("lower_pre", sum(k[1] for k in pre), sum(k[2] for k in pre))]
    FLAT_PASSES = ("lower_pre", "sched_hist", "sched_scan", "sched_scatter", "lower_count", "lower_scan", "lower_scan_funcs", "lower_reserve", "lower_bases",
                   "lower_rv", "lower_emit", "flat_check")
    flat = [k for k in kstats if k[0] in FLAT_PASSES]
    flat_passes = {k[0]: round(k[1] / args.steps, 3) for k in flat}
    if flat:
        kstats = kstats + [("lower_flat", sum(k[1] for k in flat), args.steps)]
    held = ctx.memory_held()
    mem = ctx.memory_stats()      # live results / cached blocks / temporaries arena / high-water mark
    total_bytes_all = nbytes
    if dist is not None and workload == "c5":
        from opus_amd import sharding as _sh6
        total_bytes_all = int(_sh6.all_reduce_sum(nbytes, dist))
    sharded_totals = None
    if dist is not None and workload == "c4":
        steps_run = max(1, args.steps + args.warmup)
        counts["comm_ms_x1000"] = int(1e6 * shard_times["comm_s"] / steps_run)
        counts["header_reparse_ms_x1000"] = int(1e6 * shard_times["header_reparse_s"] / steps_run)
        # what the ranks produced together: comparable with the same logical source lexed / parsed / lowered as ONE unit
        # (tests/test_shard_gpu.py::test_two_process_gloo_run_matches_the_whole_source)
        from opus_amd import sharding as _sh7
        sharded_totals = {k: int(_sh7.all_reduce_sum(counts[k], dist)) for k in ("source_bytes", "tokens", "nodes", "definitions_own", "instrs", "vars")}
        # per step, mean over ranks: the carry / halo / header exchanges (host clock, waiting included) and the re-parse of the other ranks' headers
        sharded_totals["comm_ms"] = round(_sh7.all_reduce_sum(counts["comm_ms_x1000"], dist) / 1000.0 / world, 3)
        sharded_totals["header_reparse_ms"] = round(_sh7.all_reduce_sum(counts["header_reparse_ms_x1000"], dist) / 1000.0 / world, 3)
        sharded_totals["header_bytes_rank0"] = shard_times["header_bytes"]
        # exchange steps per pass: all-gathers and point-to-point messages of rank 0 (ONE carry all-gather when every shard
        # starts in code context at depth 0 — `single_all_gather_passes` counts the passes where that held)
        # the exchanges alone (no waiting for a slower rank): every pattern of one pass repeated behind a barrier, with the sizes of the last pass
        hb = int(_sh7.all_reduce_max(float(max(1, shard_times["header_bytes"] // max(1, world - 1))), dist))      # one size for every rank
        probe_t = torch.zeros(hb, dtype=torch.uint8, device="cuda")
        halo = torch.zeros(4096, dtype=torch.uint8, device="cuda")
        reps, acc = 5, 0.0
        for _ in range(reps):
            dist.barrier()
            t = time.perf_counter()
            sharding.DistComm.gather(gather, "probe", [(1, 2, 3, 4, 5, 6, 7, 8)])
            sharding.DistComm.gather(gather, "probe", [(0, hb)])
            sharding.DistComm.gather_sized(gather, "probe", [probe_t], [hb] * world)
            sharding.DistComm.p2p(gather, "heads", [[(rank - 1, halo)] if rank > 0 else []], [[(rank + 1, None)] if rank + 1 < world else []], [rank])
            sharding.DistComm.p2p(gather, "prefixes", [[(rank - 1, halo)] if rank > 0 else []], [[(rank + 1, 4096)] if rank + 1 < world else []], [rank])
            acc += time.perf_counter() - t
        counts["exchange_only_ms_x1000"] = int(1e6 * acc / reps)
        sharded_totals["exchange_only_ms"] = round(_sh7.all_reduce_sum(counts["exchange_only_ms_x1000"], dist) / 1000.0 / world, 3)
        sharded_totals["collectives_per_pass"] = round((gather.collectives - 3 * reps) / steps_run, 2)
        sharded_totals["p2p_messages_per_pass_rank0"] = round((gather.p2p_messages - (2 * reps if rank > 0 else 0)) / steps_run, 2)
        sharded_totals["single_all_gather_passes"], sharded_totals["dependent_chain_passes"] = shard_times["fast"], shard_times["slow"]
    if dist is not None:
        from opus_amd import sharding as _sh
        dt = _sh.all_reduce_max(dt, dist)

    S, T, N = counts["source_bytes"], counts["tokens"], counts["nodes"]
    I, V = counts["instrs"], counts["vars"]

    out = None
    if rank == 0:
        ms_per_step = dt / args.steps * 1e3
        value = (total_bytes_all if workload == "c5" else world * nbytes) / 1e6 / (dt / args.steps)
        kstats.sort(key=lambda k: -k[1])
        # ---- CPU baseline (the oracle, "port") and the oracle as checker of the measured path, on bounded samples ----
        cpu = parity = None
        node_ratio = None        # reference-tree nodes per flat-AST node (the flat AST has list / pair carrier nodes the reference's boxed tree lacks)
        if world == 1 and not args.no_cpu and workload != "c5":
            sample = src[:int(min(args.cpu_sample_bytes, nbytes))]
            # cut the sample at a top-level definition boundary (a line that starts in column 0 after a newline)
            cut = len(sample)
            while cut > 0 and not (sample[cut - 1:cut] == b"\n" and cut < len(src) and src[cut:cut + 1] not in (b"\t", b"\n", b" ", b";")):
                cut -= 1
            sample = sample[:cut] if cut > 0 else sample
            tcpu, c2 = O.time_run(sample, lower=(args.stage == "ir"))
            # the oracle as checker: the GPU path on the same sample must produce the same element counts
            us = ctx.get_definitions(sample)
            if us.status == opus_amd.OK and args.stage == "ir":
                us.compile_source()
            cs = us.counts()
            pairs = [("tokens", c2.tokens), ("definitions", c2.definitions)] if args.stage != "lex" else []
            if args.stage == "ir":
                pairs += [("instrs", c2.instrs), ("vars", c2.vars)]
            parity = {"sample_bytes": len(sample), "oracle_status": c2.status, "gpu_status": us.status,
                      "counts_match": all(cs[k] == v for k, v in pairs) and (us.status == 0) == (c2.status in (0, 3)),      # (oracle 3: IR generated, with diagnostics)
                      "compared": [k for k, _ in pairs]}
            if cs["nodes"]:
                node_ratio = c2.nodes / cs["nodes"]
            us.close()
            # the reference's other single-thread cases, C1 (10k lines) and C2 (1 MB)
            c1 = O.generate_lines(1, 10000)
            t1, _ = O.time_run(c1, lower=(args.stage == "ir"))
            c2src = O.generate(2, 1_000_000)
            t2, _ = O.time_run(c2src, lower=(args.stage == "ir"))
            cpu = {"value": round(len(sample) / 1e6 / tcpu, 3), "unit": "MB/s", "cores": 1, "kind": "port",
                   "sample": f"first {len(sample)} bytes of the same source (whole definitions), oracle {'lex+parse+IR' if args.stage == 'ir' else 'lex+parse'}, single thread like the reference; status {c2.status}",
                   "c1_10k_lines_mbs": round(len(c1) / 1e6 / t1, 3), "c2_1mb_mbs": round(len(c2src) / 1e6 / t2, 3)}
        elif world == 1 and not args.no_cpu and workload == "c5":
            # files are independent: one oracle worker per hardware thread this process may use (the GPU boxes give 16 to a one-GPU job)
            from concurrent.futures import ThreadPoolExecutor
            import hashlib
            cores = max(1, min(16, os.cpu_count() or 1))
            budget, sample_files, acc = 6e6 * cores * 4, [], 0      # about 4 s of work per worker at the single-thread rate
            for f in corpus:
                if acc >= budget:
                    break
                sample_files.append(f); acc += len(f)
            t0c = time.perf_counter()
            with ThreadPoolExecutor(max_workers=cores) as ex:
                res = list(ex.map(lambda f: O.time_run(f, lower=(args.stage == "ir"))[1].status, sample_files))
            tc = time.perf_counter() - t0c
            # checker: the first files (every faulty one among them) must give the oracle's error line / IR text
            check = sample_files[:200]
            g = ctx.compile_group(check, lower=(args.stage == "ir"))
            ok, nerr = True, 0
            for i, f in enumerate(check):
                want_err = O.dump(f, O.ERROR)
                nerr += want_err != "Ok\n"
                if g.dump_error(i) != want_err:
                    ok = False
                elif want_err == "Ok\n" and args.stage == "ir":
                    if g.status(i) != opus_amd.OK or hashlib.sha1(g.dump_ir(i).encode()).digest() != hashlib.sha1(O.dump(f, O.IR).encode()).digest():
                        ok = False
            g.close()
            parity = {"files_checked": len(check), "files_with_syntax_error": nerr, "error_tuples_and_ir_text_match": ok}
            cpu = {"value": round(acc / 1e6 / tc, 3), "unit": "MB/s", "cores": cores, "kind": "port",
                   "sample": f"{len(sample_files)} files ({int(acc)} bytes) of the same corpus, one oracle worker per hardware thread ({cores} threads), statuses {sorted(set(res))}"}
        # ---- the small-input regimes on the same line (BASELINE configs[1] and configs[4]), bounded to a few seconds each ----
        extra = None
        if world == 1 and workload == "c3" and not args.no_cpu and not args.no_extra and args.stage == "ir":
            extra = {"c2": extra_c2(ctx, opus_amd, O, torch), "c5": extra_c5(ctx, opus_amd, O, args.extra_files)}
        # ---- algorithmic bytes: S + 16 T + 24 N + 24 I + 8 V with N = nodes of the REFERENCE's tree ----
        N_ref = int(round(N * node_ratio)) if node_ratio else N
        alg = {"lex": S + 16 * T, "parse": 16 * T + 24 * N_ref, "lower": 24 * N_ref + 24 * I + 8 * V, "other": S}
        total_alg = S + 16 * T + 24 * N_ref + 24 * I + 8 * V
        roofline = None
        if kstats:
            name, ms, launches = kstats[0]
            avg_ms = ms / max(1, launches)
            per_step = launches / args.steps
            a = alg[stage_of(name)] / max(1.0, per_step) / (avg_ms * 1e-3) / 1e9
            # HBM traffic per launch: PMC counters cannot be read from inside this process; the committed rocprofv3 passes
            # (tools/profile.sh -> profiles/*_traffic.json) hold FETCH_SIZE + WRITE_SIZE per launch for one source size:
            # reported only when this run is that same workload on that same library build, else null
            traffic = None
            try:
                tdir = os.path.join(ROOT, "profiles")
                for fn in sorted(os.listdir(tdir), reverse=True):
                    if fn.endswith("_traffic.json"):
                        tj = json.load(open(os.path.join(tdir, fn)))
                        if workload == "c3" and tj.get("source_bytes") == nbytes and name in tj["kernels"] and tj.get("library_sha1") == lib_sha1():
                            traffic = int(tj["kernels"][name]["fetch"] + tj["kernels"][name]["write"])
                            break
            except OSError:
                pass
            roofline = {"bound": "hbm", "kernel": name, "achieved": round(a, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": round(a / HBM_PEAK_GBS, 5), "traffic": traffic, "avg_ms": round(avg_ms, 4),
                        "launches_per_step": per_step, "algorithmic_bytes_per_launch": int(alg[stage_of(name)] / max(1.0, per_step)),
                        **({"passes_ms_per_step": flat_passes, "note": "the lowering stage as one unit: the sum of its passes (lower_par.h), priced against 24 N + 24 I + 8 V"} if name == "lower_flat" else {}),
                        "nodes_priced": "reference tree (oracle / GPU node ratio of the checked sample)" if node_ratio else "flat AST (no CPU leg ran: a few % high)"}
        stage_name = {"lex": "lex only (PARTIAL path)", "parse": "lex+parse (PARTIAL path)", "ir": "lex+parse+IR"}[args.stage]
        if workload == "c3":
            wl = f"C3: one synthetic source of {nbytes} B per GPU, generator G(seed={args.seed}), depth-1000 regions, {world} GPU(s)"
            shd = "none" if world == 1 else "independent shards, no collective"
        elif workload == "c4":
            wl = f"C4: ONE synthetic source of {world * nbytes} B cut at arbitrary bytes over {world} GPUs ({nbytes} B each), carries by all-gather, generator G(seed={args.seed})"
            shd = "byte-range shards; ONE all-gather of the 64-byte carries, head / prefix halos point to point, (status, length) + definition headers by all-gather"
        else:
            wl = f"C5: corpus of {args.files} independent files of 1-64 KB ({total_bytes_all} B), 1 % with one injected syntax fault, dealt to {world} GPU(s) by cumulative bytes; host buffers (H2D included)"
            shd = "files by cumulative bytes, no collective"
        out = {"metric": f"source MB/s {stage_name} (whole node)", "value": round(value, 2), "unit": "MB/s", "n_gpus": world,
               "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3), "higher_is_better": True,
               "scaling": "weak" if workload != "c5" else "strong", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
               "config": {"workload": wl, "source_bytes_per_gpu": nbytes, "tokens": T, "nodes": N, "nodes_reference_tree": N_ref if node_ratio else None,
                          "instrs": I, "vars": V, "functions": counts["functions"], "algorithmic_bytes": total_alg, "sharding": shd,
                          "files": counts.get("files"), "files_with_syntax_error": counts.get("files_with_syntax_error")},
               "hbm_gbs_whole_path": round(world * total_alg / (dt / args.steps) / 1e9, 2),
               "hbm_frac_whole_path": round(total_alg / (dt / args.steps) / 1e9 / HBM_PEAK_GBS, 5),
               "hbm_held_bytes": held, "hbm_peak_bytes": mem["peak"], "hbm_bytes_by_kind": {k: mem[k] for k in ("live", "cached", "arena")}, "lowering_functions": lowering or None, "sharded_totals": sharded_totals,
               "roofline": roofline, "cpu_baseline": cpu, "parity_check": parity if cpu is not None else None, "extra": extra,
               "kernels": [{"name": n, "ms_per_step": round(ms / args.steps, 4), "launches_per_step": l / args.steps} for n, ms, l in kstats[:28]],
               "gen_seconds": round(gen_s, 1)}
    ctx.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if out is not None:
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
