#!/usr/bin/env python3
"""Benchmark of the batch tag pipeline on MI355X.

One "step" is one pass of the whole hot path (tp_run_lines_device: workspace sizing, the per-file pipeline
kernels, caption-offset scan, caption/histogram emit) over one resident shard of synthetic tag files, plus —
on more than one GPU — the RCCL all-reduce of the tag histogram and counters.  Inputs are in HBM before
the timed region starts.  Weak scaling (the default): every rank owns `--files` files (default 1,000,000: BASELINE
config 2, "1M tag files, ~40 tags/file, 20k-entry lexicon, single MI355X").  `--scaling strong` (BASELINE config 3's
shape: ONE corpus sharded over the GPUs): `--files` is the whole job's file count, rank r takes the contiguous range
shard_range(files, r, N) of that one seeded corpus, and after the timed region rank 0 checks that the shards, in rank
order, are the single-GPU output (per-shard sha256 against its own run over the whole corpus, when that fits
`--verify-files`), that the all-reduced histogram and counters equal the single-GPU ones, and - at the committed golden
size - that the whole run's digest is the one in tests/golden/corpus_digests.json.

`--gpus N` with N > 1, started as a plain `python bench.py`, starts N child processes of this same script (one
per GPU, before this process has imported torch or made any GPU call) and waits for them; started under
torch.distributed.run (WORLD_SIZE already in the environment) the process is one of the ranks itself.

Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E peak


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--files", type=int, default=1_000_000, help="files per GPU")
    ap.add_argument("--config", default="c2", help="corpus configuration: c1..c5")
    ap.add_argument("--seed", type=int, default=1234)
    ap.add_argument("--cpu-sample", type=int, default=300_000, help="files of the same corpus timed on the host cores (rank 0, 1 GPU only); 0 = skip")
    ap.add_argument("--cpu-threads", type=int, default=0, help="0 = min(16, cpu count)")
    ap.add_argument("--traffic-json", default=os.path.join(ROOT, "profiles", "pmc_traffic.json"))
    ap.add_argument("--cpu-single-sample", type=int, default=20_000, help="files timed on ONE host thread; 0 = skip")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"], help="nccl (RCCL) is the measured configuration; gloo (sums on the host) is for rehearsals")
    ap.add_argument("--one-gpu", action="store_true", help="rehearsal: every rank uses GPU 0 (RCCL refuses two ranks on one GPU: use --backend gloo)")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"], help="weak: --files per GPU; strong: --files in all, one corpus sharded by rank")
    ap.add_argument("--verify-files", type=int, default=2_000_000, help="strong scaling: rank 0 reruns the whole corpus on its own GPU to check the shards when the corpus has at most this many files")
    ap.add_argument("--no-check", action="store_true", help="diagnostic builds whose outputs are wrong on purpose: skip the oracle check")
    ap.add_argument("--e2e-runs", type=int, default=2, help="1 GPU: timed calls of tp_run_lines (host buffers in, host results out) for the end-to-end rate; 0 = skip")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # not under a launcher: be the launcher.  Nothing in this process has touched the GPU.
        from imagecaptioner_amd.dist import spawn_ranks
        raise SystemExit(spawn_ranks(args.gpus, [sys.executable, os.path.abspath(__file__)] + sys.argv[1:]))

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != max(args.gpus, 1):
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py rank %d of %d needs a GPU: the tag pipeline has no CPU path" % (rank, world))
    if args.one_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=dev)
        else:
            dist.init_process_group(backend="gloo")

    from imagecaptioner_amd import Config, TagPipe
    from imagecaptioner_amd._abi import tp_device_io
    from imagecaptioner_amd.corpus import make_corpus

    # ---- synthetic shard of this rank ----
    # weak: every rank draws its own --files files from the same tables (seeded by rank);
    # strong: ONE corpus of --files files, of which this rank builds and owns the contiguous range shard_range gives it
    from imagecaptioner_amd.dist import check_shards_against_whole, shard_digest, shard_range
    strong = args.scaling == "strong"
    t0 = time.time()
    if strong:
        shard_lo, shard_hi = shard_range(args.files, rank, world)
        corpus = make_corpus(args.config, args.files, seed=args.seed, file_range=(shard_lo, shard_hi))
    else:
        corpus = make_corpus(args.config, args.files, seed=args.seed + 7919 * rank)
    gen_s = time.time() - t0
    cfg = Config(device=local_rank, **corpus.config_kwargs)
    pipe = TagPipe(cfg)
    n = corpus.n_files
    in_bytes = int(corpus.offsets[-1])

    arena = torch.from_numpy(corpus.arena).to(dev)
    arena = torch.cat([arena, torch.zeros(64, dtype=torch.uint8, device=dev)])
    offsets = torch.from_numpy(corpus.offsets.astype(np.int64)).to(dev)
    cap_capacity = max(4 * in_bytes, 1 << 20)
    captions = torch.empty(cap_capacity, dtype=torch.uint8, device=dev)
    cap_offsets = torch.empty(n + 1, dtype=torch.int64, device=dev)
    verdicts = torch.empty(max(n, 1) * 16, dtype=torch.uint8, device=dev)
    hist = torch.zeros(pipe.info.n_strings + 1, dtype=torch.int64, device=dev)
    counters = torch.zeros(8, dtype=torch.int64, device=dev)

    io = tp_device_io()
    io.arena, io.offsets, io.n_files = arena.data_ptr(), offsets.data_ptr(), n
    io.captions, io.caption_capacity, io.caption_offsets = captions.data_ptr(), cap_capacity, cap_offsets.data_ptr()
    io.verdicts, io.histogram, io.counters = verdicts.data_ptr(), hist.data_ptr(), counters.data_ptr()
    io.fired, io.facet_choice = None, None
    io.stream = torch.cuda.current_stream().cuda_stream

    def step():
        hist.zero_()
        counters.zero_()
        nbytes = pipe.run_lines_device(io)
        if world > 1 and args.backend == "nccl":
            dist.all_reduce(hist)        # RCCL sum over xGMI: the corpus-wide tag histogram
            dist.all_reduce(counters)
        elif world > 1:                  # rehearsal: the same sums through host memory
            for t in (hist, counters):
                h = t.cpu()
                dist.all_reduce(h)
                t.copy_(h)
        return nbytes

    for _ in range(args.warmup):
        step()
    sel_ms, gen_ms, emit_ms, tot_ms, front_ms, back_ms, fused_ms, gen_files = [], [], [], [], [], [], [], 0
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t_begin = time.perf_counter()
    cap_bytes = 0
    for _ in range(args.steps):
        cap_bytes = step()
        tm = pipe.last_timing()        # HIP events recorded by the library on the stream its kernels run on
        sel_ms.append(tm["select_ms"]); gen_ms.append(tm["general_ms"]); emit_ms.append(tm["emit_ms"]); tot_ms.append(tm["total_ms"])
        front_ms.append(tm["lean_front_ms"]); back_ms.append(tm["lean_back_ms"]); fused_ms.append(tm["fused_ms"])
        gen_files = tm["general_files"]
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t_begin
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        # every rank counted its own files into the summed histogram: the sum over ranks of the files' tag counts
        tags = torch.tensor([int(np.frombuffer(verdicts.cpu().numpy().tobytes(), dtype=np.uint16)[1::8].astype(np.int64).sum())], dtype=torch.int64,
                            device=t.device)
        dist.all_reduce(tags)
        if int(hist.sum().item()) != int(tags.item()):
            raise SystemExit("bench.py: the all-reduced histogram does not add up to the ranks' tag counts")

    # ---- strong scaling: the shards in rank order are the single-GPU output; the summed histogram is the single-GPU one ----
    strong_check = None
    if strong:
        mine = shard_digest(captions[:int(cap_bytes)].cpu().numpy(), range(n + 1))
        digests = [mine]
        if world > 1:
            digests = [None] * world
            dist.all_gather_object(digests, mine)
        if rank == 0:
            strong_check = {"files_in_all": args.files, "shard_files": [d[2] for d in digests], "shard_caption_bytes": [d[1] for d in digests],
                            "whole_run_checked": False, "golden_checked": False}
            if args.files <= args.verify_files:
                import hashlib
                full = corpus if world == 1 else make_corpus(args.config, args.files, seed=args.seed)
                fa = torch.cat([torch.from_numpy(full.arena).to(dev), torch.zeros(64, dtype=torch.uint8, device=dev)])
                fo = torch.from_numpy(full.offsets.astype(np.int64)).to(dev)
                fcap_n = max(4 * int(full.offsets[-1]), 1 << 20)
                fcap = torch.empty(fcap_n, dtype=torch.uint8, device=dev)
                fcoff = torch.empty(args.files + 1, dtype=torch.int64, device=dev)
                fver = torch.empty(max(args.files, 1) * 16, dtype=torch.uint8, device=dev)
                fhist, fctr = torch.zeros_like(hist), torch.zeros_like(counters)
                fio = tp_device_io()
                fio.arena, fio.offsets, fio.n_files = fa.data_ptr(), fo.data_ptr(), args.files
                fio.captions, fio.caption_capacity, fio.caption_offsets = fcap.data_ptr(), fcap_n, fcoff.data_ptr()
                fio.verdicts, fio.histogram, fio.counters = fver.data_ptr(), fhist.data_ptr(), fctr.data_ptr()
                fio.fired, fio.facet_choice = None, None
                fio.stream = torch.cuda.current_stream().cuda_stream
                fbytes = pipe.run_lines_device(fio)
                torch.cuda.synchronize()
                whole_caps = fcap[:fbytes].cpu().numpy()
                whole_offs = fcoff.cpu().numpy().astype(np.uint64)
                msg = check_shards_against_whole(digests, whole_caps, whole_offs, world)
                if msg is None and not torch.equal(fhist, hist):
                    msg = "the all-reduced histogram differs from the single-GPU histogram"
                if msg is None and not torch.equal(fctr, counters):
                    msg = "the all-reduced counters differ from the single-GPU counters"
                if msg is not None:
                    raise SystemExit("bench.py --scaling strong: " + msg)
                strong_check["whole_run_checked"] = True
                try:
                    with open(os.path.join(ROOT, "tests", "golden", "corpus_digests.json")) as f:
                        g = json.load(f).get(args.config)
                except OSError:
                    g = None
                if g and g["n_files"] == args.files and args.seed == 1234:
                    if hashlib.sha256(whole_caps.tobytes()).hexdigest() != g["captions_sha256"]:
                        raise SystemExit("bench.py --scaling strong: the captions' digest is not the committed golden digest")
                    strong_check["golden_checked"] = True
                del fa, fo, fcap, fcoff, fver

    # ---- correctness spot check outside the timed region: device result == oracle on a prefix ----
    check_n = min(n, 2000)
    from oracle.oracle_py import Oracle
    oracle = Oracle(Config(**corpus.config_kwargs))
    want = oracle.run_lines(corpus.arena, corpus.offsets[:check_n + 1], threads=8)   # (a rank whose shard is empty checks an empty prefix)
    got_off = cap_offsets[:check_n + 1].cpu().numpy().astype(np.uint64)
    got_cap = captions[:int(got_off[-1])].cpu().numpy()
    got_ver = verdicts[:check_n * 16].cpu().numpy().copy().view(want.verdicts.dtype)
    want_ver = want.verdicts.copy()
    assert np.all(got_ver["n_evidence"] == 0xFFFF)     # the evidence set is only counted for callers that take the detail arrays
    got_ver["n_evidence"] = 0
    want_ver["n_evidence"] = 0
    ok = bool(np.array_equal(got_off, want.caption_offsets) and np.array_equal(got_cap, want.captions) and got_ver.tobytes() == want_ver.tobytes())
    if not ok and not args.no_check:
        raise SystemExit("bench.py: device captions differ from the oracle on the first %d files" % check_n)

    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    files_total = args.files if strong else n * world
    value = files_total * args.steps / elapsed
    # algorithmic bytes per file: input line + input offset + caption line + caption offset + verdict record
    alg_bytes = in_bytes + 8 * n + int(cap_bytes) + 8 * n + 16 * n
    # Roofline of the step: the files' algorithmic bytes over the time from the first kernel's start to the last kernel's
    # end (every kernel of the step, HIP events on the stream the library launches on).  The longest kernel is named beside it.
    mean = lambda v: float(np.mean(v))
    kernel_ms = {"k_lean_a": mean(front_ms), "k_lean_b": mean(back_ms), "later_wave_stages": mean(sel_ms) - mean(front_ms) - mean(back_ms),
                 "k_lean_bx": mean(fused_ms), "k_pipeline": mean(gen_ms), "k_emit+k_hist": mean(emit_ms), "first_to_last_kernel": mean(tot_ms)}
    dominant = max((k for k in kernel_ms if k != "first_to_last_kernel"), key=lambda k: kernel_ms[k])
    step_ms = kernel_ms["first_to_last_kernel"]
    achieved = alg_bytes / (step_ms * 1e-3) / 1e9
    traffic = None
    try:
        with open(args.traffic_json) as f:
            tj = json.load(f)
        if tj.get("workload") == "%s:%d" % (args.config, n):
            traffic = tj.get("hbm_bytes_per_step")
    except Exception:
        pass

    out = {
        "metric": "tag files captioned/sec (whole node)",
        "value": value,
        "unit": "files/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": args.scaling,
        "vs_baseline": None,
        "dtype": "u8",
        "data": "synthetic",
        "config": {"workload": ("%s: ONE corpus of %d tag files sharded over %d GPU(s), %.0f input bytes/file" % (args.config, args.files, world, in_bytes / max(n, 1))) if strong
                               else "%s: %d tag files per GPU, %.0f input bytes/file" % (args.config, n, in_bytes / n),
                   "files_per_gpu": n, "input_bytes_per_file": in_bytes / n, "caption_bytes_per_file": cap_bytes / n,
                   "kernel_ms": kernel_ms,
                   "files_in_general_kernel": int(gen_files), "fused_reruns": int(tm["fused_reruns"]),
                   "parallelism": ("files sharded by rank; histogram all-reduce (%s)" % ("RCCL" if args.backend == "nccl" else "gloo rehearsal")) if world > 1 else "single GPU",
                   "collective": {"backend": (args.backend if world > 1 else None), "world_size_seen": (dist.get_world_size() if world > 1 else 1)},
                   "strong_scaling_check": strong_check,
                   "corpus_gen_s": gen_s},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                     "traffic": traffic, "algorithmic_bytes_per_launch": alg_bytes, "avg_launch_ms": step_ms,
                     "definition": "algorithmic bytes of one step (input lines, both offset arrays, captions, verdicts) / first-to-last kernel time of "
                                   "the step, all kernels; traffic = HBM bytes of all kernels of one step from the FETCH_SIZE/WRITE_SIZE passes",
                     "longest_kernel": dominant, "longest_kernel_ms": kernel_ms[dominant]},
    }

    if world == 1 and args.e2e_runs > 0:
        # End to end through tp_run_lines: pinned host buffers in (H2D), the same kernels, every result back in host memory
        # (D2H: captions, offsets, verdicts, histogram, detail arrays).  File-system time is not part of it.  Never `value`.
        h_arena = torch.from_numpy(corpus.arena).pin_memory()
        h_offs = torch.from_numpy(corpus.offsets.astype(np.int64)).pin_memory()
        L, h = pipe._L, pipe._h
        from imagecaptioner_amd._abi import tp_result
        best = None
        for k in range(args.e2e_runs + 1):                       # the first call grows the library's staging buffers
            res = tp_result()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            rc = L.tp_run_lines(h, h_arena.data_ptr(), h_offs.data_ptr(), n, C.byref(res))
            dt = time.perf_counter() - t1
            if rc != 0:
                raise SystemExit("bench.py: tp_run_lines failed: %s" % L.tp_last_error().decode())
            assert int(res.caption_offsets[n]) == int(cap_bytes)
            L.tp_release(h, C.byref(res))
            if k > 0:
                best = dt if best is None else min(best, dt)
        out["config"]["end_to_end_files_per_s"] = n / best
        out["config"]["end_to_end"] = {"files_per_s": n / best, "ms_per_call": best * 1e3, "calls_timed": args.e2e_runs,
                                       "what": "tp_run_lines: pinned host arena + offsets in (H2D %.2f GB), all kernels, captions + offsets + verdicts + histogram + "
                                               "detail arrays back to host memory (D2H %.2f GB), in pieces of 131072 files whose copies run beside the kernels of the piece between them; "
                                               "best of the timed calls; no file-system time" %
                                               ((in_bytes + 8 * n) / 1e9, (int(cap_bytes) + 8 * n + 16 * n) / 1e9)}

    if world == 1 and args.cpu_sample > 0:
        threads = args.cpu_threads or min(16, os.cpu_count() or 1)
        m = min(n, args.cpu_sample)
        t1 = time.perf_counter()
        oracle.run_lines(corpus.arena, corpus.offsets[:m + 1], threads=threads)
        dt = time.perf_counter() - t1
        out["cpu_baseline"] = {"value": m / dt, "unit": "files/s", "cores": threads, "kind": "port",
                               "sample": "first %d files of the same corpus, C++ restatement of Program.cs (not the .NET binary), %d threads, %.1f s" % (m, threads, dt)}
        m1 = min(n, args.cpu_single_sample)
        if m1 > 0:
            t1 = time.perf_counter()
            oracle.run_lines(corpus.arena, corpus.offsets[:m1 + 1], threads=1)
            dt1 = time.perf_counter() - t1
            out["cpu_baseline"]["single_thread"] = {"value": m1 / dt1, "unit": "files/s", "cores": 1,
                                                    "sample": "first %d files of the same corpus, same restatement, 1 thread, %.1f s" % (m1, dt1)}
    print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
