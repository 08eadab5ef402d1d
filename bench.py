#!/usr/bin/env python3
"""Benchmark of the batch tag pipeline on MI355X.

One "step" is one pass of the whole hot path (tp_run_lines_device: workspace sizing, the per-file pipeline
kernels, caption-offset scan, caption/histogram emit) over one resident shard of synthetic tag files, plus —
on more than one GPU — the RCCL all-reduce of the tag histogram and counters.  Inputs are in HBM before
the timed region starts.  Weak scaling: every rank owns `--files` files (default 1,000,000: BASELINE
config 2, "1M tag files, ~40 tags/file, 20k-entry lexicon, single MI355X").

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

    # ---- synthetic shard of this rank (seeded; every rank draws its own files from the same tables) ----
    t0 = time.time()
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
    verdicts = torch.empty(n * 16, dtype=torch.uint8, device=dev)
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
    sel_ms, gen_ms, emit_ms, tot_ms, front_ms, back_ms, gen_files = [], [], [], [], [], [], 0
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t_begin = time.perf_counter()
    cap_bytes = 0
    for _ in range(args.steps):
        cap_bytes = step()
        tm = pipe.last_timing()        # HIP events recorded by the library on the stream its kernels run on
        sel_ms.append(tm["select_ms"]); gen_ms.append(tm["general_ms"]); emit_ms.append(tm["emit_ms"]); tot_ms.append(tm["total_ms"])
        front_ms.append(tm["lean_front_ms"]); back_ms.append(tm["lean_back_ms"])
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

    # ---- correctness spot check outside the timed region: device result == oracle on a prefix ----
    check_n = min(n, 2000)
    from oracle.oracle_py import Oracle
    oracle = Oracle(Config(**corpus.config_kwargs))
    want = oracle.run_lines(corpus.arena, corpus.offsets[:check_n + 1], threads=8)
    got_off = cap_offsets[:check_n + 1].cpu().numpy().astype(np.uint64)
    got_cap = captions[:int(got_off[-1])].cpu().numpy()
    got_ver = verdicts[:check_n * 16].cpu().numpy().copy().view(want.verdicts.dtype)
    want_ver = want.verdicts.copy()
    assert np.all(got_ver["n_evidence"] == 0xFFFF)     # the evidence set is only counted for callers that take the detail arrays
    got_ver["n_evidence"] = 0
    want_ver["n_evidence"] = 0
    ok = bool(np.array_equal(got_off, want.caption_offsets) and np.array_equal(got_cap, want.captions) and got_ver.tobytes() == want_ver.tobytes())
    if not ok:
        raise SystemExit("bench.py: device captions differ from the oracle on the first %d files" % check_n)

    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    files_total = n * world
    value = files_total * args.steps / elapsed
    # algorithmic bytes per file: input line + input offset + caption line + caption offset + verdict record
    alg_bytes = in_bytes + 8 * n + int(cap_bytes) + 8 * n + 16 * n
    # Roofline of the step: the files' algorithmic bytes over the time from the first kernel's start to the last kernel's
    # end (every kernel of the step, HIP events on the stream the library launches on).  The longest kernel is named beside it.
    mean = lambda v: float(np.mean(v))
    kernel_ms = {"k_lean_a": mean(front_ms), "k_lean_b": mean(back_ms), "later_wave_stages": mean(sel_ms) - mean(front_ms) - mean(back_ms),
                 "k_pipeline": mean(gen_ms), "k_emit+k_hist": mean(emit_ms), "first_to_last_kernel": mean(tot_ms)}
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
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "u8",
        "data": "synthetic",
        "config": {"workload": "%s: %d tag files per GPU, %.0f input bytes/file" % (args.config, n, in_bytes / n),
                   "files_per_gpu": n, "input_bytes_per_file": in_bytes / n, "caption_bytes_per_file": cap_bytes / n,
                   "kernel_ms": kernel_ms,
                   "files_in_general_kernel": int(gen_files),
                   "parallelism": ("files sharded by rank; histogram all-reduce (%s)" % ("RCCL" if args.backend == "nccl" else "gloo rehearsal")) if world > 1 else "single GPU",
                   "corpus_gen_s": gen_s},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                     "traffic": traffic, "algorithmic_bytes_per_launch": alg_bytes, "avg_launch_ms": step_ms,
                     "definition": "algorithmic bytes of one step (input lines, both offset arrays, captions, verdicts) / first-to-last kernel time of "
                                   "the step, all kernels; traffic = HBM bytes of all kernels of one step from the FETCH_SIZE/WRITE_SIZE passes",
                     "longest_kernel": dominant, "longest_kernel_ms": kernel_ms[dominant]},
    }

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
