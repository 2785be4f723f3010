#!/usr/bin/env python3
"""Headline benchmark: Lloyd iterations/s of the sparsified K-means hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one full Lloyd iteration (assign -> accumulate -> [RCCL all-reduce] -> finalize,
kmeans_sparsified.m:417-486) over the whole synthetic dataset, which is generated directly into
HBM before the timed region (FWHT-mixed, 5 %-sparsified Gaussian mixture).
Workload: BASELINE.json's metric config -- N=1e8 points, d=1024, K=100 -- held by ONE GPU at
--gpus 1 (≈62 GB of the 288 GB HBM) and sharded by points over N GPUs otherwise (strong scaling).

What the timed steps are (so that `value` does not depend on --warmup / --steps): iterations of kmeans_sparsified's
own loop -- runs from the start centres until norm(centersOld-centers,'fro') < Tol (kmeans_sparsified.m:476,
Tol = 1e-6, MaxIter = 100), the host reading dff after every iteration as the driver does; when a run has
converged the next one starts from the same start centres with the library's carried state dropped
(spkm_shard_reset_policy: what a new replicate does).  The W warm-up steps run the same loop and are followed by
such a reset, so the K timed steps always begin with a cold first iteration.  `regimes` reports one complete run
iteration by iteration (cold first iteration / mean over the run / converged) on this dataset and on the same
mixture in shuffled point order.

Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md chip table: 8.0 TB/s spec
FP64_VALU_PEAK_TOPS = 39.3     # 256 CU x 4 SIMD x 16 f64 lanes/clk x 2.4 GHz (non-fused ops; 78.6 TFLOP/s counts FMA as 2)
TOL, MAXITER = 1e-6, 100       # kmeans_sparsified.m:133,135 defaults
DUMP = bool(os.environ.get("SPKM_BENCH_DUMP"))   # per-call kernel times and forms of the timed region on stderr (diagnostics)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--n-total", type=float, default=1e8)
    ap.add_argument("--dim", type=int, default=1024)
    ap.add_argument("--clusters", type=int, default=100)
    ap.add_argument("--sparsity", type=float, default=0.05)
    ap.add_argument("--seed", type=int, default=234)
    ap.add_argument("--cpu-sample", type=int, default=1_000_000, help="points timed on the CPU oracle (0 = skip)")
    ap.add_argument("--gen-chunk", type=int, default=131072)
    ap.add_argument("--order", choices=["block", "shuffled"], default="block",
                    help="point order of the dataset the headline value is measured on: cluster-contiguous blocks "
                         "(example_sparseKMeans.m:19-22, SURVEY 8(d)) or arbitrary order; the other one is reported "
                         "under `regimes` unless --no-regimes")
    ap.add_argument("--start", choices=["sample", "planted"], default="sample",
                    help="initial centres: K mixture points drawn with replacement (default: duplicate and uncovered "
                         "clusters, a run needs many iterations) or the K planted means + noise (converges at once)")
    ap.add_argument("--noise", type=float, default=0.1,
                    help="sigma of the mixture components (example_sparseKMeans.m:20-21 uses 0.1: clusters 14 radii apart). "
                         "The `overlap` regime reruns the headline at --overlap-noise, where the sampled distances of "
                         "neighbouring clusters overlap and the carried bounds keep failing")
    ap.add_argument("--overlap-noise", type=float, default=1.5)
    ap.add_argument("--layout", choices=["records", "csc"], default="records",
                    help="what the device sparsifier writes for the synthetic workloads: the library's record layout, adopted "
                         "as it is (the entries exist once: peak = resident), or CSC arrays (the reference's format; the "
                         "library builds its layouts from them on the first call and the arrays are released afterwards: the "
                         "entries exist twice for a moment)")
    ap.add_argument("--no-regimes", action="store_true", help="skip the traced runs to convergence (quick experiments)")
    ap.add_argument("--no-pmc", action="store_true",
                    help="do not measure roofline.traffic in this invocation (two short child runs of this script under "
                         "`rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE`, before the dataset is generated; single GPU, "
                         "screen path only): the figure then comes from the committed profile and says so")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)   # (a child run of the PMC passes)
    ap.add_argument("--detail-out", default=None,
                    help="where the full result goes (regimes' per-iteration arrays, per-kernel notes): default "
                         "gpurun_out/bench_detail.json when that directory exists, else ./bench_detail.json; stdout carries "
                         "one compact line (< 8 KB) only")
    ap.add_argument("--workload", choices=["headline", "config3", "config5"], default="headline",
                    help="headline: BASELINE.json's metric config (N=1e8, d=1024, K=100).  config3: MNIST-shaped "
                         "60000 x 784 -> 1024, K=10.  config5: one GPU's shard of the 1e9 x 784 one-pass config "
                         "(1.25e8 points per GPU, K=10), 8-bit points streamed from pinned host memory through the "
                         "sparsifier; reports the ingest rate beside the Lloyd rate")
    args = ap.parse_args()
    if args.workload == "config3":
        args.n_total, args.dim, args.clusters = 6e4, 784, 10
    elif args.workload == "config5":
        args.n_total, args.dim, args.clusters, args.order = 1.25e8 * args.gpus, 784, 10, "shuffled"

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # not under a launcher: start one process per GPU ourselves (the contract's command line, 127.0.0.1 rendezvous)
        # and hand its output and exit status through -- `python bench.py --gpus N` is a complete command
        sys.exit(self_launch(args.gpus))

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} processes")
    # developer aid (1-GPU boxes): SPKM_BENCH_ONE_DEVICE=1 puts every rank on cuda:0 and
    # SPKM_BENCH_BACKEND=gloo replaces RCCL, so the multi-rank control flow can be exercised there
    if os.environ.get("SPKM_BENCH_ONE_DEVICE"):
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("SPKM_BENCH_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)

    if os.environ.get("SPKM_AB_LIB"):   # developer aid: another build of libspkm.so (A/B on one box; tools/ab_lib.py)
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import ab_lib  # noqa: F401
    from sparsifiedkmeans_amd import _lib, synth
    from sparsifiedkmeans_amd.engine import LloydEngine, Shard, mix_device, mix_sample_device, torch_context

    ctx = torch_context(local_rank)
    L = _lib.lib()
    # the per-iteration all-reduce: libspkm.so's own RCCL communicator on the context's stream (spkm_lloyd_iter: one
    # library call per iteration); torch.distributed only carries the rendezvous token, the barriers and the timing max.
    # If the communicator cannot be set up the exchange falls back to torch.distributed.all_reduce (also RCCL).
    allreduce_via = "none (single GPU)"
    if world > 1:
        from sparsifiedkmeans_amd.engine import attach_rccl
        try:
            if os.environ.get("SPKM_BENCH_TORCH_ALLREDUCE") or os.environ.get("SPKM_BENCH_ONE_DEVICE"):
                raise RuntimeError("switched off by environment")
            attach_rccl(ctx)
            allreduce_via = "libspkm.so RCCL communicator (ncclAllReduce f64 SUM inside spkm_lloyd_iter)"
        except Exception as e:
            allreduce_via = f"torch.distributed.all_reduce ({dist.get_backend()}); libspkm communicator unavailable: {e}"
    n_total = int(args.n_total)
    p, K = args.dim, args.clusters
    # roofline.traffic, measured in THIS invocation: HBM bytes of the full-work launch of the assignment kernel from two
    # counter passes (MI355X_MICROARCH.md: FETCH_SIZE and WRITE_SIZE do not fit one pass; --pmc only, no trace domain)
    live_pmc = None
    if rank == 0 and world == 1 and not args.no_pmc and not args.pmc_child and args.workload != "config5":
        live_pmc = pmc_passes(args)
    first = rank * n_total // world
    n_local = (rank + 1) * n_total // world - first

    def sync_all():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def make_dataset(order, noise_sigma=None):
        t = time.time()
        noise_sigma = args.noise if noise_sigma is None else noise_sigma
        if args.workload == "config5":
            data = synth.streamed_pixel_dataset(ctx, p, n_local, first, K, args.sparsity, seed=args.seed, chunk=args.gen_chunk,
                                                layout=args.layout)
        else:
            data = synth.sparsified_gmm_device(ctx, p, n_local, n_total, first, K, args.sparsity, seed=args.seed,
                                               chunk=args.gen_chunk, order=order, noise=noise_sigma, layout=args.layout)
        if "rec" in data:
            shard = Shard.from_records(ctx, data["p2"], n_local, data["s"], data["rec"], data["ir_bits"])
        else:
            shard = Shard.from_device(ctx, data["p2"], data["jc"], data["ir"], data["x"], nnz=data["nnz"])
        # initial centres: K mixture points in the ORIGINAL space passed through mix(), as the
        # 'Start'-matrix path does (kmeans_sparsified.m:401-406); identical on every rank
        g = torch.Generator(device="cuda")
        g.manual_seed(args.seed + 17)
        lab = torch.randint(0, K, (K,), generator=g, device="cuda")
        if args.start == "planted":
            lab = torch.arange(K, device="cuda")
        noise = 10.0 if args.workload == "config5" else noise_sigma
        start = data["means"][lab] + noise * torch.randn((K, p), generator=g, device="cuda", dtype=torch.float64)
        centers0 = mix_device(ctx, start.contiguous(), data["p2"], data["sign"], 1.0, float(np.sqrt(np.float64(data["p2"]))))
        torch.cuda.synchronize()
        return data, shard, centers0, time.time() - t

    data, shard, centers0, t_gen = make_dataset(args.order)
    p2, s, gamma = data["p2"], data["s"], data["gamma"]

    # the one-off preconditioner, reported separately: dense mix() and the fused mix+sample on one chunk
    fw = None
    if rank == 0:
        mcols = min(131072, n_local)
        xd = torch.randn((mcols, p), device="cuda", dtype=torch.float64)
        irt = torch.zeros(mcols * s + 16, dtype=torch.int16 if p2 <= 65536 else torch.int32, device="cuda")
        xt = torch.zeros(mcols * s + 16, dtype=torch.float64, device="cuda")
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        mix_device(ctx, xd, p2, data["sign"], 1.0, 32.0)
        ev[0].record()
        for _ in range(5):
            mix_device(ctx, xd, p2, data["sign"], 1.0, float(np.sqrt(np.float64(p2))))
        ev[1].record()
        for _ in range(5):
            mix_sample_device(ctx, xd, p2, data["sign"], 1.0, float(np.sqrt(np.float64(p2))), s, 1, 0, irt, xt)
        ev[2].record()
        torch.cuda.synchronize()
        t_dense, t_fused = ev[0].elapsed_time(ev[1]) / 5e3, ev[1].elapsed_time(ev[2]) / 5e3
        fw = {"columns": mcols, "m": p2,
              "dense_mix_GBs": mcols * (p + p2) * 8 / t_dense / 1e9, "dense_mix_columns_per_s": mcols / t_dense,
              "fused_mix_sample_GBs": mcols * (p * 8 + s * 12) / t_fused / 1e9,
              "fused_mix_sample_columns_per_s": mcols / t_fused}
        del xd, irt, xt

    READBACK = bool(os.environ.get("SPKM_BENCH_READBACK"))

    class Loop:
        """kmeans_sparsified's iteration loop on the engine: iterate, read dff (one small D2H per iteration, as the
        driver does), restart from the start centres when dff < Tol or MaxIter is reached."""

        def __init__(self, shard_, centers0_):
            self.shard, self.c0 = shard_, centers0_
            # as kmeans_sparsified() with Display off: the objective is wanted for the LAST iteration of a run only
            # (kmeans_sparsified.m:489-503), so the library may leave it out of the other iterations' calls
            # (spkm_shard_set_lazy_stats); it comes with the run's distances, on demand
            shard_.set_lazy_stats(not os.environ.get("SPKM_BENCH_EAGER_STATS"))
            self.eng = LloydEngine(shard_, K, gamma)
            self.centers = centers0_.clone()
            self.prev = centers0_.clone()           # the centres the latest assignment was computed with
            self.restart()
            self.runs_completed, self.run_lengths = 0, []
            self.forms = []
            self.calls, self.cold_calls = 0, []     # fused calls issued / which of them were a run's first (cold) iteration

        def restart(self):
            self.centers.copy_(self.c0)
            self.shard.reset_policy()
            self.it = 0

        def step(self):
            """one Lloyd iteration; returns (dff, obj, converged_or_capped)"""
            self.prev.copy_(self.centers)
            # (the driver needs dff to decide whether to go on: spkm_lloyd_iter_host hands it over through pinned host memory
            #  the device maps -- no copy kernel, no stream synchronisation; SPKM_BENCH_READBACK=1: iterate + a device-to-host read)
            if READBACK:
                out = self.eng.iterate(self.centers, want_mind=False).cpu().numpy()
            else:
                out = self.eng.iterate_host(self.centers, want_mind=False)
            self.it += 1
            if self.it == 1:
                self.cold_calls.append(self.calls)
            self.calls += 1
            if DUMP:   # diagnostics: the form each call took (rounds for all centroids, early-finished pairs, skipped steps)
                md = self.eng.last_screen_mode()
                self.forms.append((self.it, md[0], self.eng.last_screen_rounds()[0], md[3], md[4]))
            dff = float(np.sqrt(out[0]))
            done = dff < TOL or self.it >= MAXITER
            obj = float(np.sqrt(out[1]))            # NaN when the library left the objective out of this call
            if done:
                # what the run returns besides the centres: IDX (eng.assign, written every iteration), D, the
                # distances of the LAST iteration, and its objective -- materialised once per run, as
                # kmeans_sparsified() does
                obj = self.objective_now()
            return dff, obj, done

        def objective_now(self):
            """distances + objective of the latest iteration (under the centres its assignment was computed with)"""
            self.eng.distances(self.prev)
            o2 = self.eng.stats[0:1].clone()
            if world > 1:
                dist.all_reduce(o2, op=dist.ReduceOp.SUM)
            return float(np.sqrt(o2.item()))

        def steps(self, k):
            for _ in range(k):
                _, _, done = self.step()
                if done:
                    self.run_lengths.append(self.it)
                    self.runs_completed += 1
                    self.restart()

    def read_tlog(cap):
        buf = (C.c_double * cap)()
        cnt = C.c_int()
        _lib.check(L.spkm_timing_read(ctx.handle, buf, cap, C.byref(cnt)))
        return np.array(buf[:min(cnt.value, cap)])

    irb = 2 if p2 <= 65536 else 4
    nnz_local = int(shard.nnz)
    # algorithmic bytes of one Lloyd iteration over this GPU's points ...
    b_iter = nnz_local * 12 + (n_local + 1) * 8 + n_local * 12 + 24 * p2 * K
    # ... of which the assignment kernel is credited what IT moves when that is less: on single-tile shapes (K <= 32) the
    # screen reads its own f32 / u16 copy of a point (6 B per stored-entry slot) and writes 12 B per point -- about half
    # of SURVEY 8(d)'s 632 B -- so crediting it with the whole iteration would report more than the memory system moved
    tiles = (K + 31) // 32
    b_screen_own = n_local * (((s + 3) // 4) * 4 * 6 + 12 * tiles) + tiles * (p2 + 1) * 32 * 4
    b_screen = min(b_iter, b_screen_own) if tiles == 1 else b_iter
    # ... and of the exact accumulation pass alone: values + row ids once, the sort permutation in (4 B), the library's
    # upper bound out (4 B; the 8-B min-distance is stored on demand only, once per run), the per-cluster sums and counts out
    b_acc = nnz_local * (8 + irb) + n_local * 8 + 16 * p2 * K
    steps_per_tile = (n_local + 15) // 16

    # host copies for the CPU legs are taken now: the shard's CSC arrays are about to be released
    cpu_data = None
    if rank == 0 and world == 1 and args.cpu_sample > 0 and args.workload != "config5":
        cpu_data = cpu_sample_arrays(data, centers0, s, min(max(args.cpu_sample, 4_000_000), n_local))
    loop = Loop(shard, centers0)
    # (at least two calls: the first builds the shard's record layout and screen copy, the second is where a shard in
    #  arbitrary order is regrouped inside the library -- one-off layout work, like the sparsifier: never in the timed window)
    warm_eff = max(args.warmup, 2)
    sync_all()
    t_w = time.perf_counter()
    loop.steps(warm_eff)
    torch.cuda.synchronize()
    warm_ms = (time.perf_counter() - t_w) * 1e3
    regroup_info = {"own_order_after_warmup": bool(shard.order_info()[0]), "warmup_calls": warm_eff, "warmup_ms": round(warm_ms, 2)}
    free_pk, total_pk = torch.cuda.mem_get_info()   # the moment everything exists at once: dataset + the library's layouts + state
    hbm_after_first_call_GB = round((total_pk - free_pk) / 1e9, 1)
    # from here on the record layout is the only copy of the exact entries (spkm_shard_release_csc): 53 GB of the
    # 146 GB a 1e8-point shard and its layouts occupy go back to the allocator
    csc_released = ("rec" in data) or (shard.release_csc() if not os.environ.get("SPKM_BENCH_KEEP_CSC") else False)
    if csc_released:
        data.pop("x", None)
        data.pop("ir", None)
        torch.cuda.empty_cache()
    loop.restart()                                  # the timed steps start a run, whatever W was
    loop.runs_completed, loop.run_lengths = 0, []
    loop.calls, loop.cold_calls = 0, []
    work0 = loop.eng.screen_work_totals()                                  # running totals of screen rounds (executed, full-work)
    skipped0 = loop.eng.last_screen_mode()[5] if args.warmup > 0 else 0   # running total of steps skipped so far
    acc_pts0 = loop.eng.exact_pass_points()[0]                             # ... and of points the exact pass streamed
    _lib.check(L.spkm_timing_log(ctx.handle, 2))   # screen path: two pairs per call (screen, exact accumulation)
    sync_all()
    t0 = time.perf_counter()
    loop.steps(args.steps)
    sync_all()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    eng = loop.eng
    # the exchange as the library saw it (a SCALE record then proves RCCL ran over N ranks) and what ONE all-reduce of the
    # reduce buffer costs here, timed by itself after the window (20 back-to-back, barrier on both sides, max over ranks)
    from sparsifiedkmeans_amd.engine import comm_info
    c_n, c_r = comm_info(ctx)
    exchange = {"library_comm_nranks": c_n, "library_comm_rank": c_r, "world_size": world, "reduce_buffer_bytes": int(eng.reduce.numel() * 8),
                "allreduce_ms": None}
    if world > 1:
        keep = eng.reduce.clone()
        sync_all()
        ta = time.perf_counter()
        for _ in range(20):
            eng.allreduce_step()
        sync_all()
        dt_ar = torch.tensor([(time.perf_counter() - ta) / 20.0 * 1e3], dtype=torch.float64, device="cuda")
        dist.all_reduce(dt_ar, op=dist.ReduceOp.MAX)
        exchange["allreduce_ms"] = float(dt_ar.item())
        eng.reduce.copy_(keep)
    path, listed = eng.last_path_info()
    mode = eng.last_screen_mode()
    kms = read_tlog(2 * max(args.steps, 1))
    _lib.check(L.spkm_timing_log(ctx.handle, 0))
    if DUMP and rank == 0:
        print("per-call ms:", [round(float(v), 3) for v in kms], "last screen mode:", mode, file=sys.stderr)
        print("forms (iteration, form, rounds for all, early pairs, skipped steps):", loop.forms[-args.steps:], file=sys.stderr)
    if path == 1 and kms.size == 2 * args.steps:
        screen_ms, acc_ms = float(kms[0::2].mean()), float(kms[1::2].mean())
    else:                                       # all-exact path (or a mix after a back-off): the tile kernel only
        screen_ms, acc_ms = (float(kms.mean()) if kms.size else float("nan")), 0.0
    # What the timed window's screen launches did, two ways: the share of their 16-point steps they entered at all (the
    # others were skipped on the bounds carried between calls), and -- what the window's byte credit is weighted by -- the
    # share of their ROUNDS they executed for all centroids (a step that the two-phase forms finish after A of NR rounds is
    # credited A / NR, not 1: spkm_screen_work_totals, counted on the device)
    done = 1.0 - (mode[5] - skipped0) / (steps_per_tile * args.steps) if path == 1 and steps_per_tile else 1.0
    work1 = eng.screen_work_totals()
    rounds_share = ((work1[0] - work0[0]) / (work1[1] - work0[1])) if path == 1 and work1[1] > work0[1] else done
    # share of the points the exact pass streamed in the timed launches (clusters that no point left or entered and whose
    # centroid did not move are not streamed again): it is credited with that share of its bytes only
    acc_share = (eng.exact_pass_points()[0] - acc_pts0) / (n_local * args.steps) if path == 1 and n_local else 1.0
    scr_name = dominant_kernel(path, s)
    bytes_note = ("SURVEY 8(d) bytes of an iteration" if b_screen == b_iter else
                  "single centroid tile: the bytes of its own f32 / u16 copy + its 12-B results, less than SURVEY 8(d)'s 632 B per point")
    # (1) THE roofline quantity: the launch that does ALL the work -- the plain form over every 16-point step,
    # a run's first (cold) iteration -- against an iteration's algorithmic bytes.  The timed window always begins with one.
    cold = [c for c in loop.cold_calls if 2 * c < kms.size] if path == 1 and kms.size == 2 * args.steps else []
    full_ms = float(np.mean([kms[2 * c] for c in cold])) if cold else (screen_ms if path != 1 else float("nan"))
    rl_full = roofline_obj(scr_name, full_ms, int(b_screen),
                           f"assignment kernel, the launch that does all the work: every 16-point step, every round, all K centroids "
                           f"(the cold first iteration of a run; {len(cold)} such launch(es) in the timed window, HIP events on the "
                           f"library's stream); {bytes_note}.  VALU / LDS bound under the power cap at K=100, not HBM bound (DESIGN.md section 4)")
    # (2) the window's mean launch under the share model (what the timed steps mix: plain, hinted and list forms)
    rl_screen = roofline_obj(scr_name, screen_ms, int(b_screen * rounds_share),
                             f"assignment kernel; mean over the {args.steps} timed launches, which executed {rounds_share:.3f} of their "
                             f"rounds for all centroids ({done:.3f} of their 16-point steps were entered at all; the rest skipped on "
                             f"carried bounds, steps finished early by the two-phase forms credited by the rounds they ran); {bytes_note} "
                             "scaled by the rounds share -- a work-avoidance figure, not the section 8(d) roofline quantity")
    rl_acc = roofline_obj("k_exact_accumulate", acc_ms, int(b_acc * acc_share),
                          f"accumulation pass (k_exact_accumulate_rec over every member, or k_accumulate_stream on small K x p; in "
                          "incremental calls k_accumulate_events over the points that changed cluster, credited 0 streamed points); "
                          f"mean over the timed launches, which streamed {acc_share:.3f} of the points (bytes scaled by that share).  "
                          "HBM bound: one pass over the f64 values and row ids in counting-sort order, reference arithmetic "
                          "for each point's distance to its centroid fused with the per-cluster sums (DESIGN.md section 4.2); "
                          "bytes = nnz*(8+2) + n*8 + 16*p*K, all streamed in every launch") if acc_ms > 0 else None
    # traffic: HBM bytes per launch of the full-work form -- measured by this invocation's own counter passes
    # (pmc_passes), else read from the committed profile of the same workload and SAID so
    if live_pmc and live_pmc.get("hbm_bytes_per_launch"):
        rl_full["traffic"] = live_pmc["hbm_bytes_per_launch"]
        rl_full["traffic_over_algorithmic"] = live_pmc["hbm_bytes_per_launch"] / b_screen if b_screen else None
        rl_full["traffic_source"] = {"source": "pmc passes of this invocation", **{k: v for k, v in live_pmc.items() if k != "hbm_bytes_per_launch"}}
    else:
        t, ratio, which = pmc_traffic(scr_name, n_local, K, p2, args.start)
        rl_full["traffic"], rl_full["traffic_over_algorithmic"] = t, ratio
        rl_full["traffic_source"] = {"source": "committed profile" if t else "none", "record": which, "file": "profiles/pmc_latest.json",
                                     "why": (live_pmc or {}).get("error", "--no-pmc" if args.no_pmc else "not a single-GPU screen run")}
    for rl in (rl_screen, rl_acc):
        if rl:   # (by_kernel: the committed passes' figures for the window's kernels, each marked as such)
            t, ratio, which = pmc_traffic(rl["kernel"], n_local, K, p2, args.start)
            rl["traffic"], rl["traffic_over_algorithmic"] = t, ratio
            rl["traffic_source"] = {"source": "committed profile" if t else "none", "record": which, "file": "profiles/pmc_latest.json"}
    # the timed WINDOW's launches (plain, hinted and list forms mixed): launch-weighted HBM bytes from the committed PMC passes
    for rl in (rl_screen, rl_acc):
        if rl:
            w = pmc_window(rl["kernel"], n_local, K, p2)
            if w:
                rl["traffic_window_mean"], rl["traffic_window_source"] = w
                rl["traffic_window_over_algorithmic"] = (w[0] / rl["algorithmic_bytes_per_launch"]) if rl["algorithmic_bytes_per_launch"] else None
    # the VALU wall of the screen's formulation: packed f32 add + fma per (stored entry, centroid
    # pair), 4 issue cycles each on 1024 SIMDs, at the clock the part holds under this kernel (profiles/pmc_latest.json:
    # GRBM_GUI_ACTIVE over the traced duration) and at the 2.4 GHz it is specified for
    if path == 1 and scr_name.startswith("k_screen_quad") and rl_full["kernel_ms"]:
        pk = (n_local / 16.0) * (K / 32.0) * (((s + 3) // 4) * 4) * 8.0       # wave-instructions of a launch that does ALL the work
        clk = pmc_clock(scr_name, n_local, K, p2) or 1.79
        floor_sust, floor_nom = pk * 4.0 / 1024.0 / (clk * 1e9) * 1e3, pk * 4.0 / 1024.0 / 2.4e9 * 1e3
        rl_full["frac_of_valu_floor"] = floor_sust / rl_full["kernel_ms"]
        rl_full["valu_floor"] = {
            "packed_wave_instructions": pk, "issue_cycles_each": 4, "simds": 1024, "sustained_clock_ghz": clk,
            "valu_floor_ms": floor_sust, "valu_floor_ms_at_2.4GHz": floor_nom,
            "frac_of_hbm_roofline_at_the_floor": (b_iter / (floor_sust * 1e-3) / 1e9 / HBM_PEAK_GBS),
            "note": "what a launch that evaluates every (entry, centroid) term would take if it issued nothing but the packed "
                    "add / fma pairs: the ceiling of this formulation, about 0.4 of the HBM roofline -- the 0.60 of "
                    "BASELINE.json's target is not reachable by a packed-f32 VALU screen at K = 100.  The plain launch itself "
                    "(regimes.*.roofline_cold_no_carry) runs 1.7x the floor: 11 VALU instructions per entry where 8 are "
                    "arithmetic, plus winner selection per step; tools/ubench_quad.hip (profiles/r04_ubench_quad.txt) shows "
                    "its rounds alone take 28-29 ms-equivalent because the part drops to 1.57 GHz under VALU + LDS together "
                    "(1.98 GHz VALU only, 2.2 GHz LDS only): power, not issue slots, is what is left"}
    # top level = the section 8(d) quantity: the full-work launch of the assignment kernel (its frac is reproducible from the
    # kernel trace alone: B_iter / average duration of k_screen_quad<NR, IR, 0, false> / 8 TB/s); the window's share model
    # and the accumulation pass hang below it
    roofline = dict(rl_full)
    roofline["window"] = {"kernel": scr_name, "kernel_ms_mean": rl_screen["kernel_ms"], "rounds_executed_share": rounds_share,
                          "steps_entered_share": done, "exact_pass_points_share": acc_share,
                          "credited_bytes_per_launch": rl_screen["algorithmic_bytes_per_launch"], "achieved": rl_screen["achieved"],
                          "frac": rl_screen["frac"], "traffic_window_mean": rl_screen.get("traffic_window_mean"),
                          "traffic_window_over_credited": rl_screen.get("traffic_window_over_algorithmic"),
                          "note": rl_screen["note"]}
    roofline["by_kernel"] = {scr_name + " (full-work launch)": rl_full, scr_name + " (window mean)": rl_screen,
                             **({"k_exact_accumulate": rl_acc} if rl_acc else {})}
    roofline["screen_steps_processed_share"] = done
    roofline["screen_rounds_executed_share"] = rounds_share
    roofline["exact_pass_points_share"] = acc_share
    free_b, total_b = torch.cuda.mem_get_info()
    ops = 3.0 * nnz_local * K
    final_obj = loop.objective_now() if loop.it > 0 else float("nan")   # (after the timed region; every rank calls it)

    result = {
        "metric": "Lloyd iters/sec + achieved HBM GB/s, N=1e8 d=1024 K=100" if args.workload == "headline"
                  else f"Lloyd iters/sec + achieved HBM GB/s, {args.workload}: N={n_total} d={p} K={K}",
        "value": args.steps / elapsed,
        "unit": "Lloyd iters/sec",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "warmup_effective": warm_eff,
        "ms_per_step": 1e3 * elapsed / args.steps,
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "value_definition": "timed steps = consecutive iterations of kmeans_sparsified-style runs from the start centres "
                            f"to dff < {TOL:g} (MaxIter {MAXITER}), host reads dff every iteration; a converged run is "
                            "followed by the next one from the same start with the library's carried state reset; the timed "
                            "region begins at a run's first (cold) iteration regardless of --warmup.  As kmeans_sparsified() "
                            "with Display off, the loop asks for the objective and the distances of a run's LAST iteration only "
                            "(spkm_shard_set_lazy_stats; SPKM_BENCH_EAGER_STATS=1: every iteration)",
        "config": {"workload": f"sparsified GMM N={n_total} d={p} (p2={p2}) K={K} s={s} nnz/point, {args.order} point order, "
                               f"points sharded over {world} GPU(s), dense-centre Lloyd runs to convergence from a "
                               f"'{args.start}' start",
                   "n_total": n_total, "n_per_gpu": n_local, "p2": p2, "K": K, "nnz_per_point": s, "start": args.start,
                   "order": args.order, "tol": TOL, "maxiter": MAXITER,
                   "gamma": gamma, "parallelism": f"dp{world} (1 RCCL all-reduce/iter)" if world > 1 else "single GPU",
                   "allreduce": allreduce_via, "exchange": exchange,
                   "datagen_s": round(t_gen, 1), "final_obj": final_obj,
                   "hbm_resident_GB": round((total_b - free_b) / 1e9, 1), "csc_released": bool(csc_released),
                   "hbm_after_first_call_GB": hbm_after_first_call_GB, "dataset_layout": "records" if "rec" in data else "csc",
                   "library_order": dict(zip(("own_order", "regrouped_in_last_run"), shard.order_info())),
                   "regroup": regroup_info,
                   "runs_completed_in_timed_region": loop.runs_completed, "run_lengths": loop.run_lengths,
                   "assign_path": "f32 screen certified by a rigorous bound + exact f64 confirmation (outputs "
                                  "bit-identical to the all-exact kernels)" if path == 1 else "exact f64 tiles",
                   "uncertified_points_last_iter": listed,
                   "screen_form_last_iter": {0: "plain", 1: "two-phase", 2: "hinted"}.get(mode[0], "none"),
                   "skipped_steps_last_iter": mode[4],
                   "steps_per_centroid_tile": steps_per_tile},
        "roofline": roofline,
        "valu": {"distance_terms_per_s": (nnz_local * K) / (screen_ms * 1e-3) if screen_ms == screen_ms else None,
                 "exact_f64_op_equivalent_Tops": ops / (screen_ms * 1e-3) / 1e12 if screen_ms == screen_ms else None,
                 "f64_nonfused_peak_Tops": FP64_VALU_PEAK_TOPS,
                 "note": "op-equivalents of the exact f64 arithmetic the screen makes unnecessary; exceeds the f64 peak "
                         "whenever steps are skipped or finished early -- not a utilisation figure"},
        # SURVEY 8(d)'s bytes of an iteration over the MEAN timed step (steps that skip work included: not a bandwidth)
        "whole_iter_gbs": b_iter / (elapsed / args.steps) / 1e9,
        "fwht": fw,
    }

    # ---- regimes: one complete run to convergence, iteration by iteration, on this dataset and on the other order ----
    if args.workload == "config5":
        ing = data["ingest"]
        result["config"]["ingest"] = {"points": ing["points"], "bytes_over_pcie": ing["bytes"], "seconds": ing["seconds"],
                                      "GBs": ing["GBs"], "pcie_gen5_x16_spec_GBs": 63.0,
                                      "source": f"uint8 points in pinned host memory ({data['pool_points']} distinct, cycled), "
                                                f"{args.gen_chunk}-point chunks, copy stream + two staging buffers -> "
                                                "spkm_widen_f64_dev -> spkm_mix_sample_dev -> resident sparse shard"}
    if not args.no_regimes and args.workload == "config5":
        regimes = {args.order: traced_run(loop, L, ctx, _lib, read_tlog, world, dist, torch, b_iter, b_acc, scr_name, b_screen)}
        result["regimes"] = regimes
    elif not args.no_regimes:
        regimes = {args.order: traced_run(loop, L, ctx, _lib, read_tlog, world, dist, torch, b_iter, b_acc, scr_name, b_screen)}
        del loop, eng
        shard.close()
        del data, shard
        torch.cuda.empty_cache()
        # the same mixture in the other point order, and -- the worst case beside the best case -- with overlapping
        # clusters (sigma raised until the sampled distances of neighbouring clusters overlap: the carried bounds keep
        # failing, few clusters ever settle, the skip-dependent numbers above do not apply)
        other = "shuffled" if args.order == "block" else "block"
        for name, order_, sigma in ((other, other, None), ("overlap", args.order, args.overlap_noise)):
            data2, shard2, centers02, t_gen2 = make_dataset(order_, sigma)
            loop2 = Loop(shard2, centers02)
            loop2.steps(1)                                     # set-up of the shard's screen copy happens on the first call
            if csc_released and "rec" not in data2 and shard2.release_csc():
                data2.pop("x", None)
                data2.pop("ir", None)
                torch.cuda.empty_cache()
            loop2.restart()
            regimes[name] = traced_run(loop2, L, ctx, _lib, read_tlog, world, dist, torch, b_iter, b_acc, scr_name, b_screen)
            regimes[name]["datagen_s"] = round(t_gen2, 1)
            if sigma is not None:
                regimes[name]["noise_sigma"] = sigma
            del loop2
            shard2.close()
            del data2, shard2
            torch.cuda.empty_cache()
        result["regimes"] = regimes
        # what a user waits for, independent of --steps / --warmup: iterations / time of one whole run from the cold start
        # (to dff < Tol 1e-6, or capped at MaxIter 100: `ended_by` says which) per dataset regime
        result["whole_run_iters_per_s"] = {o: regimes[o]["whole_run_iters_per_s"] for o in regimes}
        result["whole_run_ended_by"] = {o: regimes[o]["ended_by"] for o in regimes}
    if result.get("regimes"):
        # what lazy statistics defer: a run owes the distances + objective pass once (kmeans_sparsified.m:471 evaluates obj
        # every iteration); the timed window of 20 iterations of a 100-iteration run never contains it, so it is added here
        own = result["regimes"].get(args.order) or {}
        fetch = own.get("distances_and_objective_once_per_run_ms")
        result["distances_and_objective_once_per_run_ms"] = {o: r_.get("distances_and_objective_once_per_run_ms")
                                                             for o, r_ in result["regimes"].items()}
        if fetch is not None and fetch == fetch:
            result["value_incl_run_tail"] = args.steps / (elapsed + max(fetch, 0.0) * 1e-3)
        result["whole_iter_cold"] = {k: _r(v) for k, v in (own.get("whole_iter_cold") or {}).items()} or None

    if cpu_data is not None:
        result["cpu_baseline"] = cpu_baseline(cpu_data, p2, K, gamma, s, min(args.cpu_sample, n_local), n_total)
    elif rank == 0:
        result["cpu_baseline"] = None
    if rank == 0:
        emit(result, args.detail_out)
    if world > 1:
        dist.destroy_process_group()


def _strict(o):
    """NaN / inf -> null (lazy calls report NaN objectives by design): the line must be strict JSON."""
    if isinstance(o, float):
        return o if o == o and abs(o) != float("inf") else None
    if isinstance(o, dict):
        return {str(k): _strict(v) for k, v in o.items()}
    if isinstance(o, (list, tuple)):
        return [_strict(v) for v in o]
    if isinstance(o, np.generic):
        return _strict(o.item())
    return o


def _r(v, nd=4):
    return round(v, nd) if isinstance(v, float) and v == v and abs(v) != float("inf") else v


LINE_LIMIT = 8192   # the driver keeps a bounded tail of stdout: BENCH_r05 lost a 22-KB line (parsed = null)


def headline_line(result):
    """The ONE stdout line: the contract's keys, the section-8(d) roofline object once, the CPU baseline, the whole-run
    rates -- nothing that grows with --steps or MaxIter.  Everything else goes to the detail file (emit)."""
    rl = result.get("roofline") or {}
    ts = rl.get("traffic_source") or {}
    win = rl.get("window") or {}
    cfg = result.get("config") or {}
    cb = result.get("cpu_baseline")
    line = {k: result.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "warmup_effective", "ms_per_step",
                                       "higher_is_better", "scaling", "vs_baseline", "dtype", "data")}
    line["value_incl_run_tail"] = result.get("value_incl_run_tail")
    line["config"] = {k: cfg.get(k) for k in ("workload", "n_total", "n_per_gpu", "p2", "K", "nnz_per_point", "start", "order", "tol",
                                              "maxiter", "parallelism", "allreduce", "final_obj", "hbm_resident_GB",
                                              "runs_completed_in_timed_region")
                      if k in cfg}
    if cfg.get("exchange"):
        line["config"]["exchange"] = cfg["exchange"]
    if cfg.get("regroup"):
        line["config"]["regroup"] = cfg["regroup"]
    if cfg.get("ingest"):
        line["config"]["ingest_GBs"] = _r(cfg["ingest"].get("GBs"), 2)
    line["roofline"] = {"schema": 2, "bound": rl.get("bound"), "achieved": _r(rl.get("achieved"), 1), "peak": rl.get("peak"), "unit": rl.get("unit"),
                        "frac": _r(rl.get("frac")), "traffic": rl.get("traffic"), "kernel": rl.get("kernel"),
                        "kernel_ms": _r(rl.get("kernel_ms")), "algorithmic_bytes_per_launch": rl.get("algorithmic_bytes_per_launch"),
                        "traffic_over_algorithmic": _r(rl.get("traffic_over_algorithmic")),
                        "traffic_source": ts.get("source"), "what": "full-work (cold, plain-form) launch of the assignment kernel "
                        "against SURVEY 8(d) B_iter; HIP events on the library's stream",
                        "frac_of_valu_floor": _r(rl.get("frac_of_valu_floor")),
                        "window_mean": {"kernel_ms": _r(win.get("kernel_ms_mean")), "rounds_executed_share": _r(win.get("rounds_executed_share")),
                                        "frac_share_model": _r(win.get("frac"))},
                        "accumulate": None}
    acc = (rl.get("by_kernel") or {}).get("k_exact_accumulate")
    if acc:
        line["roofline"]["accumulate"] = {"kernel_ms": _r(acc.get("kernel_ms")), "frac_share_model": _r(acc.get("frac"))}
    if result.get("whole_iter_cold"):
        line["roofline"]["whole_iter_cold"] = result["whole_iter_cold"]
    if cb:
        line["cpu_baseline"] = {"value": cb.get("value"), "unit": cb.get("unit"), "cores": cb.get("cores"), "kind": cb.get("kind"),
                                "sample": (cb.get("sample") or "")[:300]}
        if cb.get("all_cores"):
            line["cpu_baseline"]["all_cores"] = {"value": cb["all_cores"].get("value"), "cores": cb["all_cores"].get("cores")}
    else:
        line["cpu_baseline"] = None
    for k in ("whole_run_iters_per_s", "whole_run_ended_by", "distances_and_objective_once_per_run_ms"):
        if result.get(k) is not None:
            line[k] = {a: _r(b, 2) for a, b in result[k].items()} if isinstance(result[k], dict) else _r(result[k], 3)
    line["detail"] = result.get("detail_file")
    s = json.dumps(_strict(line), allow_nan=False, separators=(",", ":"))
    if len(s) >= LINE_LIMIT:                      # never silently: shed the optional parts, keep the contract's keys
        for k in ("detail", "whole_run_ended_by", "value_incl_run_tail"):
            line.pop(k, None)
        line["roofline"].pop("window_mean", None)
        line["config"] = {"workload": str(cfg.get("workload"))[:200]}
        s = json.dumps(_strict(line), allow_nan=False, separators=(",", ":"))
    assert len(s) < LINE_LIMIT, len(s)
    return s


def emit(result, detail_out):
    """Detail (regimes' per-iteration arrays, per-kernel notes, the window model) to a side file and to stderr; the
    compact line -- and nothing else -- to stdout, last."""
    detail = json.dumps(_strict(result), allow_nan=False)
    path = detail_out
    if path is None:
        d = os.path.join(ROOT, "gpurun_out")
        path = os.path.join(d if os.path.isdir(d) else ROOT, "bench_detail.json")
    try:
        with open(path, "w") as f:
            f.write(detail + "\n")
        result["detail_file"] = os.path.relpath(path, ROOT)
    except OSError as e:
        result["detail_file"] = f"(not written: {e})"
    print("bench detail: " + detail, file=sys.stderr, flush=True)
    sys.stdout.flush()
    print(headline_line(result), flush=True)


def self_launch(ngpus: int) -> int:
    """`python bench.py --gpus N` without a launcher: re-execute under torch.distributed.run, one rank per GPU."""
    import socket
    import subprocess

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={ngpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC only on these hosts (RCCL across processes)
    env.setdefault("OMP_NUM_THREADS", "8")
    return subprocess.call(cmd, env=env)


def roofline_obj(kernel, ms, nbytes, note):
    ok = ms == ms and ms > 0
    ach = nbytes / (ms * 1e-3) / 1e9 if ok else None
    if ach and ach > HBM_PEAK_GBS and kernel.startswith("k_screen"):
        # SURVEY 8(d) credits an iteration with 12 B per stored entry (f64 value + u32 row id); the screen reads its own
        # f32 copy with 16-bit row ids (6 B per entry), so on shapes where it is not VALU-bound (K <= 16) the credited
        # rate can exceed what the memory system delivers -- not a measurement error
        note = ("credited with SURVEY 8(d)'s 632 B per point while it reads a 306-B f32 / u16 copy of the point: "
                "a rate above the HBM peak means exactly that; the memory system moved about half of it.  " + note)
    return {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": (ach / HBM_PEAK_GBS) if ach else None, "traffic": None, "kernel": kernel, "kernel_ms": ms if ok else None,
            "algorithmic_bytes_per_launch": nbytes, "note": note}


def pmc_traffic(kern, n_local, K, p2, start):
    """(HBM bytes per launch, that / the same launches' algorithmic bytes, which record) of ``kern`` from the committed PMC
    passes (profiles/pmc_latest.json: separate --pmc runs, FETCH_SIZE x 2 + WRITE_SIZE as the guide prescribes), when a
    record for exactly this workload exists.  The ratio is taken inside the PMC record (same launches for both numbers):
    the timed launches of THIS run process a different share of their steps."""
    pmc = os.path.join(ROOT, "profiles", "pmc_latest.json")
    if not os.path.exists(pmc):
        return None, None, None
    try:
        with open(pmc) as f:
            recs = json.load(f)
        for rec in (recs if isinstance(recs, list) else [recs]):
            if (rec.get("n_local") == n_local and rec.get("K") == K and rec.get("p2") == p2
                    and rec.get("headline_for") == kern):
                hb, ab = rec.get("hbm_bytes_per_launch"), rec.get("algorithmic_bytes_per_launch")
                return hb, (hb / ab if hb and ab else None), rec.get("kernel")
    except Exception:
        return None, None, None
    return None, None, None


def pmc_passes(args):
    """HBM bytes per launch of the assignment kernel's full-work form, measured now: two child runs of this script (one cold
    iteration + one more) under `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` (the two do not fit one pass; counters only,
    no trace domain), before this process has generated its dataset.  bytes = FETCH_SIZE x 2 + WRITE_SIZE (KB -> B):
    MI355X_MICROARCH.md's gfx950 correction for wide coalesced reads, calibrated for this kernel's access pattern in
    profiles/r01_fetch_calibration.txt.  Returns a dict (with `error` when a pass did not deliver)."""
    import csv
    import glob
    import re
    import shutil
    import subprocess
    import tempfile

    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return {"error": "rocprofv3 not found"}
    child = [sys.executable, os.path.abspath(__file__), "--pmc-child", "--no-pmc", "--no-regimes", "--cpu-sample", "0", "--steps", "2",
             "--warmup", "1", "--gpus", "1", "--n-total", repr(args.n_total), "--dim", str(args.dim), "--clusters", str(args.clusters),
             "--sparsity", repr(args.sparsity), "--seed", str(args.seed), "--order", args.order, "--start", args.start,
             "--noise", repr(args.noise), "--layout", args.layout, "--workload", args.workload, "--gen-chunk", str(args.gen_chunk),
             "--detail-out", os.devnull]
    plain = re.compile(r"k_screen_quad<\d+, unsigned (short|int), 0, false>")
    out, t0 = {}, time.time()
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="spkm_pmc_", dir="/tmp")
        try:
            env = dict(os.environ, TMPDIR="/tmp")
            env.pop("SPKM_BENCH_DUMP", None)
            r = subprocess.run([exe, "--pmc", ctr, "--kernel-include-regex", "k_screen_quad", "--output-format", "csv", "-d", d,
                                "-o", "pmc", "--"] + child, cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                               timeout=240)
            vals = []
            for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                with open(f) as fh:
                    for row in csv.DictReader(fh):
                        if row.get("Counter_Name") == ctr and plain.search(row.get("Kernel_Name", "")):
                            vals.append(float(row["Counter_Value"]))
            if not vals:
                return {"error": f"{ctr} pass delivered no plain-form k_screen_quad dispatch (rc {r.returncode}): "
                                 + r.stderr.decode(errors="replace")[-300:]}
            out[ctr] = (sum(vals) / len(vals), len(vals))
        except Exception as e:   # a report, never a reason to lose the bench line
            return {"error": f"{ctr} pass: {e!r}"}
        finally:
            shutil.rmtree(d, ignore_errors=True)
    fetch_kb, write_kb = out["FETCH_SIZE"][0], out["WRITE_SIZE"][0]
    return {"hbm_bytes_per_launch": (fetch_kb * 2.0 + write_kb) * 1024.0, "FETCH_SIZE_raw_KB": fetch_kb, "WRITE_SIZE_raw_KB": write_kb,
            "dispatches": out["FETCH_SIZE"][1], "kernel": "k_screen_quad<NR, IR, 0, false> (plain form)",
            "correction": "FETCH_SIZE x 2 + WRITE_SIZE (gfx950: 128-B requests tallied at 64 B; MI355X_MICROARCH.md, "
                          "profiles/r01_fetch_calibration.txt)",
            "command": "rocprofv3 --pmc <FETCH_SIZE | WRITE_SIZE> --kernel-include-regex k_screen_quad -- python bench.py "
                       "--steps 2 --warmup 1 --no-regimes --cpu-sample 0 (same workload), two separate passes",
            "seconds": round(time.time() - t0, 1)}


def _pmc_records():
    pmc = os.path.join(ROOT, "profiles", "pmc_latest.json")
    try:
        with open(pmc) as f:
            recs = json.load(f)
        return recs if isinstance(recs, list) else [recs]
    except Exception:
        return []


def pmc_window(kern, n_local, K, p2):
    """(launch-weighted mean HBM bytes per launch over ALL forms of ``kern`` in the profiled bench run, what it was taken
    over) from profiles/pmc_latest.json's `window_for` record -- the counterpart of `traffic`, which is the plain form's."""
    for rec in _pmc_records():
        if rec.get("n_local") == n_local and rec.get("K") == K and rec.get("p2") == p2 and rec.get("window_for") == kern:
            return rec.get("hbm_bytes_per_launch"), rec.get("note")
    return None


def pmc_clock(kern, n_local, K, p2):
    for rec in _pmc_records():
        if rec.get("n_local") == n_local and rec.get("K") == K and rec.get("p2") == p2 and rec.get("headline_for") == kern:
            return rec.get("effective_clock_ghz")
    return None


def traced_run(loop, L, ctx, _lib, read_tlog, world, dist, torch, b_iter, b_acc, scr_name, b_scr=None):
    """One complete run from the start centres to dff < Tol with a wall-clock stamp after every iteration's host read
    (max over ranks per iteration) and the two hot kernels' HIP-event times per launch."""
    loop.restart()
    b_scr = b_iter if b_scr is None else b_scr
    _lib.check(L.spkm_timing_log(ctx.handle, 2))
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    stamps = [time.perf_counter()]
    objs = []
    form = []
    while True:
        dff, obj, done = loop.step()
        stamps.append(time.perf_counter())
        objs.append(obj)
        if done:
            break
    its = loop.it
    ms = np.diff(np.array(stamps)) * 1e3
    if world > 1:
        t = torch.tensor(ms, dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        ms = t.cpu().numpy()
    kms = read_tlog(2 * its + 8)
    _lib.check(L.spkm_timing_log(ctx.handle, 0))
    scr = kms[0::2][:its] if kms.size >= 2 * its else np.array([])
    acc = kms[1::2][:its] if kms.size >= 2 * its else np.array([])
    total = float(ms.sum()) * 1e-3
    # (a run's LAST step also fetches that iteration's distances and objective -- once per run, as kmeans_sparsified()
    #  does: the settled figure is taken over the three iterations before it, the fetch reported by itself)
    tail = ms[-min(4, its):-1] if its >= 2 else ms[-1:]
    fetch_ms = float(ms[-1] - tail.mean()) if its >= 2 else None
    r = {"iterations": its, "converged": bool(dff < TOL), "ended_by": "tol" if dff < TOL else "maxiter",
         "final_dff": dff, "final_obj": objs[-1], "seconds": total,
         # one whole run from the cold start: to dff < Tol if `converged`, else capped at MaxIter (then it is a
         # MaxIter-iteration mean, not a time to convergence)
         "whole_run_iters_per_s": its / total,
         "whole_run_mean_ms": float(ms.mean()),
         "cold_no_carry_ms": float(ms[0]),
         # the whole cold iteration against the HBM roofline: SURVEY 8(d)'s bytes of an iteration / its wall time
         "whole_iter_cold": {"bytes": b_iter, "ms": float(ms[0]), "GBs": b_iter / (float(ms[0]) * 1e-3) / 1e9,
                             "frac": b_iter / (float(ms[0]) * 1e-3) / 1e9 / HBM_PEAK_GBS},
         "converged_ms": float(tail.mean()), "converged_iters_per_s": 1e3 / float(tail.mean()),
         "distances_and_objective_once_per_run_ms": fetch_ms,
         # for comparison across rounds only: round 1 timed iterations W+1 .. W+K of ONE run (its --warmup 5 --steps 20 window)
         "r01_window_iters_6_to_25_per_s": (20.0 / (float(ms[5:25].sum()) * 1e-3)) if its >= 25 else None,
         "per_iter_ms": [round(float(v), 2) for v in ms],
         # (own order: the library keeps this shard's points in an order of its own -- data in arbitrary order is regrouped by
         #  cluster inside the library once, spkm_shard_order_info; this run: whether that happened during THIS run)
         "library_order": dict(zip(("own_order", "regrouped_in_this_run"), loop.shard.order_info())) if hasattr(loop, "shard") else None,
         "kernels_ms": {scr_name: [round(float(v), 2) for v in scr], "k_exact_accumulate": [round(float(v), 2) for v in acc]}}
    if scr.size:
        # one kernel per regime, one byte model each: the cold iteration is the assignment kernel over every step,
        # the converged one the exact accumulation pass
        r["roofline_cold_no_carry"] = roofline_obj(scr_name, float(scr[0]), b_scr,
                                                   "plain screen, every 16-point step; " + ("SURVEY 8(d) bytes of an iteration" if b_scr == b_iter
                                                   else "single centroid tile: the bytes the kernel itself moves (f32 / u16 copy + results)"))
        if scr_name.startswith("k_screen_quad") and b_scr == b_iter:
            nn, KK, ss = loop.shard.n, loop.eng.K, int(loop.shard.nnz // max(loop.shard.n, 1))
            clk = pmc_clock(scr_name, nn, KK, loop.shard.p) or 1.79
            floor = (nn / 16.0) * (KK / 32.0) * (((ss + 3) // 4) * 4) * 8.0 * 4.0 / 1024.0 / (clk * 1e9) * 1e3
            r["roofline_cold_no_carry"]["valu_floor_ms"] = floor
            r["roofline_cold_no_carry"]["frac_of_valu_floor"] = floor / float(scr[0]) if scr[0] > 0 else None
            r["roofline_cold_no_carry"]["valu_floor_note"] = (
                f"packed add + fma per (entry, centroid pair) only, 4 issue cycles each, 1024 SIMDs, {clk:.2f} GHz sustained: the "
                "ceiling of a packed-f32 VALU screen (roofline.by_kernel.k_screen_quad.valu_floor)")
        last_pts = loop.eng.exact_pass_points()[1]
        share = last_pts / max(loop.shard.n, 1)
        r["exact_pass_points_share_last_iter"] = share
        r["roofline_converged"] = roofline_obj("k_exact_accumulate", float(acc[-min(4, its):-1].mean()) if its >= 2 else float(acc[-1]), int(b_acc * share),
                                               f"exact confirmation + accumulation pass over the {share:.3f} of the points in clusters "
                                               "that changed; (nnz*(8+2) + n*8 + 16*p*K) bytes scaled by that share")
    loop.restart()
    return r


def dominant_kernel(path, s):
    """Name of the assignment kernel: the f32 screen (4 lanes per point for columns of up to
    64 entries, 16 lanes per point beyond) or, on the all-exact path, the f64 tile kernel."""
    if path != 1:
        return "k_assign_tile"
    return "k_screen_quad" if s <= 64 else "k_screen_tile"


def host_cores():
    """cores this process may run on (the container's share of the box, not the box's total)"""
    try:
        return max(1, len(os.sched_getaffinity(0)))
    except Exception:
        return max(1, os.cpu_count() or 1)


def cpu_sample_arrays(data, centers0, s, n_cpu):
    """host copies of the first n_cpu points (and the start centres) for the CPU legs, taken before the dataset is freed"""
    import torch

    jc = np.arange(0, (n_cpu + 1) * s, s, dtype=np.uint64)
    if "rec" in data:                                   # records: s float64 values, then s 16-bit row ids, per point
        R = data["R"]
        r = data["rec"][: n_cpu * R].view(n_cpu, R)
        x = r[:, : s * 8].contiguous().view(torch.float64).reshape(-1).cpu().numpy()
        ir = r[:, s * 8: s * 10].contiguous().view(torch.int16).reshape(-1).cpu().numpy().astype(np.uint16).astype(np.uint64)
    else:
        ir = data["ir"][: n_cpu * s].cpu().numpy().astype(np.uint16).astype(np.uint64)
        x = data["x"][: n_cpu * s].cpu().numpy()
    return dict(jc=jc, ir=ir, x=x, C0=centers0.cpu().numpy().T.copy(), n=n_cpu)


def _reference_lloyd_iteration(O, p2, K, gamma, s, jc, ir, x, C0, n, chunk=250_000):
    """One iteration with the distance loop run by the REFERENCE's own compiled code (oracle/_ref/libref_sparse.so =
    private/SparseMatrixMinusCluster.c:131-183) and the MATLAB-level steps around it ported: centers/gamma
    (findClusterAssignments.m:76-82), min over the K rows (:169), per-cluster sums / counts and the ML-corrected centres
    (kmeans_sparsified.m:430-448).  Returns seconds spent in (distance loop, everything else)."""
    cg = np.ascontiguousarray((C0 / gamma).T).ravel()                     # p2 x K column-major, true IEEE divide
    a, mind = np.zeros(n, np.int32), np.zeros(n)
    D = np.zeros(min(chunk, n) * K)
    t_dist = t_rest = 0.0
    for lo in range(0, n, chunk):
        m = min(chunk, n - lo)
        jcc = np.ascontiguousarray(jc[lo: lo + m + 1] - jc[lo])
        t0 = time.perf_counter()
        O.ref_dist_csc_flat(p2, m, K, jcc, ir[lo * s: (lo + m) * s], x[lo * s: (lo + m) * s], cg, D[: m * K])
        t1 = time.perf_counter()
        O.lib().orc_min_cols(K, m, D[: m * K], mind[lo: lo + m], a[lo: lo + m])
        t_dist += t1 - t0
        t_rest += time.perf_counter() - t1
    t0 = time.perf_counter()
    S, Cnt, nk = O.accumulate(p2, n, K, jc[: n + 1], ir[: n * s], x[: n * s], a)
    O.finalize_centers(S, Cnt, nk, gamma, C0)
    t_rest += time.perf_counter() - t0
    return t_dist, t_rest, a


def cpu_baseline(cd, p2, K, gamma, s, n_one, n_total):
    """The reference's CPU path on the host's cores, on a bounded sample of the same workload, scaled linearly to N:
      * `value`, cores = 1, kind "reference": the distance loop is the reference's own C (`switch (K)`,
        private/SparseMatrixMinusCluster.c:131-183, single-threaded as in the reference; built into oracle/_ref by
        oracle/Makefile in the build container and travelled here as a binary), the MATLAB-level steps around it are
        ported (`kind_detail`).  Without that binary: kind "port", `orc_lloyd` (oracle/orc_sparse.c) as in rounds 1-3;
      * `port_one_thread`: orc_lloyd on the same sample (the restatement; agrees with the reference leg bit for bit);
      * `all_cores`: the same iteration with the points column-partitioned over every host core the way the reference's
        one threaded mex partitions its columns (hadamard_pthreads.c:121-204) -- the box-level baseline;
      * `fwht`: the reference's own threaded FWHT (hadamard_pthreads.c:57-119 behind oracle/_ref) when it travelled."""
    from oracle import oracle as O

    O.lib()
    jc, ir, x, C0 = cd["jc"], cd["ir"], cd["x"], cd["C0"]
    t0 = time.perf_counter()
    port = O.lloyd(p2, n_one, jc[: n_one + 1], ir[: n_one * s], x[: n_one * s], C0, gamma, maxiter=1, tol=0.0)
    dt = time.perf_counter() - t0
    port_leg = {"value": 1.0 / (dt * n_total / n_one), "unit": "Lloyd iters/sec", "cores": 1,
                "sample": f"1 Lloyd iteration of oracle/orc_sparse.c orc_lloyd (gcc -O, single thread) on the first "
                          f"{n_one} points of the same dataset in {dt:.2f} s, scaled linearly to N={n_total}"}
    out = dict(port_leg, kind="port")
    if O.ref_available("sparse"):
        try:
            td, tr, a_ref = _reference_lloyd_iteration(O, p2, K, gamma, s, jc, ir, x, C0, n_one)
            out = {"value": 1.0 / ((td + tr) * n_total / n_one), "unit": "Lloyd iters/sec", "cores": 1, "kind": "reference",
                   "kind_detail": "reference C (private/SparseMatrixMinusCluster.c:131-183 compiled with setup_kmeans.m:19's "
                                  "-O, oracle/_ref/libref_sparse.so) for the distance loop + port of the MATLAB steps "
                                  "(centers/gamma, min, sums / counts, ML-corrected centres: oracle/orc_sparse.c)",
                   "sample": f"1 Lloyd iteration on the first {n_one} points of the same dataset: {td:.2f} s in the "
                             f"reference's distance loop + {tr:.2f} s in the ported steps, single thread, scaled linearly "
                             f"to N={n_total}",
                   "distance_loop_share": td / (td + tr),
                   "assignments_equal_port": bool(np.array_equal(a_ref, port["assign"])),
                   "port_one_thread": port_leg}
        except Exception as e:  # the baseline is a report, never a reason to lose the bench line
            out["reference_leg_error"] = str(e)
    out["host_cpus"], out["host_cpus_usable"] = os.cpu_count(), host_cores()
    try:
        threads = host_cores()
        n_all = cd["n"]
        t0 = time.perf_counter()
        O.lloyd_iter_threads(p2, n_all, jc[: n_all + 1], ir[: n_all * s], x[: n_all * s], C0, gamma, threads)
        dta = time.perf_counter() - t0
        out["all_cores"] = {"value": 1.0 / (dta * n_total / n_all), "unit": "Lloyd iters/sec", "cores": threads,
                            "sample": f"1 Lloyd iteration of orc_lloyd_iter_threads (points column-partitioned over "
                                      f"{threads} pthreads as hadamard_pthreads.c:121-204 partitions its columns) on the first "
                                      f"{n_all} points in {dta:.2f} s, scaled linearly to N={n_total}"}
    except Exception as e:  # the baseline is a report, never a reason to lose the bench line
        out["all_cores"] = {"error": str(e)}
    # the preconditioner's CPU path: the reference's one multi-threaded mex (private/hadamard_pthreads.c, static
    # column partition over NTHREADS = maxNumCompThreads(), setup_kmeans.m:45) -- its own worker + kernels (:57-119)
    # from oracle/_ref when the binary travelled (our restatement of the partition around them), else orc_fwht_threads
    try:
        threads = max(1, min(host_cores(), 64))
        cols = 32768
        xin = np.random.default_rng(0).standard_normal(cols * p2)
        yout = np.zeros_like(xin)
        if O.ref_available("pthreads"):
            fn, what, kind = O._ref("pthreads").ref_hadamard_pthreads, "private/hadamard_pthreads.c:57-119 (oracle/_ref)", "reference"
        else:
            fn, what, kind = O.lib().orc_fwht_threads, "oracle/orc_fwht.c orc_fwht_threads", "port"
        fn(p2, cols, xin, yout, threads)          # warm-up (thread creation, page faults)
        t0 = time.perf_counter()
        fn(p2, cols, xin, yout, threads)
        dt = time.perf_counter() - t0
        out["fwht"] = {"columns_per_s": cols / dt, "threads": threads, "m": p2, "kind": kind,
                       "sample": f"{cols} columns of length {p2}, {what}"}
    except Exception as e:
        out["fwht"] = {"error": str(e)}
    return out


if __name__ == "__main__":
    main()
