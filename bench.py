#!/usr/bin/env python3
"""Headline benchmark: YOLACT post-process images/sec (decode + Fast NMS + masks), 550x550, batch 32.

    python bench.py --gpus N --steps K --warmup W [--config 2|3|4|5] [--scaling weak|strong]

A "step" is one pass of the whole hot path (yp_run: score_filter -> nms (+ mask zero-fill) ->
mask_assemble) over one batch of synthetic head outputs already resident in HBM.

Multi-GPU: one process per GPU, no data-path collective (images are independent,
detection.py:71); the barrier and the MAX over ranks are the only communication (RCCL).  The ranks
come from the driver (`python -m torch.distributed.run ... bench.py --gpus N`, WORLD_SIZE set) or,
when `--gpus N` is given without WORLD_SIZE, from this script itself: the parent starts N rank
processes BEFORE anything touches the GPU, relays rank 0's line and exits with their status.
  --scaling weak    every rank post-processes its own batch of `cfg.batch` images
  --scaling strong  ONE batch of `cfg.batch` images cut into contiguous chunks (sharding.shard_bounds),
                    the partitioning BASELINE.json's configs 3 and 5 name (default for those)
Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
METRIC = "post-process images/sec (decode+FastNMS+masks) 550x550, bs32; % HBM roofline"


def algorithmic_bytes(A, C, P, K, S, N):
    """bytes(img) = 4AC + 16A + 4KS + 4PK + N(4P+24); priors once per batch."""
    S, N = np.asarray(S, np.float64), np.asarray(N, np.float64)
    per_img = 4.0 * A * C + 16.0 * A + 4.0 * K * S + 4.0 * P * K + N * (4.0 * P + 24.0)
    return float(per_img.sum() + 16.0 * A)


def kernel_bytes(A, C, P, K, S, N, nms_mode):
    """Algorithmic bytes per launch of each kernel."""
    B = len(N)
    S, N = np.asarray(S, np.float64), np.asarray(N, np.float64)
    return {
        # logits once; per candidate (counted as one per surviving anchor): its prior and offsets gathered (16 + 16)
        # and its key and decoded box written to the image's lists (8 + 16)
        "score_filter": float(B * 4.0 * A * C + (S * (16 + 16 + 8 + 16)).sum()),
        # candidate keys + boxes read once; per kept detection its output row, crop edges and anchor (24 + 16 + 4);
        # the zero-fill of the mask rows rides in this launch: every mask row written once
        "nms": float((S * (8 + 16)).sum() + (N * (24 + 16 + 4)).sum() + (N * 4.0 * P).sum()),
        # prototypes once + per kept detection its coefficient row, crop edges and anchor; the pixels inside the crop
        # boxes are not priced (data dependent)
        "mask_assemble": float(B * 4.0 * P * K + (N * (128 + 16 + 4)).sum()),
    }


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", type=int, default=2, help="BASELINE.json config number (2 = headline)")
    ap.add_argument("--nms-mode", default="per_class", choices=["per_class", "cc"])
    ap.add_argument("--batch", type=int, default=None, help="images per batch (default: the config's)")
    ap.add_argument("--scaling", default="auto", choices=["auto", "weak", "strong"],
                    help="auto: strong for configs 3 and 5 (one batch of 64 sharded over the GPUs), weak otherwise")
    ap.add_argument("--rotate-outputs", type=int, default=3,
                    help="output sets written in turn by consecutive steps (a caller that consumes its results "
                         "does not rewrite one buffer that lives in the 256 MiB Infinity Cache); `value` is measured "
                         "this way, `value_same_buffer` with one set")
    ap.add_argument("--rotate-inputs", type=int, default=None,
                    help="input sets read in turn by consecutive steps (default: as many as output sets): config 2's "
                         "200 MB of logits would otherwise be re-read from the 256 MiB Infinity Cache; `value` rotates both")
    ap.add_argument("--box-coverage", type=float, default=None,
                    help="scale the box-size offsets (dw, dh) of the synthetic batch until the kept detections' crop boxes "
                         "cover about this fraction of the proto pixels each (the headline distribution, SURVEY 8d, gives "
                         "~0.01; real detections are larger).  Never the headline: reported under its own workload name")
    ap.add_argument("--launch-timeout", type=float, default=900.0, help="self-launched ranks: overall limit in seconds")
    ap.add_argument("--settle", type=int, default=100,
                    help="untimed steps before the W warm-up steps: the GPU needs ~10 ms of work to reach its steady clocks "
                         "(measured: with 5 warm-up steps the 20 timed steps that follow run 3 %% slower than after 100); "
                         "reported in the line as config.settle_steps")
    ap.add_argument("--skip-same-buffer", action="store_true",
                    help="profiling runs: leave out the second timed region (one output set), so that every launch a "
                         "profiler sees belongs to the headline measurement")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-threads", type=int, default=16)
    ap.add_argument("--reference-layout", action="store_true",
                    help="the reference as shipped (detection.py:21-22,83,181): cross-class NMS, top_k 15, "
                         "300 zero-padded rows per image")
    ap.add_argument("--kept-outputs", action="store_true",
                    help="the output sets are tracked by the library (yp_run_kept; what Detect(reuse_outputs=True) uses): "
                         "every step clears the crop boxes the step before wrote into its set instead of refilling every "
                         "row.  Never the headline (yp_run's contract is caller-owned buffers in an unknown state): "
                         "reported under its own workload name; the default run reports it as value_kept_outputs")
    return ap.parse_args(argv)


def launch_ranks(args):
    """--gpus N without WORLD_SIZE: start the N ranks here.  The parent never imports torch or touches
    the GPU; every rank is a fresh child process (nothing is exec'ed from a process that holds a GPU).
    The children are polled: when one exits with an error, or the overall limit passes, the others - which would
    sit in their barrier for ever - are terminated (these exact child processes, by handle), and the launcher
    exits non-zero.  Returns 0 only if every rank did."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True))
    deadline = time.monotonic() + args.launch_timeout
    failed = None
    while True:
        rcs = [p.poll() for p in procs]
        bad = [i for i, rc in enumerate(rcs) if rc not in (None, 0)]
        if bad:
            failed = f"rank {bad[0]} exited with status {rcs[bad[0]]}"
        elif time.monotonic() > deadline:
            failed = f"no result within {args.launch_timeout:.0f} s"
        if failed or all(rc == 0 for rc in rcs):
            break
        time.sleep(0.2)
    if failed:
        for p in procs:
            if p.poll() is None:
                p.terminate()
        for p in procs:
            try:
                p.wait(timeout=10)
            except subprocess.TimeoutExpired:
                p.kill()
        sys.stderr.write(f"bench.py: {failed}; the other ranks were stopped\n")
        return 1
    out0 = procs[0].stdout.read()
    sys.stdout.write(out0)
    sys.stdout.flush()
    return 0


def main():
    args = parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: refusing to benchmark the wrong world")
    if os.environ.get("YP_BENCH_DIE_RANK") == str(rank):      # launcher rehearsal: this rank fails before the rendezvous
        sys.exit(3)

    from yolact_amd import synth
    from yolact_amd.sharding import shard_bounds

    cfg = synth.CONFIGS[args.config]
    scaling = args.scaling if args.scaling != "auto" else ("strong" if args.config in (3, 5) else "weak")
    B_total = args.batch or cfg.batch
    if scaling == "strong":
        lo, hi = shard_bounds(B_total, world, rank)       # one batch, contiguous chunks
    else:
        lo, hi = rank * B_total, (rank + 1) * B_total     # every rank its own batch
    B = hi - lo
    images_per_step = B_total if scaling == "strong" else B_total * world

    import torch
    import torch.distributed as dist

    # Rehearsal knobs (the driver's runs use none of them): YP_BENCH_BACKEND=gloo instead of RCCL (RCCL
    # refuses two ranks on one GPU), YP_BENCH_ONE_DEVICE=1 puts every rank on device 0, YP_BENCH_DRY=1
    # skips the GPU work altogether (launcher / rendezvous / sharding check on a CPU-only box; the line
    # says "dry_run": true and carries no measurement).
    backend = os.environ.get("YP_BENCH_BACKEND", "nccl")
    dry = bool(os.environ.get("YP_BENCH_DRY"))
    if os.environ.get("YP_BENCH_ONE_DEVICE"):
        local_rank = 0
    dev = None
    if not dry:
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs a HIP device (the hot path has no CPU fallback)")
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    def barrier():
        if dev is not None:
            torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        if dev is not None:
            torch.cuda.synchronize(dev)

    def max_over_ranks(x):
        if world == 1:
            return x
        t = torch.tensor([x], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    top_k, max_dets, zero_pad = cfg.top_k, cfg.max_dets, False
    if args.reference_layout:
        args.nms_mode, top_k, max_dets, zero_pad = "cc", 15, 300, True

    base = {
        "metric": METRIC, "value": 0.0, "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 0.0, "higher_is_better": True, "scaling": scaling, "vs_baseline": None, "dtype": "f32",
        "data": "synthetic",
    }

    if dry:
        barrier()
        elapsed = max_over_ranks(1.0)
        chunks = [None] * world
        if world > 1:
            dist.all_gather_object(chunks, (lo, hi))
        else:
            chunks = [(lo, hi)]
        if rank == 0:
            base.update({"dry_run": True, "config": {"workload": f"DRY RUN config {args.config}", "chunks": chunks,
                                                     "images_per_step": images_per_step}, "elapsed": elapsed})
            print(json.dumps(base), flush=True)
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return

    from yolact_amd import PostConfig, PostProcessor

    if B < 1:
        raise SystemExit(f"rank {rank}: empty shard (batch {B_total} over {world} ranks)")
    net, shp = synth.make_batch(cfg, batch=B, first_image=lo)
    A, C = net["pred_cls"].shape[1:]
    ph, pw = net["proto_out"].shape[1:3]
    P, K = ph * pw, 32
    pc = PostConfig(A, C, ph, pw, cfg.img_h, cfg.img_w, conf_thresh=cfg.conf_thresh, nms_thresh=cfg.nms_thresh,
                    nms_mode=args.nms_mode, top_k=top_k, max_dets=max_dets, zero_pad_masks=zero_pad, max_batch=B,
                    device_id=local_rank)
    pp = PostProcessor(pc)
    NR = max(1, args.rotate_outputs)
    NI = max(1, args.rotate_inputs if args.rotate_inputs is not None else NR)
    outs = [pp.alloc_kept(B) if args.kept_outputs else pp.alloc_outputs(B) for _ in range(NR)]

    def upload(n):
        dd = {k: torch.from_numpy(v).to(dev) for k, v in n.items()}
        return (dd["pred_offset"], dd["pred_cls"], dd["pred_mask_coef"], dd["priors"], dd["proto_out"])

    def coverage(out):
        """Mean share of the proto pixels inside a kept detection's crop box (padding included: utils.crop)."""
        bx = out["detection_boxes"].cpu().numpy().astype(np.float64)
        nd = out["num_detections"].cpu().numpy()
        fr = []
        for b in range(bx.shape[0]):
            q = bx[b, :nd[b]]
            y1 = np.clip(np.minimum(q[:, 0], q[:, 2]) * ph - 1, 0, ph); y2 = np.clip(np.maximum(q[:, 0], q[:, 2]) * ph + 1, 0, ph)
            x1 = np.clip(np.minimum(q[:, 1], q[:, 3]) * pw - 1, 0, pw); x2 = np.clip(np.maximum(q[:, 1], q[:, 3]) * pw + 1, 0, pw)
            fr.append(((y2 - y1) * (x2 - x1) / (ph * pw)))
        fr = np.concatenate(fr) if fr else np.zeros(0)
        return float(fr.mean()) if fr.size else 0.0

    box_scale = 1.0
    if args.box_coverage is not None:
        # decoded box sides are prior * exp(0.2 * dw) (utils/utils.py:202): adding ln(s) / 0.2 to dw and dh scales every
        # box by s.  A few runs settle s (larger boxes suppress each other more, so the kept set changes with s).
        base_off = net["pred_offset"].copy()
        for _ in range(6):
            net["pred_offset"] = base_off.copy()
            net["pred_offset"][..., 2:] += np.float32(np.log(box_scale) / 0.2)
            o = pp.run(*upload(net), out=outs[0]); pp.sync()
            got = coverage(o)
            if got <= 0 or abs(got / args.box_coverage - 1.0) < 0.05:
                break
            box_scale *= float(np.sqrt(args.box_coverage / got))
    # further input sets hold OTHER images of the same generator (and the same box scaling): a tracked output set then
    # finds other boxes than the ones it has to clear, as in a real stream of batches.  The statistics and the parity
    # report below come from a final run on set 0.
    def input_set(j):
        if j == 0:
            return net
        n2, _ = synth.make_batch(cfg, batch=B, first_image=lo + 4096 * j)
        if args.box_coverage is not None:
            n2["pred_offset"][..., 2:] += np.float32(np.log(box_scale) / 0.2)
        return n2
    ins_sets = [upload(input_set(j)) for j in range(NI)]
    ins = ins_sets[0]

    def timed(nsets, nin, sets=None, shift=False):
        sets = outs if sets is None else sets
        # shift: an output set meets a different input set every time it comes round (tracked sets: the boxes of
        # the step before must differ from this step's, or there is nothing to clear)
        pick = (lambda i: (i + i // nsets) % nin) if shift else (lambda i: i % nin)
        for i in range(args.warmup):
            pp.run(*ins_sets[pick(i)], out=sets[i % nsets])
        barrier()
        t0 = time.perf_counter()
        for i in range(args.steps):
            pp.run(*ins_sets[pick(i)], out=sets[i % nsets])
        barrier()
        return max_over_ranks(time.perf_counter() - t0)

    pp.profile(False)
    for i in range(max(0, args.settle)):      # clocks, TLBs, the allocator: not a measurement
        pp.run(*ins_sets[i % NI], out=outs[i % NR])
    elapsed = timed(NR, NI, shift=args.kept_outputs)   # the headline: consecutive steps read different input sets and write different output sets
    # Informational: the same steps on output sets the library tracks (yp_run_kept).  Needs >= 2 input sets.
    elapsed_kept = None
    if world == 1 and not args.skip_same_buffer and not args.kept_outputs and NI >= 2:
        ksets = [pp.alloc_kept(B) for _ in range(NR)]
        elapsed_kept = timed(NR, NI, sets=ksets, shift=True)
        for ks in ksets:
            ks.close()
        del ksets
    elapsed_same = elapsed if (args.skip_same_buffer or (NR == 1 and NI == 1)) else timed(1, 1)   # one set each: they live in the Infinity Cache

    # Informational: two handles on two streams taking the steps in turn (what a serving loop with two batches
    # in flight sees: the latency-bound launches of one batch beside the bandwidth-bound ones of the other).
    # Same steps, same rotating output sets per handle; never the headline.
    elapsed_two = None
    if world == 1 and not args.skip_same_buffer:
        pp2 = PostProcessor(pc)
        outs2 = [pp2.alloc_outputs(B) for _ in range(NR)]
        lanes = [(pp, outs, torch.cuda.Stream(device=dev)), (pp2, outs2, torch.cuda.Stream(device=dev))]

        def step2(i):
            h, o, st = lanes[i & 1]
            with torch.cuda.stream(st):
                h.run(*ins_sets[i % NI], out=o[(i >> 1) % NR])
        for i in range(2 * args.warmup):
            step2(i)
        torch.cuda.synchronize(dev)
        t1 = time.perf_counter()
        for i in range(args.steps):
            step2(i)
        torch.cuda.synchronize(dev)
        elapsed_two = time.perf_counter() - t1
        del pp2, outs2, lanes

    # Per-kernel durations: same launches, HIP events recorded on the launch stream inside yp_run and read
    # back after every step (host-serialised, so this region is NOT the one `value` comes from).
    pp.profile(True)
    acc = {}
    barrier()
    t2 = time.perf_counter()
    for i in range(args.steps):
        pp.run(*ins_sets[i % NI], out=outs[i % NR])
        for k, v in pp.kernel_times_ms().items():
            acc[k] = acc.get(k, 0.0) + v
    barrier()
    ms_per_step_events = (time.perf_counter() - t2) / args.steps * 1e3
    kern_ms = {k: v / args.steps for k, v in acc.items()}
    pp.profile(False)

    out = pp.run(*ins_sets[0], out=outs[0])
    pp.sync()
    # The gather north_star names ("gathered on the host"): this rank's outputs copied to PINNED host buffers, one
    # synchronisation per shard.  Not part of `value` (the metric is device-resident in and out); reported
    # beside it so that a sharded caller can see what the host side of a step costs.
    gather_ms = None
    if world > 1 or scaling == "strong":
        host = {k: torch.empty(v.shape, dtype=v.dtype, pin_memory=True) for k, v in out.items() if v is not None}
        reps = 3
        barrier()
        t3 = time.perf_counter()
        for _ in range(reps):
            for k, hv in host.items():
                hv.copy_(out[k], non_blocking=True)
            torch.cuda.current_stream(dev).synchronize()
        gather_ms = max_over_ranks((time.perf_counter() - t3) / reps * 1e3)
        del host
    S = pp.last_survivors(B)
    N = out["num_detections"].cpu().numpy()
    rows = np.full_like(N, max_dets) if zero_pad else N     # mask rows actually written (tf.zeros rows included)
    total_bytes = algorithmic_bytes(A, C, P, K, S, rows)
    kb = kernel_bytes(A, C, P, K, S, rows, args.nms_mode)
    dominant = max(kern_ms, key=lambda k: kern_ms[k])
    ach = kb[dominant] / (kern_ms[dominant] * 1e-3) / 1e9
    ms_per_step = elapsed / args.steps * 1e3

    shard = f"one batch of {B_total} sharded over {world} GPU(s), {B} on rank 0" if scaling == "strong" \
        else f"batch {B} per GPU"
    result = dict(base)
    result.update({
        "value": round(images_per_step * args.steps / elapsed, 1),
        "ms_per_step": round(ms_per_step, 4),
        "value_same_buffer": round(images_per_step * args.steps / elapsed_same, 1),
        "ms_per_step_same_buffer": round(elapsed_same / args.steps * 1e3, 4),
        "value_two_streams": round(images_per_step * args.steps / elapsed_two, 1) if elapsed_two else None,
        "value_kept_outputs": round(images_per_step * args.steps / elapsed_kept, 1) if elapsed_kept else None,
        "ms_per_step_kept_outputs": round(elapsed_kept / args.steps * 1e3, 4) if elapsed_kept else None,
        "kept_outputs_note": "informational, never the headline: yp_run_kept on library-tracked output sets - each step clears "
                             "the crop boxes the step before wrote into its set instead of zero-filling every mask row; same "
                             "results bit for bit (tests/test_gpu_parity.py::test_tracked_output_set_equals_fresh_runs)",
        "config": {
            "workload": f"BASELINE config {args.config}: {cfg.img_h}x{cfg.img_w}, {A} anchors, {C} classes, "
                        f"proto {ph}x{pw}x32, {shard}, conf {cfg.conf_thresh}, top_k {top_k}, "
                        f"max {max_dets} dets, nms_mode {args.nms_mode}" + (", 300 zero-padded rows" if zero_pad else "")
                        + (", library-tracked output sets (yp_run_kept)" if args.kept_outputs else "")
                        + (f", boxes scaled x{box_scale:.2f} for --box-coverage {args.box_coverage}" if args.box_coverage is not None else ""),
            "batch_per_gpu": B, "images_per_step": images_per_step, "nms_mode": args.nms_mode,
            "rotate_outputs": NR, "rotate_inputs": NI, "settle_steps": max(0, args.settle), "mean_survivors": float(np.mean(S)), "mean_dets": float(np.mean(N)),
            "box_coverage": round(coverage(out), 4),
        },
        "roofline": {
            "bound": "hbm", "kernel": dominant,
            "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": None, "traffic_source": None,
            "bytes_per_launch": kb[dominant], "avg_launch_ms": round(kern_ms[dominant], 5),
            "timing": "HIP events on the launch stream, this run",
            # north_star: "MFMA utilisation for mask assembly is reported"
            "mfma_util": 0.0,
            "mfma_note": "no kernel of the shipped path issues MFMA. Measured, not argued (profiles/r03/gc_mfma_experiment.md): the one "
                         "GEMM-shaped piece, F4's grad_coef ([rows x pixels] x [pixels x 32]), was built on v_mfma_f32_32x32x2_f32 behind "
                         "-DYP_EXP_GC_MFMA and took 177.5 us against the fma-chain kernel's 72.5 us at 13 % box coverage, matrix cores "
                         "busy 6.8 % of the SIMD cycles - the A operand has to be laid out dense and two thirds of it are zeros between "
                         "the crop boxes. mask_assemble's K=32 contraction has the same shape problem (--box-coverage 0.13: 43 us, "
                         "still latency/HBM-bound) and a fixed K-ordered fma chain as its arithmetic contract",
        },
        "gather_ms": None if gather_ms is None else round(gather_ms, 3),
        "gather_note": None if gather_ms is None else "this rank's outputs device -> pinned host, max over ranks; not part of value",
        "kernels_ms": {k: round(v, 5) for k, v in kern_ms.items()},
        "ms_per_step_with_events": round(ms_per_step_events, 4),
        "kernels_gbs": {k: round(kb[k] / (kern_ms[k] * 1e-3) / 1e9, 1) if kern_ms[k] > 0 else 0.0 for k in kern_ms},
        "pipeline": {
            "bytes_per_step": total_bytes,
            "achieved_gbs": round(total_bytes / (ms_per_step * 1e-3) / 1e9, 1),
            "frac_of_hbm_peak": round(total_bytes / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
        },
    })

    # Figures that this run did NOT measure: HBM bytes per launch from the PMC passes and per-kernel
    # durations from rocprofv3 --kernel-trace of the same command, as committed under profiles/.
    if args.config == 2 and B == cfg.batch and world == 1 and args.box_coverage is None:
        try:
            tj = json.load(open(os.path.join(ROOT, "profiles", "hbm_traffic.json")))
            key = f"{args.nms_mode}:{dominant}"
            if key in tj:
                result["roofline"]["traffic"] = tj[key]
                result["roofline"]["traffic_source"] = "profiles/hbm_traffic.json (committed rocprofv3 --pmc passes, not this run)"
        except Exception:
            pass
        try:
            rj = json.load(open(os.path.join(ROOT, "profiles", "r03", f"rocprof_summary_{args.nms_mode}.json")))
            rk = {k: v for k, v in rj.get("kernels_us", {}).items()}
            result["rocprof"] = {
                "source": f"profiles/r03/rocprof_summary_{args.nms_mode}.json (committed rocprofv3 --kernel-trace --stats, not this run)",
                "kernels_us": rk,
                "frac_of_hbm_peak": {k: round(kb[k] / (rk[k] * 1e-6) / 1e9 / HBM_PEAK_GBS, 4) for k in kb if rk.get(k)},
            }
        except Exception:
            pass
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline(net, shp, cfg, args, top_k, max_dets)
        result["parity"] = parity_report(net, shp, cfg, args, top_k, max_dets, {k: v.cpu().numpy() for k, v in out.items()
                                                                                  if k != "detection_masks"})
    if rank == 0:
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def parity_report(net, shp, cfg, args, top_k, max_dets, got):
    """Part of the cpu_baseline leg (the oracle as the checker, never the thing measured): the timed
    batch's kept detections against (i) the C oracle, which shares the kernels' arithmetic contract
    - expected bit-exact - and (ii) the numpy oracle with libm exp / numpy reductions, the stand-in
    for the reference's TF CPU arithmetic (north_star: >= 95 % kept-index agreement)."""
    from oracle import c_oracle, detect_oracle
    kw = dict(conf_thresh=cfg.conf_thresh, nms_thresh=cfg.nms_thresh, top_k=top_k, nms_mode=args.nms_mode, max_dets=max_dets)
    nb = min(16, net["pred_cls"].shape[0])
    sub = {k: (v[:nb] if k != "priors" else v) for k, v in net.items()}
    ref_c = c_oracle.detect(sub, shp, nthreads=max(1, min(args.cpu_threads, os.cpu_count() or 1)), **kw)
    ref_n = detect_oracle.detect(sub, shp, max_output_size=max_dets, **kw)

    def rates(ref):
        rows_same = rows_tot = sets = 0
        for b in range(nb):
            n, m = int(ref["num_detections"][b]), int(got["num_detections"][b])
            a = list(zip(got["detection_anchor_indices"][b, :m].tolist(), got["detection_classes"][b, :m].tolist()))
            d = list(zip(ref["detection_anchor_indices"][b, :n].tolist(), ref["detection_classes"][b, :n].tolist()))
            rows_same += sum(x == y for x, y in zip(a, d)); rows_tot += max(n, m); sets += set(a) == set(d)
        return round(rows_same / max(1, rows_tot), 4), round(sets / nb, 4)

    rc, rn = rates(ref_c), rates(ref_n)
    rows = min(got["detection_scores"].shape[1], ref_c["detection_scores"].shape[1])
    return {
        "images": nb,
        "vs_c_oracle_same_contract": {"kept_index_agreement": rc[0], "kept_set_agreement": rc[1],
                                      "scores_bit_exact": bool(np.array_equal(got["detection_scores"][:nb, :rows], ref_c["detection_scores"][:, :rows])),
                                      "boxes_bit_exact": bool(np.array_equal(got["detection_boxes"][:nb, :rows], ref_c["detection_boxes"][:, :rows]))},
        "vs_numpy_oracle_libm": {"kept_index_agreement": rn[0], "kept_set_agreement": rn[1],
                                 "note": "row-wise; rows differ only where two scores are within an ulp"},
    }


def cpu_baseline(net, shp, cfg, args, top_k, max_dets):
    """The oracle timed on this host's cores - the checker, timed as a baseline, never the thing
    shipped.  The reference itself (TF2) cannot run here.  Two legs:
    the C port with OpenMP over images (the headline `value`), and the numpy op-for-op restatement
    with the reference's structure (whole-batch softmax, then a serial per-image loop; BLAS threads as
    the host gives them) on a bounded sample."""
    from oracle import c_oracle, detect_oracle
    c_oracle.lib()
    host_cores = os.cpu_count() or 1
    threads = max(1, min(args.cpu_threads, host_cores))
    B = net["pred_cls"].shape[0]
    kw = dict(conf_thresh=cfg.conf_thresh, nms_thresh=cfg.nms_thresh, top_k=top_k, nms_mode=args.nms_mode,
              max_dets=max_dets)
    n1 = min(4, B)
    sub = {k: (v[:n1] if k != "priors" else v) for k, v in net.items()}
    t0 = time.perf_counter()
    c_oracle.detect(sub, shp, nthreads=1, **kw)
    t_single = time.perf_counter() - t0
    c_oracle.detect(net, shp, nthreads=threads, **kw)      # untimed: thread pool start, first touch of the buffers
    reps, t_multi = 0, 0.0
    t0 = time.perf_counter()
    while True:                                            # ~10 s of CPU work, bounded both ways
        c_oracle.detect(net, shp, nthreads=threads, **kw)
        reps += 1
        t_multi = time.perf_counter() - t0
        if t_multi > 10.0 or reps >= 400:
            break
    nn = min(8, B)
    subn = {k: (v[:nn] if k != "priors" else v) for k, v in net.items()}
    t0 = time.perf_counter()
    detect_oracle.detect(subn, shp, max_output_size=max_dets, **kw)
    t_numpy = time.perf_counter() - t0
    return {
        "value": round(B * reps / t_multi, 2), "unit": "images/s", "cores": threads, "kind": "port",
        "host_cores": host_cores, "threads": threads,
        "sample": f"{reps} x the same {B}-image batch, OpenMP over images, {threads} threads; "
                  f"single-thread: {n1} images in {t_single:.2f}s",
        "single_thread_value": round(n1 / t_single, 2),
        "numpy_restatement": {
            "value": round(nn / t_numpy, 2), "unit": "images/s", "kind": "port",
            "sample": f"{nn} images once: whole-batch softmax, then the serial per-image loop of detection.py:71 "
                      f"(numpy + BLAS, threads as the host gives them)",
        },
    }


if __name__ == "__main__":
    main()
