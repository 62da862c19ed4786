#!/usr/bin/env python3
"""Headline benchmark: YOLACT post-process images/sec (decode + Fast NMS + masks), 550x550, batch 32.

    python bench.py --gpus N --steps K --warmup W

A "step" is one pass of the whole hot path (yp_run: score_filter -> segment_nms -> final_select ->
mask_assemble) over one batch of synthetic head outputs already resident in HBM.  For N > 1 the
driver launches one rank per GPU (torch.distributed.run); every rank post-processes its own batch
(images are independent, detection.py:71: no data-path collective), the barrier and the MAX over
ranks use RCCL.  Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def algorithmic_bytes(cfg, A, C, P, K, S, N):
    """bytes(img) = 4AC + 16A + 4KS + 4PK + N(4P+24); priors once per batch."""
    S, N = np.asarray(S, np.float64), np.asarray(N, np.float64)
    per_img = 4.0 * A * C + 16.0 * A + 4.0 * K * S + 4.0 * P * K + N * (4.0 * P + 24.0)
    return float(per_img.sum() + 16.0 * A)


def kernel_bytes(A, C, P, K, S, N, nms_mode):
    """Algorithmic bytes per launch of each kernel."""
    B = len(N)
    S, N = np.asarray(S, np.float64), np.asarray(N, np.float64)
    return {
        # logits once; offsets+priors read, box+key written for survivors only
        "score_filter": float(B * 4.0 * A * C + (S * (16 + 16 + 16 + 8)).sum()),
        # candidate keys + boxes of survivors, coefficient rows of kept detections, output rows
        "segment_nms": float((S * (8 + 16)).sum() + (N * (128 + 128 + 16 + 24)).sum()),
        "final_select": 0.0 if nms_mode == "cc" else float((N * (8 + 16 + 128 + 128 + 16 + 24)).sum()),
        # prototypes once + every mask row written once + coefficient/crop rows
        "mask_assemble": float(B * 4.0 * P * K + (N * (4.0 * P + 128 + 16)).sum()),
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", type=int, default=2, help="BASELINE.json config number (2 = headline)")
    ap.add_argument("--nms-mode", default="per_class", choices=["per_class", "cc"])
    ap.add_argument("--batch", type=int, default=None)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-threads", type=int, default=16)
    ap.add_argument("--reference-layout", action="store_true",
                    help="the reference as shipped (detection.py:21-22,83,181): cross-class NMS, top_k 15, "
                         "300 zero-padded rows per image")
    ap.add_argument("--inflight", type=int, default=1,
                    help="batches in flight: N handles on N HIP streams taking the steps in turn")
    ap.add_argument("--timed-events", type=int, default=0,
                    help="1: the timed region records the per-kernel HIP events too (costs ~4%%); "
                         "0 (default): `value` is timed without events and the same K steps are then "
                         "repeated with events for the per-kernel durations")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (the hot path has no CPU fallback)")
    # Rehearsal knobs for a one-GPU box (the driver's multi-GPU runs use neither): all ranks on one
    # device, and gloo instead of RCCL (RCCL refuses two ranks on one GPU).
    backend = os.environ.get("YP_BENCH_BACKEND", "nccl")
    if os.environ.get("YP_BENCH_ONE_DEVICE"):
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    from yolact_amd import PostConfig, PostProcessor, synth

    cfg = synth.CONFIGS[args.config]
    B = args.batch or cfg.batch
    net, shp = synth.make_batch(cfg, batch=B, first_image=rank * B)
    A, C = net["pred_cls"].shape[1:]
    ph, pw = net["proto_out"].shape[1:3]
    P, K = ph * pw, 32
    top_k, max_dets, zero_pad = cfg.top_k, cfg.max_dets, False
    if args.reference_layout:
        args.nms_mode, top_k, max_dets, zero_pad = "cc", 15, 300, True
    pc = PostConfig(A, C, ph, pw, cfg.img_h, cfg.img_w, conf_thresh=cfg.conf_thresh, nms_thresh=cfg.nms_thresh,
                    nms_mode=args.nms_mode, top_k=top_k, max_dets=max_dets, zero_pad_masks=zero_pad, max_batch=B,
                    device_id=local_rank)
    pp = PostProcessor(pc)
    d = {k: torch.from_numpy(v).to(dev) for k, v in net.items()}
    out = pp.alloc_outputs(B)
    ins = (d["pred_offset"], d["pred_cls"], d["pred_mask_coef"], d["priors"], d["proto_out"])

    def barrier():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    pp.profile(bool(args.timed_events))   # HIP events around every kernel, recorded on the launch stream
    NF = max(1, args.inflight)
    pps = [pp] + [PostProcessor(pc) for _ in range(NF - 1)]
    outs = [out] + [pp.alloc_outputs(B) for _ in range(NF - 1)]
    if NF > 1:
        insl = [ins]
        for f in range(1, NF):   # every slot streams its own images
            netf, _ = synth.make_batch(cfg, batch=B, first_image=(rank * NF + f) * B + 1000)
            df = {k: torch.from_numpy(v).to(dev) for k, v in netf.items()}
            insl.append((df["pred_offset"], df["pred_cls"], df["pred_mask_coef"], d["priors"], df["proto_out"]))
        streams = [torch.cuda.Stream(dev) for _ in range(NF)]
    else:
        insl, streams = [ins], [torch.cuda.current_stream(dev)]

    def step(i):
        f = i % NF
        with torch.cuda.stream(streams[f]):
            pps[f].run(*insl[f], out=outs[f])

    for i in range(args.warmup):
        step(i)
    barrier()
    kern_ms = {}
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
        # reading the events of step i while step i+1.. are not yet queued would serialise the host;
        # instead each step's events are read back after the loop for the LAST step only, and the
        # per-kernel averages come from the second, instrumented region below.
    barrier()
    t1 = time.perf_counter()
    elapsed = t1 - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # Per-kernel durations: same launches, events read back after every step (host-serialised, so
    # this region is NOT the one `value` comes from; the kernels themselves are identical).
    pp.profile(True)
    acc = {}
    nprof = args.steps
    barrier()
    t2 = time.perf_counter()
    for _ in range(nprof):
        pp.run(*ins, out=out)
        for k, v in pp.kernel_times_ms().items():
            acc[k] = acc.get(k, 0.0) + v
    barrier()
    ms_per_step_events = (time.perf_counter() - t2) / nprof * 1e3
    kern_ms = {k: v / nprof for k, v in acc.items()}

    S = pp.last_survivors(B)
    N = out["num_detections"].cpu().numpy()
    rows = np.full_like(N, max_dets) if zero_pad else N     # mask rows actually written (tf.zeros rows included)
    total_bytes = algorithmic_bytes(cfg, A, C, P, K, S, rows)
    kb = kernel_bytes(A, C, P, K, S, rows, args.nms_mode)
    dominant = max(kern_ms, key=lambda k: kern_ms[k])
    ach = kb[dominant] / (kern_ms[dominant] * 1e-3) / 1e9
    ms_per_step = elapsed / args.steps * 1e3
    value = B * world * args.steps / elapsed

    result = {
        "metric": "post-process images/sec (decode+FastNMS+masks) 550x550, bs32; % HBM roofline",
        "value": round(value, 1),
        "unit": "images/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 4),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {
            "workload": f"BASELINE config {args.config}: {cfg.img_h}x{cfg.img_w}, {A} anchors, {C} classes, "
                        f"proto {ph}x{pw}x32, batch {B} per GPU, conf {cfg.conf_thresh}, top_k {top_k}, "
                        f"max {max_dets} dets, nms_mode {args.nms_mode}" + (", 300 zero-padded rows" if zero_pad else ""),
            "batch_per_gpu": B, "nms_mode": args.nms_mode,
            "mean_survivors": float(np.mean(S)), "mean_dets": float(np.mean(N)),
        },
        "roofline": {
            "bound": "hbm", "kernel": dominant,
            "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": None,
            "bytes_per_launch": kb[dominant], "avg_launch_ms": round(kern_ms[dominant], 5),
        },
        "kernels_ms": {k: round(v, 5) for k, v in kern_ms.items()},
        "ms_per_step_with_events": round(ms_per_step_events, 4),
        "kernels_gbs": {k: round(kb[k] / (kern_ms[k] * 1e-3) / 1e9, 1) if kern_ms[k] > 0 else 0.0 for k in kern_ms},
        "pipeline": {
            "bytes_per_step": total_bytes,
            "achieved_gbs": round(total_bytes / (ms_per_step * 1e-3) / 1e9, 1),
            "frac_of_hbm_peak": round(total_bytes / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
        },
    }

    # HBM bytes per launch from the PMC passes of the same command (tools/gpu_profile.sh), if committed
    try:
        tj = json.load(open(os.path.join(ROOT, "profiles", "hbm_traffic.json")))
        key = f"{args.nms_mode}:{dominant}"
        if args.config == 2 and B == cfg.batch and key in tj:
            result["roofline"]["traffic"] = tj[key]
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
    """The oracle's C port (oracle/detect_oracle.c) timed on this host: the checker, timed as a
    baseline - never the thing shipped.  The reference itself (TF2) cannot run here."""
    from oracle import c_oracle
    c_oracle.lib()
    threads = max(1, min(args.cpu_threads, os.cpu_count() or 1))
    B = net["pred_cls"].shape[0]
    kw = dict(conf_thresh=cfg.conf_thresh, nms_thresh=cfg.nms_thresh, top_k=top_k, nms_mode=args.nms_mode,
              max_dets=max_dets)
    n1 = min(4, B)
    sub = {k: (v[:n1] if k != "priors" else v) for k, v in net.items()}
    t0 = time.perf_counter()
    c_oracle.detect(sub, shp, nthreads=1, **kw)
    t_single = time.perf_counter() - t0
    reps, t_multi = 0, 0.0
    t0 = time.perf_counter()
    while True:
        c_oracle.detect(net, shp, nthreads=threads, **kw)
        reps += 1
        t_multi = time.perf_counter() - t0
        if t_multi > 8.0 or reps >= 5:
            break
    return {
        "value": round(B * reps / t_multi, 2), "unit": "images/s", "cores": threads, "kind": "port",
        "sample": f"{reps} x the same {B}-image batch, OpenMP over images, {threads} threads; "
                  f"single-thread: {n1} images in {t_single:.2f}s",
        "single_thread_value": round(n1 / t_single, 2),
    }


if __name__ == "__main__":
    main()
