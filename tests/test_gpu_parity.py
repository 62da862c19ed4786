"""Parity of the HIP path (through the C ABI) against the oracle, on a real MI355X.

Bars (BASELINE.json north_star): kept anchor indices, classes and NMS ordering bit-exact; boxes,
scores and masks within 1e-5 abs.  Against the C oracle (same YP-MATH arithmetic contract) the test
is stricter: every output float must be bit-identical.  Against the independent numpy oracle
(libm exp, numpy reductions, BLAS matmul) the 1e-5 tolerance applies and indices must be exact on
the committed golden fixtures."""
import glob
import os

import numpy as np
import pytest

from tests.util import compare, to_numpy

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
TOL = 1e-5   # north_star tolerance for boxes / scores / masks


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return torch


def run_gpu(torch, net, shp, *, mode, top_k, max_dets, conf=0.05, nms=0.5, use_crop=True, zero_pad=False,
            mask_threshold=-1.0, max_batch=None):
    from yolact_amd import PostConfig, PostProcessor
    B, A, C = net["pred_cls"].shape
    ph, pw = net["proto_out"].shape[1:3]
    pp = PostProcessor(PostConfig(A, C, ph, pw, int(shp[1]), int(shp[2]), conf_thresh=conf, nms_thresh=nms,
                                  nms_mode=mode, top_k=top_k, max_dets=max_dets, use_cropped_mask=use_crop,
                                  zero_pad_masks=zero_pad, mask_threshold=mask_threshold, max_batch=max_batch or B))
    dev = {k: torch.from_numpy(np.ascontiguousarray(v)).cuda() for k, v in net.items()}
    out = pp.run(dev["pred_offset"], dev["pred_cls"], dev["pred_mask_coef"], dev["priors"], dev["proto_out"])
    pp.sync()
    got = to_numpy(out)
    got["num_survivors"] = pp.last_survivors(B)
    pp.close()
    return got


def check_vs_c_oracle(torch, net, shp, *, mode, top_k, max_dets, conf=0.05, nms=0.5, use_crop=True, nthreads=8, label=""):
    from oracle import c_oracle as co
    got = run_gpu(torch, net, shp, mode=mode, top_k=top_k, max_dets=max_dets, conf=conf, nms=nms, use_crop=use_crop)
    ref = co.detect(net, shp, conf_thresh=conf, nms_thresh=nms, top_k=top_k, nms_mode=mode, max_dets=max_dets,
                    use_cropped_mask=use_crop, nthreads=nthreads)
    assert np.array_equal(got["num_survivors"], ref["num_survivors"]), label
    compare(got, ref, exact_values=True, label=label)
    return got, ref


MODES = [("cc", 15, 15), ("cc", 200, 100), ("per_class", 200, 100), ("per_class", 15, 7), ("cc", 256, 300)]


@pytest.mark.parametrize("seed", [0, 1, 2])
@pytest.mark.parametrize("mode,top_k,max_dets", MODES)
def test_toy_bit_exact_vs_c_oracle(torch_cuda, seed, mode, top_k, max_dets):
    from yolact_amd import synth
    if mode == "per_class" and max_dets > 256:
        pytest.skip("per-class rows <= 256")
    net, shp = synth.make_toy(seed)            # A = 240: ragged last tile, S < top_k for some settings
    check_vs_c_oracle(torch_cuda, net, shp, mode=mode, top_k=top_k, max_dets=max_dets, label=f"toy{seed}")


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLDEN, "toy_*.npz"))))
def test_golden_fixtures_vs_numpy_oracle_outputs(torch_cuda, path):
    """Committed vectors made by the independent numpy oracle: indices exact, values within 1e-5."""
    z = np.load(path)
    net = {k[3:]: z[k] for k in z.files if k.startswith("in_")}
    got = run_gpu(torch_cuda, net, z["img_shape"], mode=str(z["nms_mode"]), top_k=int(z["top_k"]),
                  max_dets=int(z["max_dets"]), conf=float(z["conf"]))
    ref = {k[4:]: z[k] for k in z.files if k.startswith("out_")}
    assert np.array_equal(got["num_survivors"], ref["num_survivors"])
    compare(got, ref, exact_values=False, atol=TOL, label=os.path.basename(path))


def test_golden_config2_image0(torch_cuda):
    from yolact_amd import synth
    z = np.load(os.path.join(GOLDEN, "config2_image0.npz"))
    cfg = synth.CONFIGS[2]
    net, shp = synth.make_batch(cfg, batch=1)
    for mode, top_k, max_dets in (("cc", 15, 15), ("cc", 200, 100), ("per_class", 200, 100)):
        tag = f"{mode}_{top_k}"
        got = run_gpu(torch_cuda, net, shp, mode=mode, top_k=top_k, max_dets=max_dets, conf=cfg.conf_thresh)
        for k in ("num_detections", "num_survivors", "detection_anchor_indices", "detection_classes"):
            assert np.array_equal(got[k], z[f"{tag}_{k}"]), (tag, k)
        for k in ("detection_scores", "detection_boxes"):
            assert np.abs(got[k] - z[f"{tag}_{k}"]).max() <= TOL, (tag, k)
        m = got["detection_masks"][0].reshape(max_dets, -1)
        assert np.abs(m.sum(axis=1, dtype=np.float64) - z[f"{tag}_mask_sums"]).max() < 1e-2
        assert np.array_equal((m > 0).sum(axis=1), z[f"{tag}_mask_nonzero"])


@pytest.mark.parametrize("mode,top_k,max_dets", [("cc", 15, 15), ("cc", 200, 100), ("per_class", 200, 100)])
def test_config2_batch8_bit_exact(torch_cuda, mode, top_k, max_dets):
    from yolact_amd import synth
    cfg = synth.CONFIGS[2]
    net, shp = synth.make_batch(cfg, batch=8)
    check_vs_c_oracle(torch_cuda, net, shp, mode=mode, top_k=top_k, max_dets=max_dets, conf=cfg.conf_thresh, label="cfg2")


def test_config1_random_init_has_no_survivors(torch_cuda):
    """BASELINE config 1 with random-init-like logits: max fg prob ~ 1/81 < conf -> empty image (detection.py:81)."""
    from yolact_amd import synth
    cfg = synth.CONFIGS[1]
    net, shp = synth.make_batch(cfg, batch=1, mode="random_init")
    got, _ = check_vs_c_oracle(torch_cuda, net, shp, mode="cc", top_k=15, max_dets=15, label="cfg1-empty")
    assert got["num_detections"][0] == 0 and got["num_survivors"][0] == 0
    assert not got["detection_scores"].any() and (got["detection_anchor_indices"] == -1).all()
    net, shp = synth.make_batch(cfg, batch=1, mode="planted")
    check_vs_c_oracle(torch_cuda, net, shp, mode="cc", top_k=15, max_dets=15, label="cfg1-planted")


@pytest.mark.parametrize("nb", [1, 2, 4])
@pytest.mark.parametrize("mode,top_k,max_dets", [("cc", 15, 15), ("per_class", 200, 100)])
def test_small_batches_in_the_shape_of_the_reference_call(torch_cuda, nb, mode, top_k, max_dets):
    """infer.py:5-11 calls the model on one image: BASELINE config 1 at B = 1, 2, 4 in both modes.  In cross-class mode
    these batches take the selection from registers (every candidate list fits the first batch of loads)."""
    from yolact_amd import synth
    cfg = synth.CONFIGS[1]
    net, shp = synth.make_batch(cfg, batch=nb, mode="planted")
    got, ref = check_vs_c_oracle(torch_cuda, net, shp, mode=mode, top_k=top_k, max_dets=max_dets, conf=cfg.conf_thresh,
                                 label=f"cfg1-B{nb}-{mode}")
    assert (ref["num_detections"] > 0).all()


@pytest.mark.parametrize("top_k,max_dets", [(15, 15), (200, 100)])
def test_cross_class_selection_when_the_scores_share_one_bin(torch_cuda, top_k, max_dets):
    """600 anchors with identical logits (equal scores; tf.math.top_k takes the lower index first) beside a few
    stronger ones: the fixed score bins of the selection from registers cannot isolate <= 256 keys around the k-th,
    so the general selection has to take over - with the same result."""
    from yolact_amd import synth
    cfg = synth.CONFIGS[2]
    net, shp = synth.make_batch(cfg, batch=2)
    cls = net["pred_cls"]
    cls[:] = 0.0
    cls[..., 0] = 8.0
    same = np.arange(7, cls.shape[1], 32)[:600]
    cls[:, same, 5] = 9.0
    cls[1, same[::100], 9] = 11.0          # image 1: six candidates above the plateau
    got, ref = check_vs_c_oracle(torch_cuda, net, shp, mode="cc", top_k=top_k, max_dets=max_dets, label="plateau")
    assert (ref["num_survivors"] >= 600).all()


@pytest.mark.parametrize("mode", ["per_class", "cc"])
def test_config4_nms_stress_bit_exact(torch_cuda, mode):
    """conf 0.01, every anchor planted: saturated top-200 in all 80 classes; in cc mode the image holds
    19248 candidates, which exercises the un-staged radix select."""
    from yolact_amd import synth
    cfg = synth.CONFIGS[4]
    net, shp = synth.make_batch(cfg, batch=2)
    got, ref = check_vs_c_oracle(torch_cuda, net, shp, mode=mode, top_k=200, max_dets=100, conf=cfg.conf_thresh, label="cfg4")
    assert (ref["num_survivors"] > 19000).all()


@pytest.mark.parametrize("cfg_id", [3, 5])
def test_non_square_configs_bit_exact(torch_cuda, cfg_id):
    from yolact_amd import synth
    cfg = synth.CONFIGS[cfg_id]
    net, shp = synth.make_batch(cfg, batch=2)
    check_vs_c_oracle(torch_cuda, net, shp, mode="per_class", top_k=200, max_dets=100, label=f"cfg{cfg_id}")
    check_vs_c_oracle(torch_cuda, net, shp, mode="cc", top_k=15, max_dets=15, label=f"cfg{cfg_id}")


@pytest.mark.parametrize("num_classes", [2, 4, 17, 33, 100, 130, 256])
def test_class_counts_even_and_odd(torch_cuda, num_classes):
    """Even C takes the padded-stride LDS path; C > 96 the wider per-lane class loops."""
    from yolact_amd import synth
    net, shp = synth.make_toy(9, num_classes=num_classes)
    check_vs_c_oracle(torch_cuda, net, shp, mode="cc", top_k=30, max_dets=30, label=f"C{num_classes}")
    check_vs_c_oracle(torch_cuda, net, shp, mode="per_class", top_k=30, max_dets=30, label=f"C{num_classes}")


def test_thresholds_and_options(torch_cuda):
    from yolact_amd import synth
    from oracle import c_oracle as co
    net, shp = synth.make_toy(4)
    check_vs_c_oracle(torch_cuda, net, shp, mode="cc", top_k=100, max_dets=100, conf=0.6, nms=0.3, label="conf.6")
    check_vs_c_oracle(torch_cuda, net, shp, mode="cc", top_k=100, max_dets=100, conf=0.999999, label="S=0")
    check_vs_c_oracle(torch_cuda, net, shp, mode="per_class", top_k=100, max_dets=5, conf=0.01, nms=0.9, label="truncate")
    check_vs_c_oracle(torch_cuda, net, shp, mode="cc", top_k=50, max_dets=50, use_crop=False, label="nocrop")
    # zero padding of unused mask rows + thresholded bitmaps
    got = run_gpu(torch_cuda, net, shp, mode="cc", top_k=15, max_dets=40, zero_pad=True, mask_threshold=0.5)
    ref = co.detect(net, shp, top_k=15, nms_mode="cc", max_dets=40)
    compare(got, ref, exact_values=True, label="zero_pad")
    for b in range(2):
        n = int(ref["num_detections"][b])
        assert not got["detection_masks"][b, n:].any()
        assert np.array_equal(got["detection_bitmaps"][b, :n], (ref["detection_masks"][b, :n] > 0.5).astype(np.uint8))
        assert not got["detection_bitmaps"][b, n:].any()


def test_handle_reuse_and_smaller_batches(torch_cuda):
    """One handle, several runs with different batch sizes: scratch counters are left clean between runs."""
    from yolact_amd import PostConfig, PostProcessor, synth
    from oracle import c_oracle as co
    torch = torch_cuda
    net, shp = synth.make_toy(6, batch=5)
    B, A, C = net["pred_cls"].shape
    for mode in ("cc", "per_class"):
        pp = PostProcessor(PostConfig(A, C, 12, 12, 64, 64, nms_mode=mode, top_k=40, max_dets=20, max_batch=5))
        for nb in (5, 2, 5, 1):
            sub = {k: (v if k == "priors" else v[:nb]) for k, v in net.items()}
            dev = {k: torch.from_numpy(np.ascontiguousarray(v)).cuda() for k, v in sub.items()}
            out = pp.run(dev["pred_offset"], dev["pred_cls"], dev["pred_mask_coef"], dev["priors"], dev["proto_out"])
            pp.sync()
            ref = co.detect(sub, shp, top_k=40, nms_mode=mode, max_dets=20)
            compare(to_numpy(out), ref, exact_values=True, label=f"reuse-{mode}-{nb}")
        with pytest.raises(Exception, match="max_batch"):
            big = {k: torch.from_numpy(np.ascontiguousarray(np.concatenate([v, v]) if k != "priors" else v)).cuda() for k, v in net.items()}
            pp.run(big["pred_offset"], big["pred_cls"], big["pred_mask_coef"], big["priors"], big["proto_out"])
        pp.close()


def test_detect_wrapper_reference_layout(torch_cuda):
    """Drop-in class, reference defaults: cross-class, top_k 15, 300 zero-padded rows (detection.py:21,65-69)."""
    from yolact_amd import Detect, synth
    from oracle import detect_oracle as po
    net, shp = synth.make_toy(8)
    det = Detect(5, max_output_size=17, per_class_max_output_size=3, conf_thresh=0.05, nms_thresh=0.5)
    res = to_numpy(det(net, shp))
    ref = po.detect(net, shp, conf_thresh=0.05, top_k=15, nms_mode="cc", max_output_size=300)
    assert res["detection_masks"].shape == (2, 300, 12, 12) and res["detection_boxes"].shape == (2, 300, 4)
    assert res["num_detections"].dtype == np.int32 and res["detection_classes"].dtype == np.float32
    compare(res, ref, exact_values=False, atol=TOL, label="Detect")
    for b in range(2):
        assert not res["detection_masks"][b, int(ref["num_detections"][b]):].any()      # tf.zeros rows
    # unlocked knobs
    det2 = Detect(5, 100, 100, 0.05, 0.4, strict_reference=False, nms_mode="per_class", top_k=200, max_dets=100)
    res2 = to_numpy(det2(net, shp))
    ref2 = po.detect(net, shp, conf_thresh=0.05, nms_thresh=0.4, top_k=200, nms_mode="per_class", max_output_size=100, max_dets=100)
    compare(res2, ref2, exact_values=False, atol=TOL, label="Detect-per-class")


@pytest.mark.parametrize("mode,top_k,max_dets,zero_pad,thr", [("per_class", 200, 100, False, -1.0), ("cc", 15, 300, True, -1.0),
                                                              ("cc", 200, 100, False, 0.5), ("per_class", 200, 40, True, 0.5)])
def test_tracked_output_set_equals_fresh_runs(torch_cuda, mode, top_k, max_dets, zero_pad, thr):
    """yp_run_kept: one tracked output set through five runs on different inputs - many detections, none at all, the
    first batch again, larger boxes, other images - equals yp_run into fresh buffers full of garbage every time, bit
    for bit, masks and bitmaps included: what the run before wrote outside this run's boxes has been cleared, nothing
    else was touched."""
    from yolact_amd import PostConfig, PostProcessor, synth
    torch = torch_cuda
    cfg = synth.CONFIGS[2]
    B = 3
    pp = None
    kept = None
    seq = [(0, "planted", 1.0), (4, "random_init", 1.0), (0, "planted", 1.0), (8, "planted", 3.0), (12, "planted_soft", 1.0)]
    for first, gen, scale in seq:
        net, shp = synth.make_batch(cfg, batch=B, first_image=first, mode=gen)
        if scale != 1.0:
            net["pred_offset"][..., 2:] += np.float32(np.log(scale) / 0.2)      # utils/utils.py:202: w = pw * exp(dw * 0.2)
        A, C = net["pred_cls"].shape[1:]
        if pp is None:
            pp = PostProcessor(PostConfig(A, C, 138, 138, cfg.img_h, cfg.img_w, conf_thresh=cfg.conf_thresh, nms_mode=mode,
                                          top_k=top_k, max_dets=max_dets, zero_pad_masks=zero_pad, mask_threshold=thr, max_batch=B))
            kept = pp.alloc_kept(B, bitmaps=thr >= 0)
            for v in kept.values():
                v.fill_(7)                                   # the first run must not rely on the buffers' contents
        dev = {k: torch.from_numpy(v).cuda() for k, v in net.items()}
        args = (dev["pred_offset"], dev["pred_cls"], dev["pred_mask_coef"], dev["priors"], dev["proto_out"])
        fresh = pp.alloc_outputs(B, bitmaps=thr >= 0)
        for v in fresh.values():
            v.fill_(9)
        pp.run(*args, out=fresh)
        got = pp.run(*args, out=kept)
        assert got is kept
        pp.sync()
        n = fresh["num_detections"]
        rows = max_dets if zero_pad else None
        for k in fresh:
            a, b = fresh[k], kept[k]
            if k in ("detection_masks", "detection_bitmaps") and rows is None:
                # rows behind a detection-capable row are not part of the contract without zero padding: compare
                # the rows the call defines (every row a detection can land in)
                lim = min(max_dets, 256) if mode == "per_class" else min(max_dets, top_k, 256)
                a, b = a[:, :lim], b[:, :lim]
            assert torch.equal(a, b), (first, gen, k)
        assert bool((n == 0).all()) == (gen == "random_init")
    kept.close()
    pp.close()


def test_tracked_output_set_refuses_other_buffers_and_batches(torch_cuda):
    from yolact_amd import PostConfig, PostProcessor, synth, _lib
    torch = torch_cuda
    cfg = synth.CONFIGS[2]
    net, shp = synth.make_batch(cfg, batch=2)
    A, C = net["pred_cls"].shape[1:]
    pp = PostProcessor(PostConfig(A, C, 138, 138, cfg.img_h, cfg.img_w, conf_thresh=cfg.conf_thresh, nms_mode="cc", top_k=15,
                                  max_dets=15, max_batch=2))
    dev = {k: torch.from_numpy(v).cuda() for k, v in net.items()}
    args = (dev["pred_offset"], dev["pred_cls"], dev["pred_mask_coef"], dev["priors"], dev["proto_out"])
    kept = pp.alloc_kept(2)
    pp.run(*args, out=kept)
    own = kept["detection_masks"]
    kept["detection_masks"] = torch.empty_like(own)                              # another buffer under the same record
    with pytest.raises(_lib.YolactPostError, match="first call"):
        pp.run(*args, out=kept)
    kept["detection_masks"] = own
    one = tuple(t if t.dim() == 2 and t.shape[0] == A else t[:1] for t in args)
    small = type(kept)({k: v[:1] for k, v in kept.items()})
    small._bind(kept._L, kept._k)
    with pytest.raises(_lib.YolactPostError, match="bound to batch 2"):
        pp.run(*one, out=small)
    small._k = None
    # the caller wrote into the masks: reset -> the next run refills
    pp.sync()
    kept["detection_masks"].fill_(5)
    kept.reset()
    pp.run(*args, out=kept)
    pp.sync()
    ref = pp.run(*args)
    pp.sync()
    assert torch.equal(kept["detection_masks"], ref["detection_masks"])
    kept.close()
    pp.close()


def test_detect_with_kept_outputs_matches_fresh_outputs(torch_cuda):
    """Detect(reuse_outputs=True): one output set per shape, zeroed once; three calls in a row on different inputs
    (the second with fewer detections than the first) return what a fresh, fully zero-padded call returns - the rows
    behind the written ones are still zero - and the same tensors every time."""
    from yolact_amd import Detect, synth
    torch = torch_cuda
    cfg = synth.CONFIGS[2]
    fresh = Detect(cfg.num_classes, 300, 100, cfg.conf_thresh, 0.5)
    kept = Detect(cfg.num_classes, 300, 100, cfg.conf_thresh, 0.5, reuse_outputs=True)
    ptrs = None
    for first, mode in ((0, "planted"), (4, "random_init"), (8, "planted")):
        net, shp = synth.make_batch(cfg, batch=2, first_image=first, mode=mode)
        dev = {k: torch.from_numpy(v).cuda() for k, v in net.items()}
        a = fresh(dev, shp)
        b = kept(dev, shp)
        torch.cuda.synchronize()
        assert a["detection_masks"].shape == (2, 300, 138, 138)
        for k in a:
            assert torch.equal(a[k], b[k]), (first, k)
        assert (b["num_detections"] == 0).all() == (mode == "random_init")
        now = {k: v.data_ptr() for k, v in b.items()}
        assert ptrs is None or now == ptrs
        ptrs = now


def test_full_size_batch32_properties_and_oracle(torch_cuda):
    """BASELINE config 2 at its full size (batch 32): bit-exact against the C oracle, plus the
    size-independent properties of the path: idempotence, image-permutation equivariance,
    descending scores, masks in [0,1] and zero outside the padded crop box."""
    from yolact_amd import synth
    cfg = synth.CONFIGS[2]
    net, shp = synth.make_batch(cfg, batch=32)
    got, ref = check_vs_c_oracle(torch_cuda, net, shp, mode="per_class", top_k=200, max_dets=100, conf=cfg.conf_thresh, label="cfg2-b32")
    again = run_gpu(torch_cuda, net, shp, mode="per_class", top_k=200, max_dets=100)
    for k in ("detection_boxes", "detection_scores", "detection_classes", "detection_anchor_indices", "detection_masks"):
        assert np.array_equal(got[k], again[k]), "not idempotent: " + k
    perm = np.random.default_rng(0).permutation(32)
    pnet = {k: (v if k == "priors" else v[perm]) for k, v in net.items()}
    pgot = run_gpu(torch_cuda, pnet, shp, mode="per_class", top_k=200, max_dets=100)
    for k in ("detection_boxes", "detection_scores", "detection_classes", "detection_anchor_indices", "num_detections"):
        assert np.array_equal(pgot[k], got[k][perm]), "not permutation equivariant: " + k
    for b in range(32):
        n = int(got["num_detections"][b])
        s = got["detection_scores"][b, :n]
        assert n <= 100 and np.all(s[:-1] >= s[1:]) and np.all(s > cfg.conf_thresh)
        m = got["detection_masks"][b, :n]
        assert m.min() >= 0.0 and m.max() <= 1.0
        bx = got["detection_boxes"][b, :n]
        y1 = np.clip(np.minimum(bx[:, 0], bx[:, 2]) * 138 - 1, 0, 138); y2 = np.clip(np.maximum(bx[:, 0], bx[:, 2]) * 138 + 1, 0, 138)
        for i in range(0, n, 17):
            rows_out = np.r_[0:int(np.floor(y1[i]))]
            assert not m[i, rows_out].any()
            assert not m[i, int(np.ceil(y2[i])) + 1:].any()


def test_per_class_large_groups_and_classes(torch_cuda):
    """Two foreground classes with ~1000 candidates each: the class-group workgroups leave their
    thread-per-candidate path (> 1024 candidates) for the class-by-class one, whose select has more
    than 256 keys to choose from."""
    from yolact_amd import synth
    net, shp = synth.make_toy(21, batch=2, fmaps=((32, 32), (8, 8)), num_classes=3, p_fg=0.9)   # A = 3264
    got, ref = check_vs_c_oracle(torch_cuda, net, shp, mode="per_class", top_k=200, max_dets=100, label="large-groups")
    assert (ref["num_survivors"] > 2200).all()
    check_vs_c_oracle(torch_cuda, net, shp, mode="cc", top_k=256, max_dets=300, label="large-groups-cc")


def test_per_class_groups_beyond_the_lds_key_array(torch_cuda):
    """~3400 candidates in each of two classes: more than the 1536 keys a class group keeps in LDS, so its
    class-by-class path reads the image's lists again for every class; its pooled survivors (> max_dets)
    are cut to the group's best max_dets before the image's selection."""
    from yolact_amd import synth
    net, shp = synth.make_toy(24, batch=2, fmaps=((48, 48), (8, 8)), num_classes=3, p_fg=0.97)   # A = 7104
    got, ref = check_vs_c_oracle(torch_cuda, net, shp, mode="per_class", top_k=200, max_dets=100, label="huge-groups")
    assert (ref["num_survivors"] > 6000).all()


@pytest.mark.parametrize("max_dets", [5, 20, 100])
def test_per_class_group_pool_cut_to_max_dets(torch_cuda, max_dets):
    """A class group of the thread-per-candidate path (a few hundred candidates) whose NMS survivors
    outnumber max_dets: the group hands only its best max_dets to the image's selection, which must not
    change the result (top max_dets of the union = top max_dets of the groups' top max_dets)."""
    from yolact_amd import synth
    net, shp = synth.make_toy(25, batch=3, fmaps=((16, 16), (4, 4)), num_classes=4, p_fg=0.8)   # A = 816
    got, ref = check_vs_c_oracle(torch_cuda, net, shp, mode="per_class", top_k=200, max_dets=max_dets, label=f"pool-cut-{max_dets}")
    assert (ref["num_detections"] == max_dets).all()


@pytest.mark.parametrize("num_classes,conf", [(81, 0.05), (81, 0.3), (7, 0.05), (200, 0.01), (3, 0.5)])
def test_early_rejects_never_drop_an_anchor_at_the_threshold(torch_cuda, num_classes, conf):
    """score_filter's two exact-safe rejects (1/(1+exp(x_bg-xf)) and the Jensen bound on the other classes)
    against rows built to sit AT the confidence threshold: the background logit is solved so that the best
    foreground probability is conf * (1 +- 1e-6 ... 1e-2), with the other logits tightly bunched (where the
    Jensen bound is sharpest), widely spread, and shifted by large offsets (|x| up to ~120, beyond which the
    second reject switches itself off).  The survivor count and every output must equal the C oracle's."""
    from yolact_amd import synth
    rng = np.random.default_rng(40 + num_classes)
    net, shp = synth.make_toy(26, batch=2, fmaps=((16, 16), (4, 4)), num_classes=num_classes)     # A = 816
    cls = net["pred_cls"].astype(np.float64)
    B, A, C = cls.shape
    spread = rng.choice([1e-3, 0.05, 1.0, 4.0], (B, A, 1))
    x = rng.normal(0.0, 1.0, (B, A, C)) * spread
    x[..., 1] += rng.uniform(0.0, 6.0, (B, A))                                      # class 1 is the best foreground class
    f = x[..., 1:].max(-1)
    others = np.exp(x[..., 1:] - f[..., None]).sum(-1)                              # sum over fg of exp(x - f), >= 1
    rel = rng.choice([-1e-2, -1e-4, -1e-6, 1e-6, 1e-4, 1e-2], (B, A))
    target = np.minimum(conf * (1.0 + rel), 0.999 / others)                         # p_fgmax = 1 / (others + exp(x0 - f))
    x[..., 0] = f + np.log(np.maximum(1.0 / target - others, 1e-30))
    x += rng.choice([0.0, 0.0, 30.0, -60.0, 110.0, -125.0, 300.0], (B, A, 1))        # softmax is shift-invariant; the kernel's sums are not
    net["pred_cls"] = x.astype(np.float32)
    for mode, top_k, md in (("per_class", 100, 50), ("cc", 200, 100)):
        got, ref = check_vs_c_oracle(torch_cuda, net, shp, mode=mode, top_k=top_k, max_dets=md, conf=conf, label=f"edge-C{num_classes}")
    assert 0.2 * A < ref["num_survivors"].min() and ref["num_survivors"].max() < 0.8 * A      # both sides of the threshold are populated


def test_bounded_candidate_lists(torch_cuda):
    """yp_create_bounded: lists sized for 2 keys per anchor of an image use a fraction of the scratch (the
    per-group lists are the same size in both handles; the worst-case part is the image's overflow list) and give
    bit-identical results with no overflow on an ordinary batch; sized for 1 key per anchor on a batch in which
    every anchor emits several keys, the overflow flag is raised (and cleared by reading it), nothing faults,
    and the unbounded handle on the same batch - whose class groups then take their slow path over the
    overflow list - does not raise it."""
    import torch
    from yolact_amd import PostConfig, PostProcessor, synth
    cfg = synth.CONFIGS[2]
    net, shp = synth.make_batch(cfg, batch=4)
    A, C = net["pred_cls"].shape[1:]; ph, pw = net["proto_out"].shape[1:3]
    d = {k: torch.from_numpy(v).cuda() for k, v in net.items()}
    ins = (d["pred_offset"], d["pred_cls"], d["pred_mask_coef"], d["priors"], d["proto_out"])
    pc = PostConfig(A, C, ph, pw, cfg.img_h, cfg.img_w, conf_thresh=cfg.conf_thresh, nms_mode="per_class", top_k=200, max_dets=100, max_batch=4)
    full, small = PostProcessor(pc), PostProcessor(pc, keys_per_anchor=2)
    assert small.scratch_bytes < 0.4 * full.scratch_bytes
    a, b = full.run(*ins), small.run(*ins)
    assert not small.overflowed() and not full.overflowed()
    for k in a:
        if a[k] is not None:
            assert torch.equal(a[k], b[k]), k
    # every anchor with five classes at p ~ 0.19: five keys per anchor
    hot = net["pred_cls"].copy(); hot[...] = -4.0; hot[..., 1:6] = 4.0
    ins_hot = (ins[0], torch.from_numpy(hot).cuda()) + ins[2:]
    tiny = PostProcessor(pc, keys_per_anchor=1)
    tiny.run(*ins_hot); full.run(*ins_hot)
    assert tiny.overflowed() and not tiny.overflowed()
    assert not full.overflowed()
    assert (full.last_survivors(4) == A).all() and (tiny.last_survivors(4) == A).all()


def test_more_work_items_than_resident_waves(torch_cuda):
    """4000 small images: score_filter has more work items than resident waves and walks them
    grid-stride; every NMS workgroup finds one list per image."""
    from yolact_amd import synth
    net, shp = synth.make_toy(22, batch=4000)
    for mode, top_k, max_dets in (("cc", 15, 15), ("per_class", 30, 10)):
        check_vs_c_oracle(torch_cuda, net, shp, mode=mode, top_k=top_k, max_dets=max_dets, label=f"many-items-{mode}")


def test_odd_proto_size_takes_the_narrow_zero_stores(torch_cuda):
    """P = 13*11 is not a multiple of 4: mask rows are not 16-byte aligned and the zero pass stores dwords."""
    from yolact_amd import synth
    net, shp = synth.make_toy(23, proto_hw=(13, 11))
    check_vs_c_oracle(torch_cuda, net, shp, mode="cc", top_k=50, max_dets=40, label="oddP")
    check_vs_c_oracle(torch_cuda, net, shp, mode="per_class", top_k=50, max_dets=40, label="oddP")


def test_masks_larger_than_the_infinity_cache(torch_cuda):
    """Config 3 shape, 6 images x 300 rows: 354 MB of masks, the low-residency launch of mask_assemble."""
    from yolact_amd import synth
    cfg = synth.CONFIGS[3]
    net, shp = synth.make_batch(cfg, batch=6)
    check_vs_c_oracle(torch_cuda, net, shp, mode="cc", top_k=200, max_dets=300, label="big-masks")


@pytest.mark.parametrize("hw", [(550, 550), (768, 1024), (800, 1333), (64, 64)])
def test_device_priors_bit_exact(torch_cuda, hw):
    """yp_make_priors against the line-by-line restatement of data/anchor.py:23-68 (and the host numpy
    generator the synthetic inputs use): every float bit-identical, anchors in the reference's order."""
    from oracle import detect_oracle as po
    from yolact_amd import anchors
    H, W = hw
    if hw == (64, 64):
        fm, scales = ((8, 8), (4, 4), (1, 3)), (12, 24, 50.5)
    else:
        fm, scales = anchors.feature_map_sizes(H, W), anchors.DEFAULT_SCALES
    ratios = (1.0, 0.5, 2.0)
    got = anchors.make_priors_device(H, W, feature_maps=fm, aspect_ratios=ratios, scales=scales)
    torch_cuda.cuda.synchronize()
    got = got.cpu().numpy()
    ref = po.generate_anchors(H, W, fm, ratios, scales)
    assert got.shape == ref.shape and got.dtype == np.float32
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))
    host = anchors.make_priors(H, W, feature_maps=fm, aspect_ratios=ratios, scales=scales)
    assert np.array_equal(host.view(np.uint32), ref.view(np.uint32))
    assert anchors.make_priors_device(H, W, feature_maps=fm, aspect_ratios=ratios, scales=scales).data_ptr() == \
        anchors.make_priors_device(H, W, feature_maps=fm, aspect_ratios=ratios, scales=scales).data_ptr()   # cached


def test_randomized_small_shapes(torch_cuda):
    """Forty random small problems (anchor counts that are not multiples of 4 or 32, 2..40 classes, odd
    proto sizes, thresholds from permissive to strict, both modes): every output bit-identical."""
    from yolact_amd import synth
    rng = np.random.default_rng(2024)
    for case in range(40):
        fh, fw = int(rng.integers(1, 9)), int(rng.integers(1, 9))
        fmaps = ((fh, fw),) if rng.random() < 0.5 else ((fh, fw), (max(1, fh // 2), max(1, fw // 2)))
        C = int(rng.integers(2, 41))
        proto_hw = (int(rng.integers(3, 20)), int(rng.integers(3, 20)))
        batch = int(rng.integers(1, 5))
        mode = "cc" if rng.random() < 0.5 else "per_class"
        top_k = int(rng.integers(1, 60))
        max_dets = int(rng.integers(1, 50))
        conf = float(rng.choice([0.0, 0.01, 0.05, 0.3, 0.9]))
        nms = float(rng.choice([0.05, 0.3, 0.5, 0.95]))
        net, shp = synth.make_toy(1000 + case, batch=batch, fmaps=fmaps, num_classes=C, proto_hw=proto_hw,
                                  p_fg=float(rng.choice([0.0, 0.1, 0.6])), scales=(12, 24)[:len(fmaps)])
        label = f"rand{case}: A={net['pred_cls'].shape[1]} C={C} P={proto_hw} B={batch} {mode} k={top_k} md={max_dets} conf={conf} nms={nms}"
        check_vs_c_oracle(torch_cuda, net, shp, mode=mode, top_k=top_k, max_dets=max_dets, conf=conf, nms=nms, label=label)


def test_kept_index_agreement_vs_independent_arithmetic(torch_cuda):
    """north_star target: >= 95 % kept-index agreement with the reference arithmetic on unfiltered seeds.
    The stand-in for the reference's TF CPU arithmetic is the numpy oracle (libm exp, numpy
    reductions, BLAS matmul) - independent of the YP-MATH contract the kernels share with the C
    oracle.  Per output row and per kept set; rows of images that agree are within 1e-5."""
    from oracle import detect_oracle as po
    from tests.util import kept_index_agreement
    from yolact_amd import synth
    cfg = synth.CONFIGS[2]
    net, shp = synth.make_batch(cfg, batch=32, first_image=200)
    for mode, top_k, md in (("per_class", 200, 100), ("cc", 15, 15)):
        got = run_gpu(torch_cuda, net, shp, mode=mode, top_k=top_k, max_dets=md, conf=cfg.conf_thresh)
        ref = po.detect(net, shp, conf_thresh=cfg.conf_thresh, top_k=top_k, nms_mode=mode, max_output_size=md, max_dets=md)
        rate = kept_index_agreement(got, ref)
        print(f"kept-index agreement {mode}: {rate}")
        assert rate["per_detection"] >= 0.95 and rate["kept_set"] >= 0.95, (mode, rate)
        same = [b for b in range(32) if np.array_equal(got["detection_anchor_indices"][b], ref["detection_anchor_indices"][b])]
        assert len(same) > 0
        assert np.abs(got["detection_scores"][same] - ref["detection_scores"][same]).max() <= TOL
        assert np.abs(got["detection_boxes"][same] - ref["detection_boxes"][same]).max() <= TOL


def test_exact_order_vs_independent_arithmetic_on_unsaturated_scores(torch_cuda):
    """Deterministic setting: scores spread over 0.2-0.99, no ulp-level near-ties.  Kept
    indices, classes and NMS ordering must then be identical to the independent numpy/libm oracle on
    every image, boxes / scores / masks within 1e-5 (the north_star bar, full batch of 32)."""
    from oracle import detect_oracle as po
    from tests.util import kept_index_agreement
    from yolact_amd import synth
    cfg = synth.CONFIGS[2]
    net, shp = synth.make_batch(cfg, batch=32, first_image=300, mode="planted_soft")
    for mode, top_k, md in (("per_class", 200, 100), ("cc", 15, 15)):
        got = run_gpu(torch_cuda, net, shp, mode=mode, top_k=top_k, max_dets=md, conf=cfg.conf_thresh)
        ref = po.detect(net, shp, conf_thresh=cfg.conf_thresh, top_k=top_k, nms_mode=mode, max_output_size=md, max_dets=md)
        assert kept_index_agreement(got, ref)["exact_order"] == 1.0, mode
        compare(got, ref, exact_values=False, atol=TOL, label=f"soft-{mode}")


def test_sharded_detect_matches_unsharded(torch_cuda):
    """SURVEY.md 8e on the GPU path: the batch split into contiguous chunks, one Detect (handle, scratch)
    per shard, results concatenated on the host in image order.  Two shards on the one device of the
    test box stand in for two GPUs; no collective is involved."""
    from yolact_amd import Detect, synth
    from yolact_amd.sharding import ShardedDetect
    net, shp = synth.make_toy(31, batch=5)
    kw = dict(strict_reference=False, nms_mode="per_class", top_k=50, max_dets=20)
    one = to_numpy(Detect(5, 100, 100, 0.05, 0.5, **kw)(net, shp))
    sd = ShardedDetect(5, 100, 100, 0.05, 0.5, devices=[0, 0], **kw)
    two = sd(net, shp)
    assert all(v.is_pinned() and v.shape[0] == 5 for v in two.values())          # whole-batch pinned buffers, no concatenation
    assert [t["images"] for t in sd.last_timing] == [3, 2] and all(t["compute_ms"] > 0 and t["gather_ms"] >= 0 for t in sd.last_timing)
    again = sd(net, shp)
    assert all(again[k].data_ptr() == two[k].data_ptr() for k in two)           # the buffers are reused call after call
    two = {k: (v.numpy() if hasattr(v, "numpy") else v) for k, v in two.items()}
    for k in ("detection_boxes", "detection_classes", "detection_scores", "detection_masks", "num_detections",
              "detection_anchor_indices"):
        assert np.array_equal(one[k], two[k]), k
    assert two["num_detections"].shape == (5,)


# ---- BASELINE.json configs 3, 4, 5 at the per-GPU batch sizes they state -------------------------
_BATCH_CACHE = {}


def _full_batch(cfg_id, nb):
    """The first nb images of config cfg_id's batch (generated once per module run, sliced after)."""
    from yolact_amd import synth
    cfg = synth.CONFIGS[cfg_id]
    if cfg_id not in _BATCH_CACHE:
        _BATCH_CACHE.clear()                      # one config's batch of 64 at a time (2-3 GB of host memory)
        _BATCH_CACHE[cfg_id] = synth.make_batch(cfg, batch=cfg.batch)
    net, shp = _BATCH_CACHE[cfg_id]
    return cfg, {k: (v if k == "priors" else v[:nb]) for k, v in net.items()}, shp


@pytest.mark.parametrize("cfg_id,nb", [(3, 64), (3, 32), (3, 16), (3, 8), (5, 64), (5, 8), (4, 32)])
def test_configs_at_their_stated_shard_sizes(torch_cuda, cfg_id, nb):
    """Config 3 (768x1024, batch 64) whole and as the per-GPU shards of 2/4/8 GPUs (32/16/8 images),
    config 5 (800x1333, batch 64) whole and as an 8-GPU shard, config 4 (NMS stress) at its batch of 32:
    the batch size changes score_filter's work-item plan, mask_assemble's XCD placement and the
    zero-fill's grid, so each size is checked bit for bit against the C restatement."""
    cfg, net, shp = _full_batch(cfg_id, nb)
    check_vs_c_oracle(torch_cuda, net, shp, mode="per_class", top_k=cfg.top_k, max_dets=cfg.max_dets, conf=cfg.conf_thresh,
                      nthreads=16, label=f"cfg{cfg_id}-b{nb}")
    if nb in (64, 32) and cfg_id != 3 or (cfg_id, nb) == (3, 64):
        check_vs_c_oracle(torch_cuda, net, shp, mode="cc", top_k=cfg.top_k, max_dets=cfg.max_dets, conf=cfg.conf_thresh,
                          nthreads=16, label=f"cfg{cfg_id}-b{nb}-cc")
    _ = cfg


@pytest.mark.parametrize("nb", [9, 12, 20])
def test_batches_that_pad_the_xcd_placement(torch_cuda, nb):
    """batch >= 8 and not a multiple of 8: mask_assemble's grid is padded to whole groups of 8 images and
    the padding workgroups must return without touching anything."""
    from yolact_amd import synth
    net, shp = synth.make_toy(40 + nb, batch=nb, proto_hw=(23, 31))
    check_vs_c_oracle(torch_cuda, net, shp, mode="per_class", top_k=50, max_dets=20, label=f"pad-{nb}")
    check_vs_c_oracle(torch_cuda, net, shp, mode="cc", top_k=15, max_dets=15, label=f"pad-{nb}")


@pytest.mark.parametrize("prefill", [0, 1])
def test_untouched_and_zero_rows_contract(torch_cuda, prefill, monkeypatch):
    """include/yolact_post.h, zero_pad_masks = 0: rows beyond the rows a detection can land in keep what
    the caller had there; rows in [num_det, that bound) are zeros, whether the zero-fill rode in the NMS
    launch (YP_TUNE_PREFILL=1, the default) or ran as a launch of its own (0).  Both placements are
    bit-exact on the detections' rows."""
    import torch
    from oracle import c_oracle as co
    from yolact_amd import PostConfig, PostProcessor, synth
    monkeypatch.setenv("YP_TUNE_PREFILL", str(prefill))          # read once, in yp_create
    net, shp = synth.make_toy(5, batch=9)
    B, A, C = net["pred_cls"].shape
    for mode in ("cc", "per_class"):
        pp = PostProcessor(PostConfig(A, C, 12, 12, 64, 64, nms_mode=mode, top_k=15, max_dets=40, max_batch=B))
        dev = {k: torch.from_numpy(v).cuda() for k, v in net.items()}
        out = pp.alloc_outputs(B)
        out["detection_masks"].fill_(7.0)
        pp.run(dev["pred_offset"], dev["pred_cls"], dev["pred_mask_coef"], dev["priors"], dev["proto_out"], out=out)
        pp.sync()
        got = to_numpy(out)
        compare(got, co.detect(net, shp, top_k=15, nms_mode=mode, max_dets=40), exact_values=True, label=f"prefill{prefill}-{mode}")
        m, n = got["detection_masks"], got["num_detections"]
        bound = 15 if mode == "cc" else 40
        for b in range(B):
            assert n[b] <= bound and (m[b, bound:] == 7.0).all()
            tail = m[b, n[b]:bound]
            assert not tail.any()
        pp.close()


def test_current_device_is_left_alone(torch_cuda):
    """Every entry point restores the calling thread's HIP device (matters with one handle per GPU);
    on the one-GPU box this checks that nothing else changes either."""
    import torch
    from yolact_amd import Detect, synth
    from yolact_amd.sharding import ShardedDetect
    net, shp = synth.make_toy(2)
    before = torch.cuda.current_device()
    det = Detect(5, 300, 100, 0.05, 0.5)
    det(net, shp)
    for pp in det._pp.values():
        pp.close()
    sd = ShardedDetect(5, 300, 100, 0.05, 0.5, devices=[0, 0])
    sd(net, shp)
    assert torch.cuda.current_device() == before


def test_yp_run_under_hip_graph_capture(torch_cuda):
    """yp_run allocates nothing, creates nothing and never synchronises, so a caller can capture it in a
    HIP graph (serving loops): captured once, replayed on new inputs written into the same buffers, the
    outputs are bit-identical to eager calls.  The allocation-free entry point is captured the same way
    (after one eager call, which resolves the static scratch's address)."""
    import ctypes
    torch = torch_cuda
    from yolact_amd import _lib, PostConfig, PostProcessor, synth
    netA, shp = synth.make_toy(51, batch=4)
    netB, _ = synth.make_toy(52, batch=4)
    B, A, C = netA["pred_cls"].shape
    for mode in ("cc", "per_class"):
        pc = PostConfig(A, C, 12, 12, 64, 64, nms_mode=mode, top_k=40, max_dets=20, max_batch=B)
        pp = PostProcessor(pc)
        keys = ("pred_offset", "pred_cls", "pred_mask_coef", "priors", "proto_out")
        buf = {k: torch.from_numpy(netA[k]).cuda() for k in keys}
        eager = {}
        for name, net in (("A", netA), ("B", netB)):
            for k in keys:
                buf[k].copy_(torch.from_numpy(net[k]))
            eager[name] = {k: v.clone() for k, v in pp.run(*[buf[k] for k in keys]).items()}
        torch.cuda.synchronize()
        out = pp.alloc_outputs(B)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            pp.run(*[buf[k] for k in keys], out=out)
        for name, net in (("B", netB), ("A", netA), ("B", netB)):
            for k in keys:
                buf[k].copy_(torch.from_numpy(net[k]))
            g.replay()
            torch.cuda.synchronize()
            n = eager[name]["num_detections"].cpu().numpy()
            for k in ("detection_boxes", "detection_classes", "detection_scores", "detection_anchor_indices", "num_detections"):
                assert torch.equal(out[k], eager[name][k]), (mode, name, k)
            for b in range(B):
                assert torch.equal(out["detection_masks"][b, :n[b]], eager[name]["detection_masks"][b, :n[b]]), (mode, name, b)
        # the allocation-free entry point under capture
        L = _lib.load()
        c = _lib.YpConfig(num_anchors=A, num_classes=C, mask_dim=32, proto_h=12, proto_w=12, img_h=64, img_w=64, conf_thresh=0.05,
                          nms_thresh=0.5, nms_mode={"cc": 0, "per_class": 1}[mode], top_k=40, max_dets=20, use_cropped_mask=1,
                          zero_pad_masks=0, mask_threshold=-1.0, max_batch=1, device_id=0)
        ptr = lambda t: ctypes.c_void_p(t.data_ptr())

        def call_static():
            rc = L.yp_run_static(ctypes.byref(c), B, *[ptr(buf[k]) for k in keys], ptr(out["detection_boxes"]),
                                 ptr(out["detection_classes"]), ptr(out["detection_scores"]), ptr(out["detection_masks"]), None,
                                 ptr(out["num_detections"]), ptr(out["detection_anchor_indices"]),
                                 ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
            assert rc == 0, L.yp_last_error()

        call_static()                       # resolves the scratch address, plans the configuration
        torch.cuda.synchronize()
        g2 = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g2):
            call_static()
        for name, net in (("A", netA), ("B", netB)):
            for k in keys:
                buf[k].copy_(torch.from_numpy(net[k]))
            for v in out.values():
                v.zero_()
            g2.replay()
            torch.cuda.synchronize()
            for k in ("detection_boxes", "detection_scores", "detection_anchor_indices", "num_detections"):
                assert torch.equal(out[k], eager[name][k]), (mode, name, k, "static")
        pp.close()
