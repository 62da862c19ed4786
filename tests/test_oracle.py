"""The oracle itself: hand-written known-answer tests (the reference has none), the
numpy restatement against the C restatement, and both against the committed golden fixtures."""
import glob
import os

import numpy as np
import pytest

from oracle import c_oracle as co
from oracle import detect_oracle as po
from yolact_amd import synth
from yolact_amd.anchors import feature_map_sizes, make_priors, proto_size

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
f32 = np.float32


def _boxes(*rows):
    return np.array(rows, dtype=f32)


# ---- utils/utils.py known answers ------------------------------------------------------------

def test_decode_zero_offsets_is_prior():
    pri = _boxes([10, 20, 50, 80], [0, 0, 24, 24])
    out = po._decode(np.zeros((2, 4), f32), pri)
    assert np.allclose(out, pri, atol=1e-5)
    for i in range(2):
        buf = np.zeros(4, f32)
        co.lib().yo_decode(co._p(np.zeros(4, f32)), co._p(pri[i].copy()), co._p(buf))
        assert np.allclose(buf, pri[i], atol=1e-5)


def test_iou_known_answers():
    a = _boxes([0, 0, 10, 10])
    assert po._iou(a, a)[0, 0] == f32(1.0)
    assert po._iou(a, _boxes([20, 20, 30, 30]))[0, 0] == 0.0            # disjoint
    assert po._iou(a, _boxes([0, 10, 10, 20]))[0, 0] == 0.0             # touching edge: inter == 0 branch
    assert po._iou(a, _boxes([0, 5, 10, 15]))[0, 0] == f32(50.0 / 150.0)
    L = co.lib()
    for b, want in (([0, 0, 10, 10], 1.0), ([20, 20, 30, 30], 0.0), ([0, 10, 10, 20], 0.0)):
        assert L.yo_iou(co._p(a[0].copy()), co._p(np.array(b, f32))) == want


def test_sanitize_and_crop_bounds():
    x1, x2 = po.sanitize_coordinates(np.array([30.0], f32), np.array([10.0], f32), 20.0, normalized=False)
    assert (x1[0], x2[0]) == (10.0, 20.0)                                # swapped + clipped to size
    # crop: normalised coordinates, 1 px padding
    bx1, bx2, by1, by2 = po.crop_bounds(_boxes([0.25, 0.5, 0.75, 0.5]), 8, 8)
    assert (bx1[0], bx2[0], by1[0], by2[0]) == (3.0, 5.0, 1.0, 7.0)      # degenerate x: two padded columns


def test_crop_full_image_keeps_everything_and_degenerate_keeps_two_columns():
    m = np.ones((6, 8, 2), f32)
    out = po.crop(m, _boxes([0, 0, 1, 1], [0, 0.5, 1, 0.5]))
    assert out[:, :, 0].sum() == 48
    assert out[:, :, 1].sum() == 6 * 2 and out[:, 3:5, 1].all()


def _two_class_net(logit_rows, boxes_px, img=64, proto=8):
    """A tiny problem with hand-picked foreground logits; priors == wanted boxes, zero offsets."""
    A = len(logit_rows)
    cls = np.array(logit_rows, f32)[None]
    net = {"pred_offset": np.zeros((1, A, 4), f32), "pred_cls": cls,
           "pred_mask_coef": np.zeros((1, A, 32), f32), "priors": np.array(boxes_px, f32),
           "proto_out": np.zeros((1, proto, proto, 32), f32)}
    return net, np.array([1, img, img, 3], np.int32)


@pytest.mark.parametrize("impl", ["numpy", "c"])
def test_nms_suppression_threshold_and_ties(impl):
    run = (lambda n, s, **k: po.detect(n, s, max_output_size=k.get("max_dets", 15), **k)) if impl == "numpy" else co.detect
    # anchors 0/1 identical boxes (iou 1 -> lower score suppressed); anchor 2 overlaps anchor 0 with
    # iou exactly 0.5 (kept, '<='); anchor 3 far away with the same score as anchor 2 (tie -> lower index first)
    boxes = [[0, 0, 10, 10], [0, 0, 10, 10], [0, 0, 10, 5], [40, 40, 50, 50]]
    logits = [[0, 3.0], [0, 2.0], [0, 1.0], [0, 1.0]]
    net, shp = _two_class_net(logits, boxes)
    out = run(net, shp, conf_thresh=0.05, top_k=15, nms_mode="cc", max_dets=15)
    assert out["num_detections"][0] == 3
    assert list(out["detection_anchor_indices"][0][:3]) == [0, 2, 3]
    assert np.all(np.diff(out["detection_scores"][0][:3]) <= 0)
    assert list(out["detection_classes"][0][:3]) == [1.0, 1.0, 1.0]
    assert list(out["detection_anchor_indices"][0][3:]) == [-1] * 12


@pytest.mark.parametrize("impl", ["numpy", "c"])
def test_conf_is_strict_and_empty_image(impl):
    run = (lambda n, s, **k: po.detect(n, s, max_output_size=k.get("max_dets", 15), **k)) if impl == "numpy" else co.detect
    # equal logits over 2 classes -> p = 0.5 exactly; conf 0.5 must drop it ('>' at detection.py:73)
    net, shp = _two_class_net([[1.0, 1.0], [0.0, 0.0]], [[0, 0, 10, 10], [20, 20, 30, 30]])
    out = run(net, shp, conf_thresh=0.5, top_k=15, nms_mode="cc", max_dets=15)
    assert out["num_detections"][0] == 0 and out["num_survivors"][0] == 0
    assert not out["detection_scores"].any() and not out["detection_boxes"].any()
    out = run(net, shp, conf_thresh=0.4999, top_k=15, nms_mode="cc", max_dets=15)
    assert out["num_detections"][0] == 2 and out["num_survivors"][0] == 2      # S < top_k


@pytest.mark.parametrize("impl", ["numpy", "c"])
def test_per_class_with_one_foreground_class_equals_cross_class(impl):
    run = (lambda n, s, **k: po.detect(n, s, max_output_size=k.get("max_dets", 40), **k)) if impl == "numpy" else co.detect
    net, shp = synth.make_toy(5, num_classes=2)
    a = run(net, shp, conf_thresh=0.05, top_k=40, nms_mode="cc", max_dets=40)
    b = run(net, shp, conf_thresh=0.05, top_k=40, nms_mode="per_class", max_dets=40)
    for k in ("num_detections", "detection_anchor_indices", "detection_classes", "detection_scores", "detection_boxes"):
        assert np.array_equal(a[k], b[k]), k


# ---- YP-MATH pieces of the C oracle ------------------------------------------------------------

def test_c_exp_matches_libm():
    xs = np.concatenate([np.linspace(-87, 87, 20001), np.random.default_rng(0).normal(0, 3, 20000)]).astype(f32)
    got = np.array([co.exp(float(x)) for x in xs], dtype=np.float64)
    want = np.exp(xs.astype(np.float64))
    rel = np.abs(got - want) / want
    assert rel.max() < 2.5e-7, rel.max()
    assert co.exp(0.0) == 1.0


def test_c_softmax_row_matches_numpy():
    rng = np.random.default_rng(1)
    for C in (2, 5, 16, 17, 81, 91, 256):
        x = rng.normal(0, 3, C).astype(f32)
        p = np.zeros(C, f32)
        co.lib().yo_softmax_row(co._p(x), C, co._p(p))
        want = po.softmax_lastaxis(x.astype(np.float64))
        assert np.abs(p - want).max() < 2e-7
        assert abs(float(p.sum(dtype=np.float64)) - 1.0) < 1e-5


# ---- the two restatements against each other and against the fixtures -------------------------

@pytest.mark.parametrize("seed", [0, 1, 2, 3, 4])
@pytest.mark.parametrize("mode,top_k,max_dets", [("cc", 15, 15), ("cc", 200, 100), ("per_class", 200, 100)])
def test_c_oracle_equals_numpy_oracle_toy(seed, mode, top_k, max_dets):
    net, shp = synth.make_toy(seed)
    a = po.detect(net, shp, top_k=top_k, nms_mode=mode, max_output_size=max_dets, max_dets=max_dets)
    b = co.detect(net, shp, top_k=top_k, nms_mode=mode, max_dets=max_dets)
    assert np.array_equal(a["num_survivors"], b["num_survivors"])
    assert np.array_equal(a["num_detections"], b["num_detections"])
    assert np.array_equal(a["detection_anchor_indices"], b["detection_anchor_indices"])
    assert np.array_equal(a["detection_classes"], b["detection_classes"])
    for k in ("detection_scores", "detection_boxes", "detection_masks"):
        assert np.abs(a[k] - b[k]).max() <= 1e-6, k                    # libm exp / numpy sums vs YP-MATH


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLDEN, "toy_*.npz"))))
def test_golden_toy_fixtures(path):
    z = np.load(path)
    net = {k[3:]: z[k] for k in z.files if k.startswith("in_")}
    kw = dict(conf_thresh=float(z["conf"]), top_k=int(z["top_k"]), nms_mode=str(z["nms_mode"]), max_dets=int(z["max_dets"]))
    a = po.detect(net, z["img_shape"], max_output_size=int(z["max_dets"]), **kw)
    b = co.detect(net, z["img_shape"], **kw)
    for got, exact in ((a, True), (b, False)):
        for k in ("num_detections", "num_survivors", "detection_anchor_indices", "detection_classes"):
            assert np.array_equal(got[k], z["out_" + k]), (path, k)
        for k in ("detection_scores", "detection_boxes", "detection_masks"):
            d = np.abs(got[k] - z["out_" + k]).max()
            assert d <= (1e-7 if exact else 1e-6), (path, k, d)


def test_golden_config2_image0():
    """BASELINE config 2, image 0 (19248 anchors, 138x138 protos), all three NMS settings."""
    z = np.load(os.path.join(GOLDEN, "config2_image0.npz"))
    cfg = synth.CONFIGS[2]
    net, shp = synth.make_batch(cfg, batch=1)
    for mode, top_k, max_dets in (("cc", 15, 15), ("cc", 200, 100), ("per_class", 200, 100)):
        tag = f"{mode}_{top_k}"
        b = co.detect(net, shp, conf_thresh=cfg.conf_thresh, top_k=top_k, nms_mode=mode, max_dets=max_dets)
        for k in ("num_detections", "num_survivors", "detection_anchor_indices", "detection_classes"):
            assert np.array_equal(b[k], z[f"{tag}_{k}"]), (tag, k)
        for k in ("detection_scores", "detection_boxes"):
            assert np.abs(b[k] - z[f"{tag}_{k}"]).max() <= 1e-6, (tag, k)
        m = b["detection_masks"][0].reshape(max_dets, -1)
        assert np.abs(m.sum(axis=1, dtype=np.float64) - z[f"{tag}_mask_sums"]).max() < 1e-2
        assert np.abs((m > 0).sum(axis=1) - z[f"{tag}_mask_nonzero"]).max() == 0


# ---- priors (data/anchor.py) -----------------------------------------------------------------

def test_priors_match_loop_restatement():
    for (h, w, A) in ((550, 550, 19248), (768, 1024, 49104), (800, 1333, 66900)):
        fm = feature_map_sizes(h, w)
        pri = make_priors(h, w)
        assert pri.shape == (A, 4)
        if A == 19248:
            ref = po.generate_anchors(h, w, fm, [1, 0.5, 2], [24, 48, 96, 192, 384])
            assert np.array_equal(ref, pri)
    assert feature_map_sizes(550, 550) == [(69, 69), (35, 35), (18, 18), (9, 9), (5, 5)]
    assert proto_size(550, 550) == (138, 138) and proto_size(800, 1333) == (200, 334)


def test_kept_index_agreement_between_the_two_arithmetics():
    """Unfiltered seeds of BASELINE config 2: the C oracle (YP-MATH contract) against the numpy oracle
    (libm exp, numpy reductions, BLAS matmul).  north_star target: >= 95 % kept-index agreement.
    Measured on 64 images: the kept SET is identical on every image and 99.3 % of the output rows are
    identical; whole images are identical row by row only 70-92 % of the time, because saturated
    scores (p ~ 0.999, one ulp = 6e-8) that differ by an ulp between the two exp implementations
    swap neighbouring rows."""
    from tests.util import kept_index_agreement
    from yolact_amd import synth
    from oracle import c_oracle as co, detect_oracle as po
    cfg = synth.CONFIGS[2]
    net, shp = synth.make_batch(cfg, batch=16, first_image=100)
    for mode, top_k, md in (("per_class", 200, 100), ("cc", 200, 100), ("cc", 15, 15)):
        c = co.detect(net, shp, conf_thresh=cfg.conf_thresh, top_k=top_k, nms_mode=mode, max_dets=md, nthreads=8)
        r = po.detect(net, shp, conf_thresh=cfg.conf_thresh, top_k=top_k, nms_mode=mode, max_output_size=md, max_dets=md)
        rate = kept_index_agreement(c, r)
        assert rate["per_detection"] >= 0.95 and rate["kept_set"] >= 0.95, (mode, rate)
        assert np.abs(np.sort(c["detection_scores"], axis=1) - np.sort(r["detection_scores"], axis=1)).max() <= 1e-5


def test_exact_agreement_on_unsaturated_scores():
    """The "deterministic" setting of SURVEY.md 8d: with foreground probabilities spread over 0.2-0.99
    instead of saturating at 0.999 (synth mode "planted_soft"), no two scores are an ulp apart, and the
    two arithmetics keep the same anchors, classes and ORDER on every image."""
    from tests.util import kept_index_agreement
    from yolact_amd import synth
    from oracle import c_oracle as co, detect_oracle as po
    cfg = synth.CONFIGS[2]
    net, shp = synth.make_batch(cfg, batch=16, first_image=300, mode="planted_soft")
    for mode, top_k, md in (("per_class", 200, 100), ("cc", 200, 100), ("cc", 15, 15)):
        c = co.detect(net, shp, conf_thresh=cfg.conf_thresh, top_k=top_k, nms_mode=mode, max_dets=md, nthreads=8)
        r = po.detect(net, shp, conf_thresh=cfg.conf_thresh, top_k=top_k, nms_mode=mode, max_output_size=md, max_dets=md)
        assert kept_index_agreement(c, r) == {"per_detection": 1.0, "kept_set": 1.0, "exact_order": 1.0}, mode
        assert np.abs(c["detection_scores"] - r["detection_scores"]).max() <= 1e-5
        assert np.abs(c["detection_boxes"] - r["detection_boxes"]).max() <= 1e-5


def test_planted_soft_batch_has_margin_on_every_decision_that_shapes_the_output():
    """SURVEY.md 8(d), "deterministic configs": on the `planted_soft` images that the GPU suite compares row by row
    with the independent (libm / numpy) arithmetic, every decision that can change an output row has margin in
    float64 (oracle/margins.py) well above what two float32 evaluations of the same expression differ by:
      confidence compare where it can reach the output   > 1e-4   (p ~ 1e-7)
      iou_max vs the NMS threshold                        > 1e-4   (iou ~ 1e-6)
      score gap at a top_k / max_dets cut                 > 1e-6   (scores ~ 1e-7)
      gap between neighbouring kept scores (row order)    > 1e-6
      crop edge to the next integer pixel coordinate      > 2e-5   (edges are <= 138: a few ulp are ~3e-5 at most;
                                                                    SURVEY's 1e-3 cannot hold over the ~400 edges of an image
                                                                    with 100 detections - tools/margins.py prints the minima)
    The plain config-2 images do NOT pass (saturated scores an ulp apart: `order` ~ 2e-8), which is why the exact-order
    comparison across arithmetics runs on planted_soft and the config-2 one reports agreement rates."""
    from oracle import margins
    from yolact_amd import synth
    cfg = synth.CONFIGS[2]
    net, shp = synth.make_batch(cfg, batch=2, first_image=300, mode="planted_soft")
    for nms, top_k, md in (("per_class", 200, 100), ("cc", 15, 15)):
        m = margins.batch_margins(net, shp, conf_thresh=cfg.conf_thresh, nms_thresh=cfg.nms_thresh, top_k=top_k, nms_mode=nms, max_dets=md)["min"]
        assert m["conf_relevant"] > 1e-4 and m["iou"] > 1e-4 and m["cut"] > 1e-6 and m["order"] > 1e-6 and m["crop_edge"] > 2e-5, (nms, m)
    hard = margins.batch_margins(*synth.make_batch(cfg, batch=2), conf_thresh=cfg.conf_thresh, top_k=200, nms_mode="per_class", max_dets=100)["min"]
    assert hard["order"] < 1e-6                       # the headline batch is not deterministic in this sense
