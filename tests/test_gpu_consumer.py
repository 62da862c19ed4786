"""yp_masks_to_rle + export_detections on a real MI355X against the consumer-step oracle: bitmaps and
run-length counts bit-exact (byte / integer work), both threshold orders of the reference's callers."""
import numpy as np
import pytest

from oracle import consumer_oracle as co

pytestmark = pytest.mark.gpu


def _run(masks, num_det, out_hw, order="resize_then_threshold", threshold=0.5, rle_cap=None, bitmaps=True):
    import torch
    from yolact_amd import consumer
    m = torch.from_numpy(np.ascontiguousarray(masks, np.float32)).cuda()
    n = torch.from_numpy(np.asarray(num_det, np.int32)).cuda()
    r = consumer.masks_to_rle(m, n, out_hw, threshold=threshold, order=order, bitmaps=bitmaps, rle_cap=rle_cap)
    torch.cuda.synchronize()
    return r


def _check(masks, num_det, out_hw, order, threshold=0.5, rle_cap=None):
    r = _run(masks, num_det, out_hw, order, threshold, rle_cap)
    counts = r["counts"].cpu().numpy().view(np.uint32)
    lens = r["len"].cpu().numpy()
    bm = r["bitmaps"].cpu().numpy()
    ref = co.masks_to_rle(masks, num_det, out_hw[0], out_hw[1], threshold, co.RESIZE_THEN_THRESHOLD if order == "resize_then_threshold" else co.THRESHOLD_THEN_RESIZE)
    for b in range(masks.shape[0]):
        for n in range(masks.shape[1]):
            if n < num_det[b]:
                rb, rc = ref[b][n]
                assert np.array_equal(bm[b, n], rb), (b, n, "bitmap", int((bm[b, n] != rb).sum()))
                assert lens[b, n] == rc.shape[0], (b, n, lens[b, n], rc.shape[0])
                assert np.array_equal(counts[b, n, :lens[b, n]], rc), (b, n, "counts")
            else:
                assert lens[b, n] == 0
    return r


def _blobs(rng, B, N, h, w):
    """Soft discs and noise: float masks in [0,1] like sigmoid outputs cropped to a box."""
    yy, xx = np.mgrid[0:h, 0:w]
    out = np.zeros((B, N, h, w), np.float32)
    for b in range(B):
        for n in range(N):
            cy, cx, r = rng.uniform(0, h), rng.uniform(0, w), rng.uniform(1.0, 1.0 + 0.4 * max(h, w))
            m = 1.0 / (1.0 + np.exp(-(r - np.hypot(yy - cy, xx - cx))))
            m = m + rng.normal(0, 0.15, (h, w)) * (rng.random() < 0.5)
            y1, y2 = sorted(rng.integers(0, h, 2)); x1, x2 = sorted(rng.integers(0, w, 2))
            box = np.zeros((h, w), bool); box[y1:y2 + 1, x1:x2 + 1] = True
            out[b, n] = np.clip(m, 0, 1) * box
    return out


@pytest.mark.parametrize("order", ["resize_then_threshold", "threshold_then_resize"])
def test_known_answer_masks(order):
    h, w = 6, 5
    masks = np.zeros((1, 6, h, w), np.float32)
    masks[0, 1] = 1.0                                     # all one
    masks[0, 2, 3, 2] = 1.0                               # single pixel
    masks[0, 3] = (np.add.outer(np.arange(h), np.arange(w)) % 2)   # checkerboard
    masks[0, 4] = 0.5                                     # exactly the threshold: strict '>' keeps nothing
    masks[0, 5, :, 0] = 0.9                               # first column set: the encoding starts with a 0
    r = _check(masks, [6], (h, w), order)                 # identity size: cv2.resize copies
    counts, lens = r["counts"].cpu().numpy().view(np.uint32), r["len"].cpu().numpy()
    assert counts[0, 0, :lens[0, 0]].tolist() == [30]
    assert counts[0, 1, :lens[0, 1]].tolist() == [0, 30]
    assert counts[0, 2, :lens[0, 2]].tolist() == [2 * h + 3, 1, 30 - 2 * h - 4]
    assert lens[0, 3] == 26 and counts[0, 4, :lens[0, 4]].tolist() == [30]   # h even: runs merge across the 4 column borders
    assert counts[0, 5, :lens[0, 5]].tolist() == [0, h, 30 - h]
    _check(masks, [6], (24, 20), order)                   # 4x enlargement of the same masks
    _check(masks, [4], (7, 9), order)                     # odd sizes, fewer detections than rows
    _check(masks, [0], (12, 12), order)                   # no detection at all


@pytest.mark.parametrize("order", ["resize_then_threshold", "threshold_then_resize"])
def test_random_masks_small_sizes(order):
    rng = np.random.default_rng(11)
    for case in range(12):
        h, w = int(rng.integers(2, 40)), int(rng.integers(2, 40))
        oh, ow = int(rng.integers(h, 4 * h + 3)), int(rng.integers(w, 4 * w + 3))
        B, N = int(rng.integers(1, 3)), int(rng.integers(1, 6))
        masks = _blobs(rng, B, N, h, w) if case % 2 else rng.random((B, N, h, w)).astype(np.float32)
        nd = rng.integers(0, N + 1, B)
        _check(masks, nd, (oh, ow), order)


def test_out_height_around_the_64_row_chunks():
    rng = np.random.default_rng(12)
    masks = rng.random((1, 3, 9, 7)).astype(np.float32)
    for oh in (63, 64, 65, 127, 128, 129):
        _check(masks, [3], (oh, 33), "resize_then_threshold")


@pytest.mark.parametrize("order", ["resize_then_threshold", "threshold_then_resize"])
@pytest.mark.parametrize("threshold", [0.0, 0.5, -0.25])
def test_zero_skipping_does_not_change_the_encoding(order, threshold):
    """The kernel skips columns and 64-row chunks whose source taps are all zero (threshold >= 0; a negative
    threshold sets every such pixel and takes the plain path): all-zero masks, single pixels in the corners
    and in the middle, box-limited blobs, negative values and -0.0 outside the box, at a size with several
    column batches and row chunks."""
    rng = np.random.default_rng(15)
    h, w, oh, ow = 37, 41, 200, 150
    masks = np.zeros((1, 8, h, w), np.float32)
    masks[0, 1, 0, 0] = 0.9; masks[0, 2, h - 1, w - 1] = 0.9; masks[0, 3, h // 2, w // 2] = 0.7
    masks[0, 4] = _blobs(rng, 1, 1, h, w)[0, 0]
    masks[0, 5] = -np.abs(_blobs(rng, 1, 1, h, w)[0, 0])            # negative inside the box, -0.0 / 0.0 outside
    masks[0, 6, 5:9, :] = rng.random((4, w)); masks[0, 6, 30:33, 10:12] = 1.0
    masks[0, 7, :, 20] = rng.random(h)
    _check(masks, [8], (oh, ow), order, threshold=threshold)


@pytest.mark.parametrize("dst", [(1100, 300), (1100, 600), (300, 1100)])
def test_tall_and_wide_outputs_take_the_other_kernel_variants(dst):
    """Outputs taller than 1024 rows run the 1024-thread workgroups; outputs whose bits exceed 64 KiB of LDS write
    the bitmap batch by batch instead of once at the end.  Both threshold orders, bitmaps on."""
    rng = np.random.default_rng(16)
    masks = _blobs(rng, 1, 3, 20, 24)
    masks[0, 2] = rng.random((20, 24)).astype(np.float32)           # no zero skipping to hide behind
    for order in ("resize_then_threshold", "threshold_then_resize"):
        _check(masks, [3], dst, order, rle_cap=dst[0] * dst[1] + 1)


def test_counts_capacity_overflow_is_reported_and_retried():
    import torch
    from yolact_amd import _lib
    rng = np.random.default_rng(13)
    masks = rng.random((1, 2, 16, 16)).astype(np.float32)       # noise: hundreds of runs
    r = _check(masks, [2], (40, 40), "resize_then_threshold", rle_cap=4)   # the wrapper retries with room
    assert int(r["len"].max()) > 4
    # the raw entry point stores only rle_cap counts and reports the true length
    L = _lib.load()
    m = torch.from_numpy(masks).cuda(); n = torch.tensor([2], dtype=torch.int32).cuda()
    counts = torch.full((1, 2, 4), -1, dtype=torch.int32).cuda(); lens = torch.zeros((1, 2), dtype=torch.int32).cuda()
    assert L.yp_masks_to_rle(0, 1, 2, 16, 16, m.data_ptr(), n.data_ptr(), 40, 40, 0.5, 0, None, counts.data_ptr(), lens.data_ptr(), 4, None) == 0
    torch.cuda.synchronize()
    ref = co.masks_to_rle(masks, [2], 40, 40)
    for k in range(2):
        assert int(lens[0, k]) == ref[0][k][1].shape[0]
        assert np.array_equal(counts[0, k].cpu().numpy().view(np.uint32), ref[0][k][1][:4])
    assert L.yp_masks_to_rle(0, 1, 2, 600, 16, m.data_ptr(), n.data_ptr(), 40, 40, 0.5, 0, None, counts.data_ptr(), lens.data_ptr(), 4, None) == -1


@pytest.mark.parametrize("src,dst", [((138, 138), (550, 550)), ((200, 334), (800, 1333))])
@pytest.mark.parametrize("order", ["resize_then_threshold", "threshold_then_resize"])
def test_reference_sizes(src, dst, order):
    """The proto -> image sizes of BASELINE configs 2 and 5 (coco_eval.py:134 resizes to the original
    image; the network sizes stand in for it)."""
    rng = np.random.default_rng(14)
    masks = _blobs(rng, 1, 3, *src)
    _check(masks, [3], dst, order)


def test_end_to_end_detect_to_annotations():
    """Detect -> export_detections against the oracle chain (Detect oracle -> consumer oracle):
    coco_eval.py:123-155's dicts, two images with different original sizes - one resize + RLE call for both."""
    import torch
    from oracle import c_oracle
    from yolact_amd import Detect, consumer, synth
    net, shp = synth.make_toy(5, batch=2, proto_hw=(18, 22))
    det = Detect(5, 300, 100, 0.05, 0.5)
    res = det(net, shp)
    torch.cuda.synchronize()
    orig = [(61, 47), (96, 130)]
    got, boxes = consumer.export_detections(res, ["img0", 7], orig, return_boxes=True)
    ref_det = c_oracle.detect(net, shp, top_k=15, nms_mode="cc", max_dets=300)
    ref = co.export_detections(ref_det, ["img0", 7], orig)
    assert len(got) == len(ref) == len(boxes) > 0
    for g, r in zip(got, ref):
        assert g == r, (g, r)
    import json
    json.dumps(got)                                                  # coco_eval.py:158-159: it must serialise


def test_export_detections_emits_exactly_the_reference_keys():
    """coco_eval.py:150-155 appends {'image_id', 'category_id', 'segmentation', 'score'} and nothing else; the list
    below is written out by hand from a Detect result built by hand (no oracle in between): image 0 has two
    detections above the 0.05 score threshold of L123 - a 2x2 block and a full mask -, image 1 none."""
    import torch
    from yolact_amd import consumer
    masks = np.zeros((2, 3, 4, 4), np.float32)
    masks[0, 0, 1:3, 1:3] = 0.9
    masks[0, 1] = 0.8
    masks[0, 2] = 0.7                                   # score 0.04: below the threshold, never exported
    res = {
        "detection_masks": torch.from_numpy(masks).cuda(),
        "detection_scores": torch.tensor([[0.9, 0.25, 0.04], [0.0, 0.0, 0.0]], dtype=torch.float32).cuda(),
        "detection_classes": torch.tensor([[3.0, 17.0, 5.0], [0.0, 0.0, 0.0]], dtype=torch.float32).cuda(),
        "detection_boxes": torch.tensor([[[0.25, 0.25, 0.75, 0.75], [0.0, 0.0, 1.0, 1.0], [0.0, 0.0, 0.5, 0.5]],
                                         [[0.0] * 4] * 3], dtype=torch.float32).cuda(),
    }
    got, boxes = consumer.export_detections(res, [42, "b"], [(4, 4), (9, 5)], return_boxes=True)   # identity size: cv2.resize copies
    # column-major runs of the 2x2 block in a 4x4 image: 5 zeros, 2 ones, 2 zeros, 2 ones, 5 zeros -> rleToString
    expected = [
        {"image_id": 42, "category_id": 3, "segmentation": {"size": [4, 4], "counts": "52203"}, "score": float(np.float32(0.9))},
        {"image_id": 42, "category_id": 17, "segmentation": {"size": [4, 4], "counts": "0`0"}, "score": 0.25},
    ]
    assert got == expected
    assert [sorted(d) for d in got] == [["category_id", "image_id", "score", "segmentation"]] * 2
    assert boxes == [[1, 1, 3, 3], [0, 0, 4, 4]]
    assert consumer.export_detections(res, [42, "b"], [(4, 4), (9, 5)]) == expected


@pytest.mark.parametrize("order", ["resize_then_threshold", "threshold_then_resize"])
def test_per_image_output_sizes_in_one_call(order):
    """coco_eval.py:112-137: every image's masks go to that image's own (_h, _w).  yp_masks_to_rle_sized takes the
    sizes of a whole batch (more images than one launch's argument block holds: 32) and must give, image by image,
    what the uniform-size entry point gives for that size - enlarging, shrinking and identity sizes mixed."""
    import torch
    from yolact_amd import consumer
    rng = np.random.default_rng(17)
    B, N, h, w = 37, 3, 14, 19
    masks = _blobs(rng, B, N, h, w)
    masks[::5] = rng.random((len(range(0, B, 5)), N, h, w)).astype(np.float32)
    nd = rng.integers(0, N + 1, B).astype(np.int32)
    sizes = [(int(rng.integers(3, 90)), int(rng.integers(3, 120))) for _ in range(B)]
    sizes[3] = (h, w)                                                 # identity
    sizes[4] = (7, 19); sizes[5] = (14, 9)                            # half of ONE axis only: still INTER_LINEAR
    sizes = [(oh + 1, ow) if (2 * oh, 2 * ow) == (h, w) else (oh, ow) for oh, ow in sizes]
    m = torch.from_numpy(masks).cuda(); n = torch.from_numpy(nd).cuda()
    r = consumer.masks_to_rle(m, n, sizes, order=order, rle_cap=64)    # small cap: the retry path too
    torch.cuda.synchronize()
    counts, lens = r["counts"].cpu().numpy().view(np.uint32), r["len"].cpu().numpy()
    o = co.RESIZE_THEN_THRESHOLD if order == "resize_then_threshold" else co.THRESHOLD_THEN_RESIZE
    for b in range(B):
        ref = co.masks_to_rle(masks[b:b + 1], nd[b:b + 1], sizes[b][0], sizes[b][1], 0.5, o)[0]
        for k in range(N):
            if k < nd[b]:
                assert lens[b, k] == ref[k][1].shape[0], (b, k, sizes[b])
                assert np.array_equal(counts[b, k, :lens[b, k]], ref[k][1]), (b, k, sizes[b])
            else:
                assert lens[b, k] == 0
    dicts = consumer.rle_dicts(r)
    assert all(d["size"] == [sizes[b][0], sizes[b][1]] for b in range(B) for d in dicts[b])


@pytest.mark.parametrize("order", ["resize_then_threshold", "threshold_then_resize"])
def test_outputs_smaller_than_the_mask(order):
    """cv::resize with INTER_LINEAR shrinks with two taps and no anti-aliasing; the staging of the source columns
    and the zero skipping must cope with scale factors above 1 (up to 13x here).  Exactly half in BOTH axes is
    the one pair cv::resize hands to INTER_AREA: rejected by the library (and by the oracle)."""
    import torch
    from yolact_amd import _lib
    rng = np.random.default_rng(18)
    masks = _blobs(rng, 2, 4, 66, 80)
    masks[1, 0] = rng.random((66, 80)).astype(np.float32)
    for dst in ((33, 41), (34, 40), (65, 79), (5, 6), (20, 200), (130, 30), (66, 40), (33, 80)):
        _check(masks, [4, 3], dst, order)
    L = _lib.load()
    m = torch.from_numpy(masks).cuda(); n = torch.tensor([4, 3], dtype=torch.int32).cuda()
    counts = torch.zeros((2, 4, 64), dtype=torch.int32).cuda(); lens = torch.zeros((2, 4), dtype=torch.int32).cuda()
    assert L.yp_masks_to_rle(0, 2, 4, 66, 80, m.data_ptr(), n.data_ptr(), 33, 40, 0.5, 0, None, counts.data_ptr(), lens.data_ptr(), 64, None) == -1
    assert b"INTER_AREA" in L.yp_last_error()
    with pytest.raises(ValueError):
        co.resize_linear_f32(masks[0, 0], 33, 40)


@pytest.mark.parametrize("use_crop", [True, False])
def test_loss_mask_assembly_forward(use_crop):
    """loss/loss_yolact.py:239-253 on the device - bit-exact against the C restatement
    (same arithmetic contract), within 1e-5 of the numpy/libm restatement; rows beyond num_pos are zeros."""
    import torch
    from oracle import c_oracle, detect_oracle as po
    from yolact_amd import consumer
    rng = np.random.default_rng(21)
    for (B, rows, ph, pw, H, W) in ((2, 7, 12, 12, 64, 64), (9, 100, 23, 31, 550, 550), (1, 256, 138, 138, 550, 550)):
        proto = np.maximum(0, rng.normal(0, 1, (B, ph, pw, 32))).astype(np.float32)
        coef = np.tanh(rng.normal(0, 1, (B, rows, 32))).astype(np.float32)
        y1 = rng.uniform(-20, H, (B, rows)); x1 = rng.uniform(-20, W, (B, rows))
        boxes = np.stack([y1, x1, y1 + rng.uniform(-5, H / 2, (B, rows)), x1 + rng.uniform(-5, W / 2, (B, rows))], -1).astype(np.float32)
        npos = rng.integers(0, rows + 1, B).astype(np.int32); npos[0] = rows
        got = consumer.assemble_loss_masks(torch.from_numpy(proto).cuda(), torch.from_numpy(coef).cuda(), torch.from_numpy(boxes).cuda(),
                                           torch.from_numpy(npos).cuda(), (H, W), use_cropped_mask=use_crop)
        torch.cuda.synchronize()
        got = got.cpu().numpy()
        for b in range(B):
            n = int(npos[b])
            ref = c_oracle.loss_masks(proto[b], coef[b, :n], boxes[b, :n], H, W, use_crop)
            assert np.array_equal(got[b, :n].view(np.uint32), ref.view(np.uint32)), (B, rows, b)
            assert not got[b, n:].any()
            if b == 0:
                refn = po.loss_mask_assembly(proto[b], coef[b, :n], boxes[b, :n], H, W, use_crop)
                assert np.abs(got[b, :n] - refn).max() <= 1e-5


@pytest.mark.parametrize("use_crop", [True, False])
def test_loss_mask_assembly_backward(use_crop):
    """SURVEY.md 8f F4, backward: gradients w.r.t. proto and pos_coef of loss/loss_yolact.py:239-253 (the
    reference leaves them to TF autodiff).  Bit-exact against the C restatement of the kernels' summation
    order, within 1e-5 (relative to the sum of the terms' magnitudes) of the float64 analytic gradient, rows
    beyond num_pos get zero coefficient gradients, and two runs give identical bits (no float atomics)."""
    import torch
    from oracle import c_oracle, detect_oracle as po
    from yolact_amd import consumer
    rng = np.random.default_rng(22)
    for (B, rows, ph, pw, H, W) in ((2, 7, 12, 12, 64, 64), (5, 100, 23, 31, 550, 550), (1, 40, 138, 138, 550, 550)):
        proto = np.maximum(0, rng.normal(0, 1, (B, ph, pw, 32))).astype(np.float32)
        coef = (0.3 * np.tanh(rng.normal(0, 1, (B, rows, 32)))).astype(np.float32)
        y1 = rng.uniform(-20, H, (B, rows)); x1 = rng.uniform(-20, W, (B, rows))
        boxes = np.stack([y1, x1, y1 + rng.uniform(-5, H / 2, (B, rows)), x1 + rng.uniform(-5, W / 2, (B, rows))], -1).astype(np.float32)
        npos = rng.integers(0, rows + 1, B).astype(np.int32); npos[0] = rows
        grad = rng.normal(0, 1, (B, rows, ph, pw)).astype(np.float32)
        t = [torch.from_numpy(x).cuda() for x in (proto, coef, boxes, npos)]
        masks = consumer.assemble_loss_masks(*t, (H, W), use_cropped_mask=use_crop)
        gp, gc = consumer.assemble_loss_masks_backward(*t, masks, torch.from_numpy(grad).cuda(), (H, W), use_cropped_mask=use_crop)
        gp2, gc2 = consumer.assemble_loss_masks_backward(*t, masks, torch.from_numpy(grad).cuda(), (H, W), use_cropped_mask=use_crop)
        torch.cuda.synchronize()
        assert torch.equal(gp, gp2) and torch.equal(gc, gc2)
        gp, gc, m = gp.cpu().numpy(), gc.cpu().numpy(), masks.cpu().numpy()
        for b in range(B):
            n = int(npos[b])
            rp, rc = c_oracle.loss_masks_backward(proto[b], coef[b, :n], m[b, :n], grad[b, :n])
            assert np.array_equal(gp[b].view(np.uint32), rp.view(np.uint32)), (B, rows, b, np.abs(gp[b] - rp).max())
            assert np.array_equal(gc[b, :n].view(np.uint32), rc.view(np.uint32)), (B, rows, b, np.abs(gc[b, :n] - rc).max())
            assert not gc[b, n:].any()
            if b == 0:
                f8 = [x.astype(np.float64) for x in (proto[b], coef[b, :n], m[b, :n], grad[b, :n])]
                ap, ac = po.loss_mask_assembly_backward(*f8)
                dz = np.abs(f8[3] * f8[2] * (1 - f8[2]))
                sp = np.einsum('nyx,nk->yxk', dz, np.abs(f8[1])) + 1e-30
                sc = np.einsum('nyx,yxk->nk', dz, np.abs(f8[0])) + 1e-30
                assert (np.abs(gp[b] - ap) / sp).max() <= 1e-5 and (np.abs(gc[b, :n] - ac) / sc).max() <= 1e-5


def test_loss_mask_assembly_autograd():
    """LossMaskAssembly: a BCE-style loss on the assembled masks back-propagates to proto and pos_coef; the
    gradients agree with torch autograd through a plain torch restatement of the forward."""
    import torch
    from yolact_amd import consumer
    g = torch.Generator().manual_seed(5)
    B, rows, ph, pw, H, W = 3, 11, 24, 20, 96, 80
    proto = torch.relu(torch.randn(B, ph, pw, 32, generator=g)).cuda().requires_grad_(True)
    coef = (0.3 * torch.tanh(torch.randn(B, rows, 32, generator=g))).cuda().requires_grad_(True)
    y1 = torch.rand(B, rows, generator=g) * H * 0.6; x1 = torch.rand(B, rows, generator=g) * W * 0.6
    boxes = torch.stack([y1, x1, y1 + 8 + torch.rand(B, rows, generator=g) * H * 0.4, x1 + 8 + torch.rand(B, rows, generator=g) * W * 0.4], -1).cuda()
    npos = torch.tensor([rows, 4, 0], dtype=torch.int32).cuda()
    target = (torch.rand(B, rows, ph, pw, generator=g) > 0.5).float().cuda()
    masks = consumer.LossMaskAssembly.apply(proto, coef, boxes, npos, H, W, True)
    loss = ((masks - target) ** 2).sum()
    loss.backward()
    gp, gc = proto.grad.clone(), coef.grad.clone()
    proto.grad = None; coef.grad = None
    # plain torch: sigmoid(proto @ coef^T) times the 0/1 crop the kernel applied (taken from its output)
    inside = (masks.detach() != 0).float()
    ref = torch.sigmoid(torch.einsum('byxk,bnk->bnyx', proto, coef)) * inside
    ((ref - target) ** 2 * inside + (1 - inside) * target ** 2).sum().backward()
    assert torch.allclose(gp, proto.grad, rtol=1e-4, atol=1e-4) and torch.allclose(gc, coef.grad, rtol=1e-4, atol=1e-3)
    assert not gc[2].any() and not gc[1, 4:].any()


def test_sharded_detect_returns_rle_instead_of_float_masks():
    """ShardedDetect(..., rle_hw=...): every shard resizes, thresholds and run-length encodes its masks on its own GPU
    and only the strings travel (two shards on the one GPU here).  Equal to the unsharded consumer chain, image by
    image, with every image at its own size."""
    from yolact_amd import Detect, consumer, synth
    from yolact_amd.sharding import ShardedDetect
    import torch
    cfg = synth.CONFIGS[2]
    net, shp = synth.make_batch(cfg, batch=5)
    dev = {k: torch.from_numpy(v).cuda() for k, v in net.items()}
    sizes = [(480, 640), (550, 550), (333, 500), (640, 427), (600, 600)]
    args = (cfg.num_classes, 300, 100, cfg.conf_thresh, 0.5)
    kw = dict(strict_reference=False, nms_mode="per_class", top_k=200, max_dets=100)
    whole = Detect(*args, **kw)(dev, shp)
    want = consumer.rle_dicts(consumer.masks_to_rle(whole["detection_masks"], whole["num_detections"], sizes))
    sd = ShardedDetect(*args, devices=[0, 0], **kw)
    got = sd(dev, shp, rle_hw=sizes)
    assert "detection_masks" not in got and len(got["detection_rle"]) == 5
    assert got["detection_rle"] == want
    assert torch.equal(got["num_detections"], whole["num_detections"].cpu())
    assert all(t["rle_ms"] is not None and t["images"] in (2, 3) for t in sd.last_timing)
