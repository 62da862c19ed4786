"""yp_run_static (include/yolact_post.h): the allocation-free variant of yp_run, whose scratch is a
static array inside the library's code object.  Called through the C ABI; its outputs must be
bit-identical to yp_run's on the same inputs - on two calls in a row, the second with a smaller
input (the array keeps the first call's contents)."""
import ctypes

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _run_both(torch, L, net, shp, *, mode, top_k, max_dets, conf, bitmaps=False):
    from yolact_amd import _lib, PostConfig, PostProcessor
    B, A, C = net["pred_cls"].shape
    ph, pw = net["proto_out"].shape[1:3]
    pc = PostConfig(A, C, ph, pw, int(shp[1]), int(shp[2]), conf_thresh=conf, nms_mode=mode, top_k=top_k, max_dets=max_dets,
                    mask_threshold=0.5 if bitmaps else -1.0, max_batch=B)
    dev = {k: torch.from_numpy(np.ascontiguousarray(v)).cuda() for k, v in net.items()}
    pp = PostProcessor(pc)
    ref = pp.run(dev["pred_offset"], dev["pred_cls"], dev["pred_mask_coef"], dev["priors"], dev["proto_out"])
    pp.sync()
    out = pp.alloc_outputs(B, bitmaps=bitmaps)
    for v in out.values():
        v.fill_(3)                                   # whatever the caller had there
    c = _lib.YpConfig(num_anchors=A, num_classes=C, mask_dim=32, proto_h=ph, proto_w=pw, img_h=int(shp[1]), img_w=int(shp[2]),
                      conf_thresh=conf, nms_thresh=0.5, nms_mode={"cc": 0, "per_class": 1}[mode], top_k=top_k, max_dets=max_dets,
                      use_cropped_mask=1, zero_pad_masks=0, mask_threshold=0.5 if bitmaps else -1.0, max_batch=1, device_id=0)
    ptr = lambda t: ctypes.c_void_p(t.data_ptr())
    stream = torch.cuda.current_stream().cuda_stream
    rc = L.yp_run_static(ctypes.byref(c), B, ptr(dev["pred_offset"]), ptr(dev["pred_cls"]), ptr(dev["pred_mask_coef"]),
                         ptr(dev["priors"]), ptr(dev["proto_out"]), ptr(out["detection_boxes"]), ptr(out["detection_classes"]),
                         ptr(out["detection_scores"]), ptr(out["detection_masks"]),
                         ptr(out["detection_bitmaps"]) if bitmaps else None, ptr(out["num_detections"]),
                         ptr(out["detection_anchor_indices"]), ctypes.c_void_p(stream))
    assert rc == 0, L.yp_last_error()
    torch.cuda.synchronize()
    n = ref["num_detections"].cpu().numpy()
    assert np.array_equal(out["num_detections"].cpu().numpy(), n)
    for k in ("detection_boxes", "detection_classes", "detection_scores", "detection_anchor_indices"):
        assert torch.equal(out[k], ref[k]), k
    keys = ["detection_masks"] + (["detection_bitmaps"] if bitmaps else [])
    for k in keys:
        for b in range(B):
            assert torch.equal(out[k][b, :n[b]], ref[k][b, :n[b]]), (k, b)
    pp.close()
    return n


def test_static_scratch_entry_point_matches_yp_run_twice_in_a_row():
    import torch
    from yolact_amd import _lib, synth
    assert torch.cuda.is_available()
    L = _lib.load()
    cfg = synth.CONFIGS[2]
    # call 1: config 2, per-class, 24 images: 77 MB of worst-case candidate lists, more than the 64 MiB
    # array holds, so the call runs as two chunks of 12 images
    net, shp = synth.make_batch(cfg, batch=24)
    n = _run_both(torch, L, net, shp, mode="per_class", top_k=200, max_dets=100, conf=cfg.conf_thresh)
    assert (n == 100).all()
    # call 2: smaller input, other shape and mode, on the array the first call left behind
    toy, tshp = synth.make_toy(3, batch=3)
    _run_both(torch, L, toy, tshp, mode="cc", top_k=15, max_dets=15, conf=0.05, bitmaps=True)
    # call 3: a smaller batch of the first shape again
    sub = {k: (v if k == "priors" else v[:5]) for k, v in net.items()}
    _run_both(torch, L, sub, shp, mode="per_class", top_k=200, max_dets=100, conf=cfg.conf_thresh)


def test_static_scratch_entry_point_rejects_what_does_not_fit():
    from yolact_amd import _lib
    L = _lib.load()
    # 2^23 anchors, per-class, conf 0.01: 8 * 2^23 * 80 bytes of candidate lists for ONE image
    c = _lib.YpConfig(num_anchors=1 << 23, num_classes=81, mask_dim=32, proto_h=8, proto_w=8, img_h=64, img_w=64,
                      conf_thresh=0.01, nms_thresh=0.5, nms_mode=1, top_k=200, max_dets=100, use_cropped_mask=1,
                      zero_pad_masks=0, mask_threshold=-1.0, max_batch=1, device_id=0)
    one = ctypes.c_void_p(16)   # never dereferenced: the size check comes first
    rc = L.yp_run_static(ctypes.byref(c), 1, one, one, one, one, one, one, one, one, one, None, one, None, None)
    assert rc == -1 and b"static scratch holds" in L.yp_last_error()
