"""Synthetic head outputs for the five BASELINE.json configs.

There is no network for datasets or checkpoints, and the reference's model
cannot be built offline (``weights='imagenet'``, yolact.py:33), so Detect is fed
synthetic tensors whose value ranges mirror the real heads: raw class logits
(layers/head.py:66,80), tanh mask coefficients (layers/head.py:81), ReLU
prototypes (layers/protonet.py:28-30).  Everything is float32 and a pure
function of ``(config, image_index)``: seed = 1000*config + image_index.
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

from .anchors import make_priors, proto_size

MASK_DIM = 32


@dataclass(frozen=True)
class SynthConfig:
    config_id: int
    img_h: int
    img_w: int
    batch: int
    num_classes: int = 81        # incl. background
    conf_thresh: float = 0.05
    nms_thresh: float = 0.5
    top_k: int = 200
    max_dets: int = 100
    p_fg: float = 0.03
    mode: str = "planted"        # "planted" | "planted_soft" | "random_init" | "saturated"

    @property
    def proto_hw(self):
        return proto_size(self.img_h, self.img_w)


# BASELINE.json "configs", in order.
CONFIGS = {
    1: SynthConfig(1, 550, 550, 1, top_k=15, max_dets=15),                  # reference-live settings
    2: SynthConfig(2, 550, 550, 32),
    3: SynthConfig(3, 768, 1024, 64, p_fg=0.02),
    4: SynthConfig(4, 550, 550, 32, conf_thresh=0.01, mode="saturated"),
    5: SynthConfig(5, 800, 1333, 64, p_fg=0.02),
}


def make_image(cfg: SynthConfig, image_index: int, priors: np.ndarray | None = None, mode: str | None = None):
    """One image's head outputs: (pred_offset[A,4], pred_cls[A,C], pred_mask_coef[A,32], proto[ph,pw,32])."""
    mode = mode or cfg.mode
    A = (priors.shape[0] if priors is not None else make_priors(cfg.img_h, cfg.img_w).shape[0])
    C = cfg.num_classes
    ph, pw = cfg.proto_hw
    rng = np.random.default_rng(1000 * cfg.config_id + image_index)
    off = np.empty((A, 4), np.float32)
    off[:, :2] = rng.normal(0.0, 0.5, (A, 2))
    off[:, 2:] = rng.normal(0.0, 0.7, (A, 2))
    if mode == "random_init":
        cls = rng.normal(0.0, 0.03, (A, C)).astype(np.float32)            # max fg prob ~ 1/81 < conf -> S = 0
    else:
        cls = rng.normal(0.0, 1.0, (A, C)).astype(np.float32)
        cls[:, 0] += np.float32(6.0)
        if mode == "saturated":
            fg = np.arange(A) % (C - 1)
            boost = rng.uniform(8.0, 14.0, A).astype(np.float32)
            cls[np.arange(A), 1 + fg] += boost
        else:
            planted = rng.random(A) < cfg.p_fg
            idx = np.nonzero(planted)[0]
            fg = rng.integers(0, C - 1, idx.shape[0])
            # "planted_soft": foreground probabilities spread over ~0.2-0.9 instead of saturating at 0.999,
            # so that scores an ulp apart - which any two exp implementations rank differently - are rare
            lo, hi = (4.5, 8.5) if mode == "planted_soft" else (6.0, 14.0)
            boost = rng.uniform(lo, hi, idx.shape[0]).astype(np.float32)
            cls[idx, 1 + fg] += boost
    coef = np.tanh(rng.normal(0.0, 1.0, (A, MASK_DIM))).astype(np.float32)
    proto = np.maximum(0.0, rng.normal(0.0, 1.0, (ph, pw, MASK_DIM))).astype(np.float32)
    return off, cls, coef, proto


def make_batch(cfg: SynthConfig, batch: int | None = None, first_image: int = 0, mode: str | None = None):
    """net_outs dict shaped like Yolact.call's `pred` (yolact.py:183-192) plus img_shape [B,H,W,3]."""
    B = cfg.batch if batch is None else batch
    priors = make_priors(cfg.img_h, cfg.img_w)
    A = priors.shape[0]
    ph, pw = cfg.proto_hw
    out = {
        "pred_offset": np.empty((B, A, 4), np.float32),
        "pred_cls": np.empty((B, A, cfg.num_classes), np.float32),
        "pred_mask_coef": np.empty((B, A, MASK_DIM), np.float32),
        "proto_out": np.empty((B, ph, pw, MASK_DIM), np.float32),
        "priors": priors,
    }
    for i in range(B):
        o, c, m, p = make_image(cfg, first_image + i, priors, mode)
        out["pred_offset"][i], out["pred_cls"][i], out["pred_mask_coef"][i], out["proto_out"][i] = o, c, m, p
    img_shape = np.array([B, cfg.img_h, cfg.img_w, 3], np.int32)
    return out, img_shape


def make_toy(seed: int, batch=2, fmaps=((8, 8), (4, 4)), img=(64, 64), num_classes=5, proto_hw=(12, 12),
             p_fg=0.1, scales=(12, 24)):
    """Small pyramid for fixtures and fast parity tests (A = 3*(64+16) = 240 by default)."""
    rng = np.random.default_rng(seed)
    priors = make_priors(img[0], img[1], feature_maps=fmaps, scales=scales)
    A = priors.shape[0]
    C = num_classes
    off = np.empty((batch, A, 4), np.float32)
    off[..., :2] = rng.normal(0.0, 0.5, (batch, A, 2))
    off[..., 2:] = rng.normal(0.0, 0.7, (batch, A, 2))
    cls = rng.normal(0.0, 1.0, (batch, A, C)).astype(np.float32)
    cls[..., 0] += np.float32(4.0)
    planted = rng.random((batch, A)) < p_fg
    bi, ai = np.nonzero(planted)
    fg = rng.integers(0, C - 1, bi.shape[0])
    cls[bi, ai, 1 + fg] += rng.uniform(3.0, 9.0, bi.shape[0]).astype(np.float32)
    coef = np.tanh(rng.normal(0.0, 1.0, (batch, A, MASK_DIM))).astype(np.float32)
    proto = np.maximum(0.0, rng.normal(0.0, 1.0, (batch, proto_hw[0], proto_hw[1], MASK_DIM))).astype(np.float32)
    net = {"pred_offset": off, "pred_cls": cls, "pred_mask_coef": coef, "proto_out": proto, "priors": priors}
    return net, np.array([batch, img[0], img[1], 3], np.int32)
