#!/usr/bin/env python3
"""SAN / QSAN golden vectors, produced by RUNNING THE REFERENCE on CPU (build container only).

    python tools/make_fixtures_san.py [s1] [s2] [s3] [s4]

s1  blocks: SOCA (covariance pooling + Newton-Schulz square root + gate), Nonlocal_CA (quadrant-wise embedded-
    Gaussian attention), RB, LSRAG, QRB, QLSRAG -- output, input grads, every parameter grad
s2  reduced-depth SAN / QSAN whole nets
s3  full-depth seed-8 init digests + handler.run_eval (forward_chop tiling) on the Set5 images
s4  five run_train steps through the SAN / QSAN handlers

Zero-initialised tensors (the non-local output projection W, the share-source gamma) are given seeded random
values first -- at their init value they would hide the attention branch from the vectors.  Parameters the
reference constructs but never uses in forward (SAN.conv_last, Nonlocal_CA.soca, LSRAG.gamma) get no gradient;
they are listed under meta['unused'] and stored as zeros.
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_fixtures as MF  # noqa: E402  (installs the import shim, imports the reference)

from SISR.models import ModelInterface  # noqa: E402
from SISR.models.advanced import architectures as A  # noqa: E402
from SISR.models.advanced import common as C  # noqa: E402
from SISR.models.advanced import SAN_blocks as S  # noqa: E402
from SISR.models.attention_manipulators import architectures as Q  # noqa: E402
from SISR.models.attention_manipulators import qsan_blocks as QS  # noqa: E402
from sr_tools.image_manipulation import ycbcr_convert  # noqa: E402
from sr_tools.metrics import psnr as ref_psnr  # noqa: E402

OUT, _np, rnd = MF.OUT, MF._np, MF.rnd


def randomize(module, keys, seed=4242, scale=0.2):
    g = torch.Generator().manual_seed(seed)
    named = dict(module.named_parameters())
    with torch.no_grad():
        for k in keys:
            named[k].copy_(torch.randn(named[k].shape, generator=g) * scale)


def record(name, module, inputs, call=None, meta=None, seed=8):
    g = torch.Generator().manual_seed(seed + 1000)
    out = call(module, inputs) if call else module(*inputs)
    cot = torch.randn(out.shape, generator=g)
    out.backward(cot)
    blob = {"out": _np(out), "cot": _np(cot)}
    for i, t in enumerate(inputs):
        blob[f"in{i}"] = _np(t)
        if t.requires_grad:
            blob[f"gin{i}"] = _np(t.grad) if t.grad is not None else np.zeros(t.shape, np.float32)
    for k, v in module.state_dict().items():
        blob["sd/" + k] = _np(v)
    unused = []
    for k, p in module.named_parameters():
        if p.grad is None:
            unused.append(k)
            blob["pg/" + k] = np.zeros(tuple(p.shape), np.float32)
        else:
            blob["pg/" + k] = _np(p.grad)
    blob["meta"] = np.array(json.dumps(dict(meta or {}, unused=unused)))
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **blob)
    print(f"{name:34s} {os.path.getsize(path) / 1e3:8.1f} KB  out{tuple(out.shape)}  unused={len(unused)}")


NL_ZERO = ("non_local.W.weight", "non_local.W.bias")
NET_ZERO = ("non_local.non_local.W.weight", "non_local.non_local.W.bias", "gamma")


def make_s1():
    relu = torch.nn.ReLU(True)
    B, Cc = 2, 64
    for tag, (h, w) in {"": (10, 12), "_odd": (13, 9)}.items():
        torch.manual_seed(8)
        record("s1_soca" + tag, S.SOCA(Cc, reduction=16), [rnd(B, Cc, h, w, seed=41)], meta={"reduction": 16})
        torch.manual_seed(8)
        m = S.Nonlocal_CA(in_feat=Cc, inter_feat=Cc // 8, reduction=8, sub_sample=False, bn_layer=False)
        randomize(m, NL_ZERO)
        record("s1_nonlocal" + tag, m, [rnd(B, Cc, h, w, seed=42)], meta={"inter_feat": Cc // 8})
    # the covariance pooling / square-root pair on its own (custom autograd Functions in the reference)
    from SISR.models.advanced import mpncov

    class CovSqrt(torch.nn.Module):
        def forward(self, x):
            return mpncov.SqrtmLayer(mpncov.CovpoolLayer(x), 5)

    record("s1_covsqrt", CovSqrt(), [rnd(B, 16, 7, 9, seed=43)], meta={"iterN": 5})
    Cc = 32  # composition vectors: half width keeps the files small
    torch.manual_seed(8)
    record("s1_rb", S.RB(C.default_conv, Cc, 3, 16, act=relu), [rnd(B, Cc, 10, 12, seed=44)])
    torch.manual_seed(8)
    record("s1_lsrag", S.LSRAG(C.default_conv, Cc, 3, 16, act=relu, res_scale=1, n_resblocks=2),
           [rnd(B, Cc, 10, 12, seed=45, scale=0.5)], meta={"n_resblocks": 2, "reduction": 16})
    torch.manual_seed(8)
    record("s1_qrb", QS.QRB(C.default_conv, Cc, 3, 16, act=relu, num_metadata=10),
           [rnd(B, Cc, 10, 12, seed=46), rnd(B, 10, 1, 1, seed=47, scale=0.3)],
           call=lambda mod, i: mod((i[0], i[1])), meta={"num_metadata": 10})
    torch.manual_seed(8)
    record("s1_qlsrag", QS.QLSRAG(C.default_conv, Cc, 3, 16, act=relu, res_scale=1, n_resblocks=2, num_metadata=10),
           [rnd(B, Cc, 10, 12, seed=48, scale=0.5), rnd(B, 10, 1, 1, seed=49, scale=0.3)],
           call=lambda mod, i: mod((i[0], i[1]))[0], meta={"n_resblocks": 2, "reduction": 16, "num_metadata": 10})


def make_s2():
    x = lambda s: rnd(2, 3, 12, 10, seed=s, scale=0.5)  # noqa: E731
    cfg = dict(n_resgroups=2, n_resblocks=2, n_feats=16, reduction=4, scale=4)
    torch.manual_seed(8)
    m = A.SAN(**cfg)
    randomize(m, NET_ZERO)
    record("s2_san", m, [x(50)], meta=cfg)
    torch.manual_seed(8)
    m = Q.QSAN(input_para=10, **cfg)
    randomize(m, NET_ZERO)
    record("s2_qsan", m, [x(51), rnd(2, 10, 1, 1, seed=52, scale=0.3)], meta=dict(cfg, input_para=10))


PARAMS = {"san": {}, "qsan": {"metadata": ["blur_kernel"]}}


def make_s3():
    ims = MF.read_set5()
    summary = {}
    for name, params in PARAMS.items():
        torch.manual_seed(8)
        model = ModelInterface.define_model(name, device=torch.device("cpu"), model_save_dir="/tmp", eval_mode=True,
                                            scale=4, **params)
        sd = model.net.state_dict()
        entry = {"sha256": MF.sd_digest(sd), "n_tensors": len(sd),
                 "n_params": int(sum(p.numel() for p in model.net.parameters())),
                 "first_keys": list(sd)[:8], "last_keys": list(sd)[-4:], "images": {}}
        # untouched init leaves W = 0 and gamma = 0 (attention branches silent): evaluate a perturbed copy as well
        randomize(model.net, NET_ZERO, seed=777, scale=0.05)
        entry["perturbed"] = {"keys": list(NET_ZERO), "seed": 777, "scale": 0.05,
                              "sha256": MF.sd_digest(model.net.state_dict())}
        crops = {}
        for im_name, (lr, hr, blur) in ims.items():
            x = torch.from_numpy(lr.transpose(2, 0, 1).copy()).float().div(255)[None]
            y = torch.from_numpy(hr.transpose(2, 0, 1).copy()).float().div(255)[None]
            kw = {}
            if "metadata" in params:
                kw = dict(metadata=torch.tensor([blur], dtype=torch.float64), metadata_keys=[("blur_kernel",)] * 10)
            out, loss, _ = model.run_eval(x, y, request_loss=True, **kw)  # forward_chop: 4 overlapping chunks
            o = out.numpy()[0]
            ycb = ycbcr_convert(np.clip(o, 0, 1), im_type="jpg", input="rgb", y_only=False)
            yref = ycbcr_convert(y.numpy()[0], im_type="jpg", input="rgb", y_only=False)
            p = float(ref_psnr(ycb[0], yref[0], max_value=1))
            entry["images"][im_name] = {"mean": float(o.mean()), "std": float(o.std()), "min": float(o.min()),
                                        "max": float(o.max()), "l1": float(loss), "y_psnr": p}
            hh, ww = o.shape[1:]
            crops[im_name] = o[:, hh // 2 - 16:hh // 2 + 16, ww // 2 - 16:ww // 2 + 16].copy()
            print(f"s3 {name:6s} {im_name:14s} psnr={p:.4f} l1={float(loss):.6f}", flush=True)
        summary[name] = entry
        np.savez_compressed(os.path.join(OUT, f"s3_{name}_crops.npz"), **crops)
    with open(os.path.join(OUT, "s3_full_depth.json"), "w") as f:
        json.dump(summary, f, indent=1)


def make_s4():
    out = {}
    sched = {"scheduler": "cosine_annealing_warm_restarts",
             "scheduler_params": {"t_mult": 1, "restart_period": 3, "lr_min": 1e-7}}
    for name, params in PARAMS.items():
        torch.manual_seed(8)
        model = ModelInterface.define_model(name, device=torch.device("cpu"), model_save_dir="/tmp", eval_mode=False,
                                            scale=4, lr=1e-4, **sched, **params)
        g = torch.Generator().manual_seed(77)
        steps = []
        for it in range(5):
            x = torch.rand(2, 3, 16, 16, generator=g)
            y = torch.rand(2, 3, 64, 64, generator=g)
            md = torch.rand(2, 10, generator=g, dtype=torch.float64) * 0.4
            kw = {}
            if "metadata" in params:
                kw = dict(metadata=md, metadata_keys=[("blur_kernel", "blur_kernel")] * 10)
            lr_before = model.get_learning_rate()
            loss, o = model.run_train(x, y, **kw)
            gn = float(torch.sqrt(sum((p.grad.double() ** 2).sum() for p in model.net.parameters()
                                      if p.grad is not None)))
            steps.append({"loss": float(loss), "lr_before": lr_before, "lr_after": model.get_learning_rate(),
                          "grad_norm": gn, "out_mean": float(o.mean()), "out_std": float(o.std())})
            print(f"s4 {name} step {it} loss={float(loss):.6f} gn={gn:.5f}", flush=True)
        sd = model.net.state_dict()
        out[name] = {"steps": steps, "final_param_sum": float(sum(v.double().sum() for v in sd.values())),
                     "final_param_abs_sum": float(sum(v.double().abs().sum() for v in sd.values())), **sched}
    with open(os.path.join(OUT, "s4_train_steps.json"), "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    which = sys.argv[1:] or ["s1", "s2", "s3", "s4"]
    for tag, fn in (("s1", make_s1), ("s2", make_s2), ("s3", make_s3), ("s4", make_s4)):
        if tag in which:
            fn()
