#!/usr/bin/env python3
"""SRMD golden vectors, produced by RUNNING THE REFERENCE on CPU (build container only).

    python tools/make_fixtures_srmd.py

m1  reduced net (nc = 64, nb = 4) on an odd-sized 13-channel input: output, every parameter gradient
m2  full-depth (nc = 128, nb = 12) seed-8 init digest + handler.run_eval on the Set5 images with their blur-kernel
    metadata (the handler builds the metadata maps itself: generate_sft_channels + channel concatenation)
m4  reduced nets of the non-default variants: act_mode 'BL' / 'L' / 'BR' / 'R' with upsample_mode 'upconv' / 'pixelshuffle'
    (python tools/make_fixtures_srmd.py m4 regenerates these alone)
m3  five handler.run_train steps (Adam 1e-4, cosine warm restarts every 3 batches)
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
from sr_tools.image_manipulation import ycbcr_convert  # noqa: E402
from sr_tools.metrics import psnr as ref_psnr  # noqa: E402

OUT, _np, rnd = MF.OUT, MF._np, MF.rnd
PARAMS = {"metadata": ["blur_kernel"], "nc": 128, "nb": 12}


def make_m1():
    torch.manual_seed(8)
    net = A.SRMD(in_nc=13, nc=64, nb=4, scale=4)
    x = rnd(2, 13, 9, 21, seed=61, scale=0.5, grad=False)
    out = net(x)
    cot = torch.randn(out.shape, generator=torch.Generator().manual_seed(62))
    out.backward(cot)
    blob = {"in0": _np(x), "out": _np(out), "cot": _np(cot)}
    for k, v in net.state_dict().items():
        blob["sd/" + k] = _np(v)
    for k, p in net.named_parameters():
        blob["pg/" + k] = _np(p.grad)
    blob["meta"] = np.array(json.dumps({"in_nc": 13, "nc": 64, "nb": 4, "scale": 4}))
    np.savez_compressed(os.path.join(OUT, "m1_srmd_reduced.npz"), **blob)
    print("m1_srmd_reduced out", tuple(out.shape))


VARIANTS = {"m4_srmd_BL_upconv": dict(act_mode="BL", upsample_mode="upconv", scale=4),
            "m4_srmd_L": dict(act_mode="L", upsample_mode="pixelshuffle", scale=2),
            "m4_srmd_BR": dict(act_mode="BR", upsample_mode="pixelshuffle", scale=3),
            "m4_srmd_R_upconv": dict(act_mode="R", upsample_mode="upconv", scale=2),
            "m4_srmd_IL_convtranspose": dict(act_mode="IL", upsample_mode="convtranspose", scale=2),
            "m4_srmd_IR_convtranspose": dict(act_mode="IR", upsample_mode="convtranspose", scale=3)}


def make_m4():
    """the non-default act_mode / upsample_mode variants (ref architectures.py:385-411), reduced nets in train() mode: output,
    parameter gradients (norm + 64 leading values), batch-norm running statistics after the forward; seed-8 weights by SHA"""
    for name, cfg in VARIANTS.items():
        torch.manual_seed(8)
        net = A.SRMD(in_nc=13, nc=64, nb=4, **cfg)
        net.train()
        sha = MF.sd_digest({k: v for k, v in net.state_dict().items() if "running_" not in k and "num_batches" not in k})
        x = rnd(2, 13, 10, 12, seed=71, scale=0.5, grad=False)
        out = net(x)
        cot = torch.randn(out.shape, generator=torch.Generator().manual_seed(72))
        out.backward(cot)
        blob = {"in0": _np(x), "out": _np(out), "cot": _np(cot), "sd_sha256": np.array(sha),
                "meta": np.array(json.dumps(dict(in_nc=13, nc=64, nb=4, **cfg)))}
        for k, p in net.named_parameters():
            blob["pgn/" + k] = np.array(float(p.grad.double().norm()))
            blob["pgh/" + k] = _np(p.grad.reshape(-1)[:64])
        for k, v in net.state_dict().items():
            if "running_" in k:
                blob["buf/" + k] = _np(v)
        net.eval()
        with torch.no_grad():
            blob["out_eval"] = _np(net(x))
        path = os.path.join(OUT, name + ".npz")
        np.savez_compressed(path, **blob)
        print(f"{name:22s} {os.path.getsize(path) / 1e3:7.1f} KB out{tuple(out.shape)}")


def make_m2():
    ims = MF.read_set5()
    torch.manual_seed(8)
    model = ModelInterface.define_model("srmd", device=torch.device("cpu"), model_save_dir="/tmp", eval_mode=True,
                                        scale=4, **PARAMS)
    sd = model.net.state_dict()
    entry = {"sha256": MF.sd_digest(sd), "n_tensors": len(sd),
             "n_params": int(sum(p.numel() for p in model.net.parameters())), "keys": list(sd), "images": {}}
    crops = {}
    for im_name, (lr, hr, blur) in ims.items():
        x = torch.from_numpy(lr.transpose(2, 0, 1).copy()).float().div(255)[None]
        y = torch.from_numpy(hr.transpose(2, 0, 1).copy()).float().div(255)[None]
        out, loss, _ = model.run_eval(x, y, request_loss=True, metadata=torch.tensor([blur], dtype=torch.float64),
                                      metadata_keys=[("blur_kernel",)] * 10)
        o = out.numpy()[0]
        ycb = ycbcr_convert(np.clip(o, 0, 1), im_type="jpg", input="rgb", y_only=False)
        yref = ycbcr_convert(y.numpy()[0], im_type="jpg", input="rgb", y_only=False)
        p = float(ref_psnr(ycb[0], yref[0], max_value=1))
        entry["images"][im_name] = {"mean": float(o.mean()), "std": float(o.std()), "l1": float(loss), "y_psnr": p}
        hh, ww = o.shape[1:]
        crops[im_name] = o[:, hh // 2 - 16:hh // 2 + 16, ww // 2 - 16:ww // 2 + 16].copy()
        print(f"m2 srmd {im_name:14s} psnr={p:.4f} l1={float(loss):.6f}")
    np.savez_compressed(os.path.join(OUT, "m2_srmd_crops.npz"), **crops)
    return entry


def make_m3():
    sched = {"scheduler": "cosine_annealing_warm_restarts",
             "scheduler_params": {"t_mult": 1, "restart_period": 3, "lr_min": 1e-7}}
    torch.manual_seed(8)
    model = ModelInterface.define_model("srmd", device=torch.device("cpu"), model_save_dir="/tmp", eval_mode=False,
                                        scale=4, lr=1e-4, **sched, **PARAMS)
    g = torch.Generator().manual_seed(77)
    steps = []
    for it in range(5):
        x = torch.rand(2, 3, 16, 16, generator=g)
        y = torch.rand(2, 3, 64, 64, generator=g)
        md = torch.rand(2, 10, generator=g, dtype=torch.float64) * 0.4
        lr_before = model.get_learning_rate()
        loss, o = model.run_train(x, y, metadata=md, metadata_keys=[("blur_kernel", "blur_kernel")] * 10)
        gn = float(torch.sqrt(sum((p.grad.double() ** 2).sum() for p in model.net.parameters())))
        steps.append({"loss": float(loss), "lr_before": lr_before, "lr_after": model.get_learning_rate(),
                      "grad_norm": gn, "out_mean": float(o.mean()), "out_std": float(o.std())})
        print(f"m3 srmd step {it} loss={float(loss):.6f} gn={gn:.5f}")
    sdv = model.net.state_dict()
    return {"steps": steps, "final_param_sum": float(sum(v.double().sum() for v in sdv.values())), **sched}


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "m4":
        make_m4()
        sys.exit(0)
    make_m1()
    make_m4()
    doc = {"full_depth": make_m2(), "train_steps": make_m3(), "params": PARAMS}
    with open(os.path.join(OUT, "m_srmd.json"), "w") as f:
        json.dump(doc, f, indent=1)
