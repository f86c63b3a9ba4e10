#!/usr/bin/env python3
"""SFTMD golden vectors, produced by RUNNING THE REFERENCE on CPU (build container only).

    python tools/make_fixtures_sftmd.py

f1  reduced net (2 blocks, seed-8 init -- the weights are NOT stored: the builder's module reproduces them from the seed,
    which f2's digest pins) on an odd-sized input with random metadata maps: output, and per parameter the gradient's
    norm and its first 32 values
f2  full-depth (16 blocks) seed-8 init digest + handler.run_eval on two Set5 images with their blur-kernel metadata
f3  five handler.run_train steps
f4  the non-default options on reduced nets (2 blocks, x2): SFT_type 'concat' / 'weak' / 'none' + q_injection, mask_para +
    q_injection, repeats -- output and per-parameter gradient norms (+ leading values)
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_fixtures as MF  # noqa: E402

from SISR.models import ModelInterface  # noqa: E402
from SISR.models.SFTMD_variants.architectures import SFTMD  # noqa: E402
from sr_tools.image_manipulation import ycbcr_convert  # noqa: E402
from sr_tools.metrics import psnr as ref_psnr  # noqa: E402

OUT, _np, rnd = MF.OUT, MF._np, MF.rnd
PARAMS = {"metadata": ["blur_kernel"], "num_blocks": 16, "num_features": 64, "in_nc": 3}


def make_f1():
    torch.manual_seed(8)
    net = SFTMD(in_nc=3, num_features=64, num_blocks=2, scale=4, input_para=10)
    x = rnd(2, 3, 9, 13, seed=91, scale=0.3, grad=False) + 0.5
    md = rnd(2, 10, 1, 1, seed=92, scale=0.3, grad=False).expand(2, 10, 9, 13).contiguous()
    out = net(x, md)
    cot = torch.randn(out.shape, generator=torch.Generator().manual_seed(93))
    out.backward(cot)
    blob = {"in0": _np(x), "in1": _np(md), "out": _np(out), "cot": _np(cot)}
    blob["sd_sha256"] = np.array(MF.sd_digest(net.state_dict()))
    for k, p in net.named_parameters():
        blob["pgn/" + k] = np.array(float(p.grad.double().norm()))
        blob["pg32/" + k] = _np(p.grad.reshape(-1)[:32])
    blob["meta"] = np.array(json.dumps({"num_blocks": 2, "scale": 4, "input_para": 10,
                                        "clamped_fraction": float(((out <= 0) | (out >= 1)).float().mean())}))
    np.savez_compressed(os.path.join(OUT, "f1_sftmd_reduced.npz"), **blob)
    print("f1_sftmd_reduced out", tuple(out.shape), "clamped", float(((out <= 0) | (out >= 1)).float().mean()))


def make_f2():
    ims = MF.read_set5()
    torch.manual_seed(8)
    model = ModelInterface.define_model("sftmd", device=torch.device("cpu"), model_save_dir="/tmp", eval_mode=True,
                                        scale=4, **PARAMS)
    sd = model.net.state_dict()
    entry = {"sha256": MF.sd_digest(sd), "n_tensors": len(sd),
             "n_params": int(sum(p.numel() for p in model.net.parameters())), "keys": list(sd), "images": {}}
    crops = {}
    for im_name in ("butterfly.png", "woman.png"):
        lr, hr, blur = ims[im_name]
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
        print(f"f2 sftmd {im_name:14s} psnr={p:.4f} l1={float(loss):.6f}")
    np.savez_compressed(os.path.join(OUT, "f2_sftmd_crops.npz"), **crops)
    return entry


def make_f3():
    sched = {"scheduler": "cosine_annealing_warm_restarts",
             "scheduler_params": {"t_mult": 1, "restart_period": 3, "lr_min": 1e-7}}
    torch.manual_seed(8)
    model = ModelInterface.define_model("sftmd", device=torch.device("cpu"), model_save_dir="/tmp", eval_mode=False,
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
        print(f"f3 sftmd step {it} loss={float(loss):.6f} gn={gn:.5f}")
    sdv = model.net.state_dict()
    return {"steps": steps, "final_param_sum": float(sum(v.double().sum() for v in sdv.values())), **sched}


VARIANTS = {  # name: (SFTMD kwargs, metadata is per-sample vectors (q_injection) instead of maps)
    "concat": (dict(SFT_type="concat", input_para=10), False),
    "weak1": (dict(SFT_type="weak", input_para=1), False),
    "none_q": (dict(SFT_type="none", q_injection=True, q_layers=2, input_para=10), True),
    "maskpara_q3": (dict(mask_para=True, q_injection=True, q_layers=3, input_para=10), True),
    "repeats3": (dict(repeats=3, input_para=10), False),
    # concat_strategy (handlers.py:12-14): the maps are ALSO concatenated to the RGB input, conv1 is (3 + 10) -> 64
    "concat_input": (dict(input_para=10, in_nc=13), False),
}


def _min_preactivation(net, x, md):
    """smallest |input| any LeakyReLU of the net sees on (x, md): a value within ~1e-6 of the kink takes either slope depending
    on the last bit of a conv's summation order -- inputs are chosen away from it (see make_f4)"""
    lo, hooks = [float("inf")], []
    for m in net.modules():
        if isinstance(m, torch.nn.LeakyReLU):
            hooks.append(m.register_forward_pre_hook(lambda mod, inp: lo.__setitem__(0, min(lo[0], float(inp[0].abs().min())))))
    with torch.no_grad():
        net(x.clone(), md.clone())
    for h in hooks:
        h.remove()
    return lo[0]


def make_f4():
    blob = {}
    for name, (kw, vector) in VARIANTS.items():
        torch.manual_seed(8)
        net = SFTMD(num_features=64, num_blocks=2, scale=2, **{"in_nc": 3, **kw})
        M = kw["input_para"]
        x = rnd(2, 3, 9, 13, seed=91, scale=0.3, grad=False) + 0.5
        if name == "weak1":
            # seed 91 puts one of this variant's 59 904 upscale-stage pre-activations 2.3e-7 from zero (found in round 2: the
            # HIP path and the reference then disagree on one mask bit).  With ~300 k LeakyReLU inputs the closest to zero is
            # typically 1e-6 away; of 100 candidate inputs take the one whose closest is furthest (fp32 summation-order
            # differences at these magnitudes are ~2e-8)
            md0 = (rnd(2, M, 1, 1, seed=92, scale=0.3, grad=False) + 1.0).expand(2, M, 9, 13).contiguous()
            best = max(range(91, 191), key=lambda sd_: _min_preactivation(
                net, rnd(2, 3, 9, 13, seed=sd_, scale=0.3, grad=False) + 0.5, md0))
            x = rnd(2, 3, 9, 13, seed=best, scale=0.3, grad=False) + 0.5
            print("f4 weak1: input seed", best, "closest LeakyReLU input to zero", _min_preactivation(net, x, md0))
        md = rnd(2, M, 1, 1, seed=92, scale=0.3, grad=False) + (1.0 if name == "weak1" else 0.0)
        if not vector:
            md = md.expand(2, M, 9, 13).contiguous()
        if kw.get("in_nc", 3) != 3:
            x = torch.cat((x, md), 1)  # QModel.channel_concat_logic (attention_manipulators/__init__.py:97-98)
        out = net(x, md)
        cot = torch.randn(out.shape, generator=torch.Generator().manual_seed(93))
        out.backward(cot)
        blob[f"{name}/out"], blob[f"{name}/cot"] = _np(out), _np(cot)
        blob[f"{name}/in0"], blob[f"{name}/in1"] = _np(x), _np(md[:, :, :1, :1])  # maps are the vector, expanded
        blob[f"{name}/sd_sha256"] = np.array(MF.sd_digest(net.state_dict()))
        for k, p in net.named_parameters():
            if p.grad is None:
                continue
            blob[f"{name}/pgn/{k}"] = np.array(float(p.grad.double().norm()))
            blob[f"{name}/pg8/{k}"] = _np(p.grad.reshape(-1)[:8])
        print("f4", name, "out", tuple(out.shape), "params with grad", sum(p.grad is not None for p in net.parameters()),
              "clamped", float(((out <= 0) | (out >= 1)).float().mean()))
    np.savez_compressed(os.path.join(OUT, "f4_sftmd_variants.npz"), **blob)


if __name__ == "__main__":
    if "--variants-only" in sys.argv:
        make_f4()
        sys.exit(0)
    make_f1()
    make_f4()
    doc = {"full_depth": make_f2(), "train_steps": make_f3(), "params": PARAMS}
    with open(os.path.join(OUT, "f_sftmd.json"), "w") as f:
        json.dump(doc, f, indent=1)
