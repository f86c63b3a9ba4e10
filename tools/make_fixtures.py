#!/usr/bin/env python3
"""Generate tests/golden/* by RUNNING THE REFERENCE ITSELF on CPU (build container only).

    python tools/make_fixtures.py            # writes tests/golden/*.npz, *.json, set5/

The reference (/root/reference) is imported unmodified through tools/_ref_import.py and
driven on seeded inputs; inputs and the reference's outputs / gradients are stored as
data fixtures.  No reference source is copied.  The fixtures pin oracle/sisr_oracle.py
(tests/test_oracle_golden.py) and, transitively, the HIP path (tests/test_*_gpu.py).

Fixture ids follow SURVEY.md §8c: G1 blocks, G2 reduced nets, G3 full-depth init +
forward on the Set5 'baby' tile, G4 train-step trajectories, G6 metadata plumbing,
G7 PSNR / Y conversion.
"""
import hashlib
import json
import os
import shutil
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _ref_import  # noqa: E402

_ref_import.install()

from SISR.models import ModelInterface  # noqa: E402
from SISR.models.advanced import architectures as A  # noqa: E402
from SISR.models.advanced import common as C  # noqa: E402
from SISR.models.advanced.HAN_blocks import CSAM_Module, LAM_Module  # noqa: E402
from SISR.models.attention_manipulators import architectures as Q  # noqa: E402
from SISR.models.attention_manipulators.q_layer import ParaCALayer  # noqa: E402
from sr_tools.image_manipulation import ycbcr_convert  # noqa: E402
from sr_tools.metrics import psnr as ref_psnr  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")
SET5 = "/root/reference/Data/example_data/Set5"
torch.set_num_threads(8)


def _np(t):
    return t.detach().cpu().numpy()


def record(name, module, inputs, call=None, meta=None, seed=8, nonzero=()):
    """Run ``module`` on ``inputs`` (list of tensors; those with requires_grad get input grads),
    backprop a seeded random cotangent, and store everything."""
    g = torch.Generator().manual_seed(seed + 1000)
    for key in nonzero:  # zero-initialised gammas would hide the attention branch entirely
        with torch.no_grad():
            dict(module.named_parameters())[key].fill_(0.37)
    out = call(module, inputs) if call else module(*inputs)
    cot = torch.randn(out.shape, generator=g)
    out.backward(cot)
    blob = {"out": _np(out), "cot": _np(cot)}
    for i, t in enumerate(inputs):
        blob[f"in{i}"] = _np(t)
        if t.requires_grad:  # an input the block ignores (QCALayer 'standard' metadata) gets a zero grad
            blob[f"gin{i}"] = _np(t.grad) if t.grad is not None else np.zeros(t.shape, np.float32)
    for k, v in module.state_dict().items():
        blob["sd/" + k] = _np(v)
    for k, p in module.named_parameters():
        blob["pg/" + k] = _np(p.grad)
    blob["meta"] = np.array(json.dumps(meta or {}))
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **blob)
    print(f"{name:34s} {os.path.getsize(path) / 1e3:8.1f} KB  out{tuple(out.shape)}")


def rnd(*shape, seed, grad=True, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).requires_grad_(grad)


def make_g1():
    relu = torch.nn.ReLU(True)
    B, Cc, H, W = 2, 64, 10, 12
    for tag, (h, w) in {"": (H, W), "_odd": (13, 9)}.items():
        torch.manual_seed(8)
        record("g1_conv64" + tag, C.default_conv(Cc, Cc, 3), [rnd(B, Cc, h, w, seed=1)])
    torch.manual_seed(8)
    record("g1_conv_head", C.default_conv(3, Cc, 3), [rnd(B, 3, H, W, seed=2)])
    torch.manual_seed(8)
    record("g1_conv_tail", C.default_conv(Cc, 3, 3), [rnd(B, Cc, H, W, seed=3)])
    torch.manual_seed(8)
    record("g1_calayer", A.CALayer(Cc, 16), [rnd(B, Cc, H, W, seed=4)])
    torch.manual_seed(8)
    record("g1_rcab", A.RCAB(C.default_conv, Cc, 3, 16, act=relu), [rnd(B, Cc, H, W, seed=5)])
    torch.manual_seed(8)
    record("g1_rcab_odd", A.RCAB(C.default_conv, Cc, 3, 16, act=relu), [rnd(B, Cc, 13, 9, seed=5)])
    torch.manual_seed(8)
    record("g1_resblock", C.ResBlock(C.default_conv, Cc, 3, act=relu, res_scale=0.1), [rnd(B, Cc, H, W, seed=6)],
           meta={"res_scale": 0.1})
    torch.manual_seed(8)
    record("g1_resgroup", A.ResidualGroup(C.default_conv, 32, 3, 16, act=relu, res_scale=1.0, n_resblocks=2),
           [rnd(B, 32, H, W, seed=7)], meta={"n_resblocks": 2})
    torch.manual_seed(8)
    record("g1_upsampler_x4", C.Upsampler(C.default_conv, 4, 16, act=False), [rnd(B, 16, 6, 5, seed=8)],
           meta={"scale": 4})
    torch.manual_seed(8)
    record("g1_upsampler_x3", C.Upsampler(C.default_conv, 3, 16, act=False), [rnd(B, 16, 6, 5, seed=8)],
           meta={"scale": 3})
    # meta-attention (ParaCALayer): M in {1,10,11,20}, nonlinearity in {F,T}
    for M in (1, 10, 11, 20):
        for nl in (False, True):
            torch.manual_seed(8)
            record(f"g1_paraca_m{M}_nl{int(nl)}", ParaCALayer(Cc, M, nonlinearity=nl),
                   [rnd(B, Cc, 5, 6, seed=9), rnd(B, M, 1, 1, seed=10, scale=0.3)],
                   meta={"num_metadata": M, "nonlinearity": nl})
    # QCALayer: all six styles.  'modulate' takes a (B,C,1,1) expanded vector.
    for style in ("standard", "modulate", "mini_concat", "max_concat", "softmax", "extended_attention"):
        M = 10
        attr = rnd(B, Cc if style == "modulate" else M, 1, 1, seed=11, scale=0.3)
        torch.manual_seed(8)
        record(f"g1_qca_{style}", Q.QCALayer(Cc, style, reduction=16, num_metadata=M),
               [rnd(B, Cc, 5, 6, seed=12), attr], meta={"style": style, "num_metadata": M})
    torch.manual_seed(8)
    record("g1_palayer", Q.PALayer(Cc), [rnd(B, Cc, H, W, seed=13)])
    for q in (0, 1):
        for pa in (0, 1):
            torch.manual_seed(8)
            m = Q.QRCAB(C.default_conv, Cc, 3, 16, style="standard", pa=bool(pa), q_layer=bool(q), act=relu,
                        num_metadata=10)
            record(f"g1_qrcab_q{q}_pa{pa}", m, [rnd(B, Cc, H, W, seed=14), rnd(B, 10, 1, 1, seed=15, scale=0.3)],
                   call=lambda mod, i: mod((i[0], i[1]))[0],
                   meta={"style": "standard", "pa": bool(pa), "q_layer": bool(q), "num_metadata": 10})
    for nl in (False, True):
        torch.manual_seed(8)
        m = Q.ParamResBlock(C.default_conv, Cc, 10, 3, act=relu, res_scale=0.1, q_layer_nonlinearity=nl)
        record(f"g1_paramresblock_nl{int(nl)}", m, [rnd(B, Cc, H, W, seed=16), rnd(B, 10, 1, 1, seed=17, scale=0.3)],
               call=lambda mod, i: mod((i[0], i[1]))[0], meta={"res_scale": 0.1, "nonlinearity": nl, "num_metadata": 10})
    torch.manual_seed(8)
    record("g1_lam", LAM_Module(16), [rnd(B, 5, 16, 6, 7, seed=18, scale=0.2)], nonzero=("gamma",))
    torch.manual_seed(8)
    record("g1_csam", CSAM_Module(16), [rnd(B, 16, 6, 7, seed=19)], nonzero=("gamma",))
    torch.manual_seed(8)
    record("g1_csam_c64", CSAM_Module(64), [rnd(1, 64, 9, 7, seed=19)], nonzero=("gamma",))


def make_g2():
    x = lambda s: rnd(2, 3, 12, 10, seed=s, scale=0.5)  # noqa: E731
    md = lambda s: rnd(2, 10, 1, 1, seed=s, scale=0.3)  # noqa: E731
    torch.manual_seed(8)
    cfg = dict(n_resblocks=2, n_resgroups=2, n_feats=16, reduction=16, scale=4)
    record("g2_rcan", A.RCAN(**cfg), [x(20)], meta=cfg)
    torch.manual_seed(8)
    cfg = dict(net_features=16, num_blocks=2, scale=4, res_scale=0.1)
    record("g2_edsr", A.EDSR(**cfg), [x(21)], meta=cfg)
    torch.manual_seed(8)
    cfg = dict(net_features=16, num_blocks=2, scale=3, res_scale=0.1)
    record("g2_edsr_x3", A.EDSR(**cfg), [x(21)], meta=cfg)
    torch.manual_seed(8)
    cfg = dict(n_resgroups=10, n_resblocks=1, n_feats=16, reduction=16, scale=4)
    record("g2_han", A.HAN(**cfg), [rnd(1, 3, 12, 12, seed=22, scale=0.5)], meta=cfg, nonzero=("la.gamma", "csa.gamma"))
    for style in ("standard", "modulate"):
        torch.manual_seed(8)
        # 'modulate' multiplies the CA gate by an n_feats-wide expanded qpi vector, so the 10-d q-layer
        # cannot coexist with it (the reference would fail on the channel count)
        cfg = dict(n_resblocks=2, n_resgroups=2, n_feats=16, reduction=16, scale=4, style=style,
                   num_metadata=(1 if style == "modulate" else 10), include_q_layer=(style != "modulate"))
        attr = rnd(2, 16, 1, 1, seed=23, scale=0.3) if style == "modulate" else md(23)
        record(f"g2_qrcan_{style}", Q.QRCAN(**cfg), [x(24), attr], meta=cfg)
    torch.manual_seed(8)
    cfg = dict(n_resblocks=3, n_resgroups=2, n_feats=16, reduction=16, scale=4, style="standard", num_metadata=10,
               include_q_layer=True, include_pixel_attention=True, selective_meta_blocks=[True, False],
               num_q_layers_inner_residual=2)
    record("g2_qrcan_selective", Q.QRCAN(**cfg), [x(25), md(26)], meta=cfg)
    torch.manual_seed(8)
    cfg = dict(num_features=16, num_blocks=2, scale=4, res_scale=0.1, input_para=10, q_layer_nonlinearity=False)
    record("g2_qedsr", Q.QEDSR(**cfg), [x(27), md(28)], meta=cfg)
    torch.manual_seed(8)
    cfg = dict(n_resgroups=10, n_resblocks=1, n_feats=16, reduction=16, num_metadata=10, scale=4)
    record("g2_qhan", Q.QHAN(**cfg), [rnd(1, 3, 12, 12, seed=29, scale=0.5), rnd(1, 10, 1, 1, seed=30, scale=0.3)],
           meta=cfg, nonzero=("la.gamma", "csa.gamma"))


def sd_digest(sd):
    h = hashlib.sha256()
    for k, v in sd.items():
        h.update(k.encode())
        h.update(np.ascontiguousarray(_np(v)).tobytes())
    return h.hexdigest()


def read_set5():
    import csv
    from PIL import Image
    rows = {}
    with open(os.path.join(SET5, "lr_random_blur", "degradation_metadata.csv")) as f:
        for r in csv.DictReader(f):
            rows[r["image"]] = json.loads(r["blur_kernel"])
    ims = {}
    for name in sorted(rows):
        lr = np.asarray(Image.open(os.path.join(SET5, "lr_random_blur", name)).convert("RGB"))
        hr = np.asarray(Image.open(os.path.join(SET5, "hr", name)).convert("RGB"))
        ims[name] = (lr, hr, rows[name])
    return ims


HANDLER_PARAMS = {
    "edsr": {},
    "rcan": {},
    "han": {},
    "qedsr": {"metadata": ["blur_kernel"]},
    "qrcan": {"metadata": ["blur_kernel"], "style": "standard", "include_q_layer": True},
    "qhan": {"metadata": ["blur_kernel"]},
}


def make_g3():
    """Full-depth seed-8 init through the reference's handlers + forward on every Set5 LR image."""
    ims = read_set5()
    summary = {}
    for name, params in HANDLER_PARAMS.items():
        torch.manual_seed(8)
        model = ModelInterface.define_model(name, device=torch.device("cpu"), model_save_dir="/tmp", eval_mode=True,
                                            scale=4, **params)
        sd = model.net.state_dict()
        entry = {"sha256": sd_digest(sd), "n_tensors": len(sd),
                 "n_params": int(sum(p.numel() for p in model.net.parameters())),
                 "first_keys": list(sd)[:6], "last_keys": list(sd)[-4:], "images": {}}
        crops = {}
        for im_name, (lr, hr, blur) in ims.items():
            x = torch.from_numpy(lr.transpose(2, 0, 1).copy()).float().div(255)[None]
            y = torch.from_numpy(hr.transpose(2, 0, 1).copy()).float().div(255)[None]
            kw = {}
            if "metadata" in params:
                kw = dict(metadata=torch.tensor([blur], dtype=torch.float64),
                          metadata_keys=[("blur_kernel",)] * 10)
            out, loss, _ = model.run_eval(x, y, request_loss=True, **kw)
            o = out.numpy()[0]
            ycb = ycbcr_convert(np.clip(o, 0, 1), im_type="jpg", input="rgb", y_only=False)
            yref = ycbcr_convert(y.numpy()[0], im_type="jpg", input="rgb", y_only=False)
            p = float(ref_psnr(ycb[0], yref[0], max_value=1))
            entry["images"][im_name] = {"mean": float(o.mean()), "std": float(o.std()), "min": float(o.min()),
                                        "max": float(o.max()), "l1": float(loss), "y_psnr": p}
            hh, ww = o.shape[1:]
            crops[im_name] = o[:, hh // 2 - 16:hh // 2 + 16, ww // 2 - 16:ww // 2 + 16].copy()
            print(f"g3 {name:6s} {im_name:14s} psnr={p:.4f} l1={float(loss):.6f}")
        summary[name] = entry
        np.savez_compressed(os.path.join(OUT, f"g3_{name}_crops.npz"), **crops)
    with open(os.path.join(OUT, "g3_full_depth.json"), "w") as f:
        json.dump(summary, f, indent=1)


def make_g4(threads=None, fname="g4_train_steps.json"):
    """run_train trajectories through the reference handlers (full-depth nets, small tiles).
    threads: run the reference at that torch thread count (G4T: how far the reference drifts from ITSELF)."""
    if threads is not None:
        torch.set_num_threads(threads)
    out = {}
    sched = {"scheduler": "cosine_annealing_warm_restarts",
             "scheduler_params": {"t_mult": 1, "restart_period": 3, "lr_min": 1e-7}}
    for name in ("edsr", "qedsr", "rcan", "qrcan"):
        params = dict(HANDLER_PARAMS[name])
        torch.manual_seed(8)
        model = ModelInterface.define_model(name, device=torch.device("cpu"), model_save_dir="/tmp", eval_mode=False,
                                            scale=4, lr=1e-4, grad_clip=(0.5 if name == "qedsr" else None),
                                            **sched, **params)
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
            gn = float(torch.sqrt(sum((p.grad.double() ** 2).sum() for p in model.net.parameters())))
            steps.append({"loss": float(loss), "lr_before": lr_before, "lr_after": model.get_learning_rate(),
                          "grad_norm": gn, "out_mean": float(o.mean()), "out_std": float(o.std())})
            print(f"g4 {name} step {it} loss={float(loss):.6f} gn={gn:.5f} lr->{model.get_learning_rate():.3e}")
        sd = model.net.state_dict()
        out[name] = {"steps": steps, "final_param_sum": float(sum(v.double().sum() for v in sd.values())),
                     "final_param_abs_sum": float(sum(v.double().abs().sum() for v in sd.values())),
                     "grad_clip": (0.5 if name == "qedsr" else None), **sched}
        # checkpoint dict schema (ref: models/__init__.py:349-386)
        st = model.save_model("x", 0, extract_state_only=True)
        out[name]["ckpt_keys"] = sorted(st.keys())
        out[name]["optimizer_group_keys"] = sorted(st["optimizer"]["param_groups"][0].keys())
    if threads is not None:
        for v in out.values():
            v["threads"] = threads
        torch.set_num_threads(8)
    with open(os.path.join(OUT, fname), "w") as f:
        json.dump(out, f, indent=1)


def make_g6():
    from SISR.models.attention_manipulators.handlers import QRCANHandler
    torch.manual_seed(8)
    res = {}
    md = np.random.RandomState(3).rand(3, 12)
    keys = [("qpi",) * 3, ("other",) * 3] + [("blur_kernel",) * 3] * 10
    h = QRCANHandler(device=torch.device("cpu"), model_save_dir="/tmp", eval_mode=True, style="standard",
                     metadata=["blur_kernel"], n_resgroups=1, n_resblocks=1)
    x = torch.zeros(3, 3, 4, 4)
    res["blur_only"] = _np(h.generate_channels(x, torch.from_numpy(md), keys))
    h2 = QRCANHandler(device=torch.device("cpu"), model_save_dir="/tmp", eval_mode=True, style="modulate",
                      metadata=None, n_resgroups=1, n_resblocks=1)
    res["modulate_qpi"] = _np(h2.generate_channels(x, torch.from_numpy(md[:, :1]), [("qpi",) * 3]))
    h3 = QRCANHandler(device=torch.device("cpu"), model_save_dir="/tmp", eval_mode=True, style="standard",
                      metadata=["qpi", "blur_kernel"], n_resgroups=1, n_resblocks=1)
    res["qpi_and_blur"] = _np(h3.generate_channels(x, torch.from_numpy(md), keys))
    h4 = QRCANHandler(device=torch.device("cpu"), model_save_dir="/tmp", eval_mode=True, style="modulate",
                      metadata=None, clamp=True, n_resgroups=1, n_resblocks=1)
    res["modulate_qpi_clamp"] = _np(h4.generate_channels(x, torch.from_numpy(md[:, :1]), [("qpi",) * 3]))
    np.savez_compressed(os.path.join(OUT, "g6_generate_channels.npz"), md=md, **res)
    print("g6", {k: v.shape for k, v in res.items()})


def make_g7():
    from PIL import Image
    ims = read_set5()
    out = {}
    for name, (lr, hr, _) in ims.items():
        up = np.asarray(Image.fromarray(lr).resize((hr.shape[1], hr.shape[0]), resample=Image.BICUBIC))
        a = up.transpose(2, 0, 1).astype(np.float32) / 255
        b = hr.transpose(2, 0, 1).astype(np.float32) / 255
        ya = ycbcr_convert(a, im_type="jpg", input="rgb", y_only=False)
        yb = ycbcr_convert(b, im_type="jpg", input="rgb", y_only=False)
        out[name] = {"y_psnr_bicubic": float(ref_psnr(ya[0], yb[0], max_value=1)),
                     "rgb_psnr_bicubic": float(ref_psnr(a, b, max_value=1)),
                     "y_mean_hr": float(yb[0].mean()), "identical": int(ref_psnr(b, b, max_value=1))}
    with open(os.path.join(OUT, "g7_psnr.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("g7", out)


def copy_set5():
    """Data files only (5 HR + 5 LR PNGs, the metadata CSV).  pca_matrix.pth is not needed."""
    d = os.path.join(OUT, "set5")
    os.makedirs(os.path.join(d, "hr"), exist_ok=True)
    os.makedirs(os.path.join(d, "lr_random_blur"), exist_ok=True)
    for sub in ("hr", "lr_random_blur"):
        for f in sorted(os.listdir(os.path.join(SET5, sub))):
            if f.endswith(".png") or f.endswith(".csv"):
                shutil.copyfile(os.path.join(SET5, sub, f), os.path.join(d, sub, f))


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    which = sys.argv[1:] or ["set5", "g1", "g2", "g3", "g4", "g6", "g7"]
    if "set5" in which:
        copy_set5()
    for tag, fn in (("g1", make_g1), ("g2", make_g2), ("g3", make_g3), ("g4", make_g4), ("g6", make_g6),
                    ("g7", make_g7)):
        if tag in which:
            fn()
    if "g4t" in which:  # the same trajectories from the reference at other thread counts (reference-vs-reference drift)
        make_g4(threads=1, fname="g4_train_steps_t1.json")
        make_g4(threads=3, fname="g4_train_steps_t3.json")


def make_g5():
    """BASELINE config 0: the reference's own train loop (TrainingHandler) on the Set5 example data, CPU, EDSR."""
    import random
    import tempfile
    from collections import defaultdict
    import pandas as pd
    from SISR.training.training_handler import TrainingHandler
    from sr_tools.helper_functions import convert_default_none_dict
    tmp = tempfile.mkdtemp()
    ds = {"name": None, "lr": os.path.join(SET5, "lr_random_blur"), "hr": os.path.join(SET5, "hr"),
          "degradation_metadata": "on_site", "metadata": ["blur_kernel"]}
    out = {}
    for model_name, internal in (("edsr", {"scale": 4, "lr": 1e-4, "num_blocks": 2}),
                                 ("qedsr", {"scale": 4, "lr": 1e-4, "num_blocks": 2, "metadata": ["blur_kernel"]})):
        params = {
            "experiment": "g5_" + model_name, "experiment_save_loc": tmp,
            "data": {"batch_size": 2, "dataloader_threads": 0,
                     "training_sets": {"data_1": dict(ds, crop=32, random_augment=True)},
                     "eval_sets": {"data_1": dict(ds)}},
            "model": {"name": model_name, "internal_params": dict(internal)},
            "training": {"gpu": "off", "seed": 8, "num_epochs": 2, "metrics": ["PSNR"], "logging": "text",
                         "save_samples": False},
        }
        cfg = json.loads(json.dumps(params))
        p = convert_default_none_dict(params)
        exp = TrainingHandler(experiment_name=p["experiment"], save_loc=p["experiment_save_loc"],
                              model_params=p["model"], **p["training"], data_params={**p["data"]})
        exp.run_experiment()
        summ = pd.read_csv(os.path.join(exp.model.logs, "summary.csv"))
        out[model_name] = {"config": cfg, "summary": {k: [float(v) for v in summ[k]] for k in summ.columns}}
        print("g5", model_name, out[model_name]["summary"])
    for m in out.values():  # paths in the stored config are rewritten by the tests
        for part in ("training_sets", "eval_sets"):
            for d in m["config"]["data"][part].values():
                d["lr"], d["hr"] = "SET5/lr_random_blur", "SET5/hr"
        m["config"]["experiment_save_loc"] = "TMP"
    with open(os.path.join(OUT, "g5_train_sisr.json"), "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__" and "g5" in sys.argv[1:]:
    make_g5()
