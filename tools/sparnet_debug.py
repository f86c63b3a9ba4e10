"""Debug aid: block-by-block activations and gradients of the reduced (Q)SPARNet, HIP path vs the float64 oracle."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch, numpy as np
import test_sparnet as T
from oracle import sisr_oracle as O
import sisr_amd
name = sys.argv[1] if len(sys.argv) > 1 else "p3_qsparnet_reduced"
a, meta, m = T._build(name)
q = name.startswith("p3")
m.to("cuda:0").train()
x = torch.from_numpy(a["in0"]).to("cuda:0")
md = torch.from_numpy(a["md"]).to("cuda:0") if q else None
# HIP: run block by block keeping outputs
acts = {}
def run_hip():
    t = sisr_amd.sparnet._rgb_in(x)
    seqs = [("encoder", m.encoder), ("res_layers", m.res_layers), ("decoder", m.decoder)]
    for sname, seq in seqs:
        for i, blk in enumerate(seq):
            r = blk((t, md)) if q else blk(t)
            t = r[0] if q else r
            t.retain_grad()
            acts[f"{sname}.{i}"] = t
    y = m.out_conv(t)
    return sisr_amd.ops.shuffle_rgb(y, 3, 1)
out = run_hip()
out.backward(torch.from_numpy(a["cot"]).to("cuda:0"))
# oracle float64 block by block
def run_oracle(dt):
    sd = {}
    for k, v in m.state_dict().items():
        v = v.detach().cpu().clone()
        v = v.to(dt) if v.is_floating_point() else v
        sd[k] = v
    cfg = {k: meta[k] for k in ("in_size", "out_size", "min_feat_size", "res_depth", "bottleneck_size")}
    down = int(np.log2(cfg["in_size"] // cfg["min_feat_size"])); up = int(np.log2(cfg["out_size"] // cfg["min_feat_size"]))
    hg = int(np.log2(64 / cfg["bottleneck_size"]))
    xx = torch.from_numpy(a["in0"]).to(dt); mm = torch.from_numpy(a["md"]).to(dt) if q else None
    res = {}
    t = O._sp_conv_layer(sd, "encoder.0", xx); t.requires_grad_(True); t.retain_grad(); res["encoder.0"] = t
    for i in range(down):
        t = O._sp_block(sd, f"encoder.{i+1}", t, "down", hg, "bn", "leakyrelu", True, mm); t.retain_grad(); res[f"encoder.{i+1}"] = t; hg -= 1
    hg += 1
    for i in range(cfg["res_depth"] + 3 - down):
        t = O._sp_block(sd, f"res_layers.{i}", t, "none", hg, "bn", "leakyrelu", True, mm); t.retain_grad(); res[f"res_layers.{i}"] = t
    for i in range(up):
        hg += 1
        t = O._sp_block(sd, f"decoder.{i}", t, "up", hg, "bn", "leakyrelu", True, mm); t.retain_grad(); res[f"decoder.{i}"] = t
    o = O._sp_conv_layer(sd, "out_conv", t)
    o.backward(torch.from_numpy(a["cot"]).to(dt))
    return res
r64, r32 = run_oracle(torch.float64), run_oracle(torch.float32)
def rel(p, r):
    return float((p.double() - r.double()).norm() / (r.double().norm() + 1e-30))
for k, t in acts.items():
    C = r64[k].shape[1]
    print("%-14s act hip %.2e ref32 %.2e | grad hip %.2e ref32 %.2e | shape %s" % (
        k, rel(t.detach().cpu()[:, :C], r64[k].detach()), rel(r32[k].detach(), r64[k].detach()),
        rel(t.grad.cpu()[:, :C], r64[k].grad), rel(r32[k].grad, r64[k].grad), tuple(r64[k].shape)))
