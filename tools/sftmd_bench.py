"""SFTMD x4 train step on one GPU: patches/s + where the time goes (HIP-event timing per kernel family, one extra eager step).

python tools/sftmd_bench.py [--batch 16] [--lr-size 64] [--steps 8]
Prints one JSON line: value (LR patches/s), ms_per_step, per-family avg launch time and, for the 3x3 MFMA convs, achieved
fp32 TFLOP/s on EXECUTED flops (merged 128 -> 64 and block-diagonal 64 -> 128 SFT convs count as executed, i.e. with their
structural zeros) beside the algorithmic figure of the reference's four convs (ref: SFTMD_variants/architectures.py:25-56)."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--lr-size", type=int, default=64)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=3)
    args = ap.parse_args()
    import sisr_amd as sisr
    from sisr_amd import hip, ops
    B, S = args.batch, args.lr_size
    torch.manual_seed(8)
    h = sisr.available_models["sftmd"](device=0, model_save_dir="/tmp", eval_mode=False, scale=4, lr=1e-4,
                                       metadata=["blur_kernel"], num_blocks=16, num_features=64, in_nc=3)
    g = torch.Generator().manual_seed(8)
    x = torch.rand(B, 3, S, S, generator=g).cuda()
    y = torch.rand(B, 3, 4 * S, 4 * S, generator=g).cuda()
    maps = (torch.rand(B, 10, 1, 1, generator=g) * 0.4).expand(B, 10, S, S).contiguous().cuda()
    for _ in range(args.warmup):
        h.train_step(x, y, extra_channels=maps)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss, _ = h.train_step(x, y, extra_channels=maps)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps

    # one more step with every library call timed alone
    events = {}
    L = hip.lib()
    orig_conv, orig_wgrad = ops.conv_c64, ops.wgrad_c64

    def timed(fam, flop, fn):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        events.setdefault(fam, []).append((e0, e1, flop))

    def conv(x_, xv, pk, b_, bnq, y_, yv, B_, H, W, cin, cout, **kw):
        fam = f"conv3x3 {cin}->{cout}" + (" leaky" if kw.get("relu") == ops.LEAKY else "") + \
            (" mask" if kw.get("mask") is not None else "") + (" +res" if kw.get("res") is not None else "") + \
            (" (bwd)" if ops.IN_BACKWARD else "")
        timed(fam, 2.0 * B_ * H * W * cin * cout * 9, lambda: orig_conv(x_, xv, pk, b_, bnq, y_, yv, B_, H, W, cin, cout, **kw))

    def wgrad(x_, xv, dy, dyv, dw, db, B_, H, W, cin, cout, **kw):
        timed(f"wgrad3x3 {cin}->{cout}", 2.0 * B_ * H * W * cin * cout * 9,
              lambda: orig_wgrad(x_, xv, dy, dyv, dw, db, B_, H, W, cin, cout, **kw))

    class Wrap:
        def __init__(self, inner):
            self._inner = inner

        def __getattr__(self, name):
            fn = getattr(self._inner, name)
            if name in ("sisr_conv9_fwd", "sisr_conv9_dgrad", "sisr_conv9_wgrad", "sisr_sft_combine_fwd", "sisr_sft_combine_bwd",
                        "sisr_map64", "sisr_sft_compose", "sisr_clamp01", "sisr_l1_loss", "sisr_adam_flat",
                        "sisr_conv3x3_cin3", "sisr_corr3x3_c3", "sisr_pack_conv3x3_both", "sisr_nchw_to_nhwc_pad"):
                def call(*a):
                    out = []
                    timed(name, 0.0, lambda: out.append(fn(*a)))
                    return out[0]
                return call
            return fn

    ops.conv_c64, ops.wgrad_c64 = conv, wgrad
    real_lib = hip.lib
    wrapped = Wrap(L)
    hip.lib = lambda: wrapped
    side, ops.WGRAD_SIDE_STREAM = ops.WGRAD_SIDE_STREAM, False
    try:
        h.train_step(x, y, extra_channels=maps)
        torch.cuda.synchronize()
    finally:
        ops.conv_c64, ops.wgrad_c64, hip.lib, ops.WGRAD_SIDE_STREAM = orig_conv, orig_wgrad, real_lib, side
    fams = []
    for fam, evs in events.items():
        ms = [a.elapsed_time(b) for a, b, _ in evs]
        flop = sum(f for _, _, f in evs)
        fams.append({"family": fam, "launches": len(ms), "avg_us": round(1e3 * sum(ms) / len(ms), 1), "ms_per_step": round(sum(ms), 2),
                     "executed_tflops": round(flop / (sum(ms) * 1e-3) / 1e12, 1) if flop else None})
    fams.sort(key=lambda d: -d["ms_per_step"])
    print(json.dumps({"metric": "sftmd_train_lr_patches_per_s", "value": B / dt, "ms_per_step": 1e3 * dt, "batch": B, "lr_size": S,
                      "loss": float(loss), "timed_sum_ms": round(sum(f["ms_per_step"] for f in fams), 1), "families": fams}))


if __name__ == "__main__":
    main()
