#!/usr/bin/env python3
"""Where a hipGraph-replayed training step spends GPU time outside the replay, WITHOUT a profiler attached.

    python tools/step_boundary.py [--workload qrcan] [--batch 4] [--steps 40] [--no-dp]
HIP events recorded on the compute stream around graph.replay() and after optimizer.step() split every step into
  replay    the captured packing + forward + loss + backward,
  tail      end of the replay -> end of the update (gradient exchange join, Adam; GPU idle while the host is late counts here),
  boundary  end of the update -> start of the next replay (input copies; GPU idle while the host prepares the next step).
rocprofv3's kernel trace slows the host enough to open gaps the plain run does not have (profiles/r03_d_step_anatomy_*: 4.6 ms);
this is the measurement that decides whether the step boundary needs work.
"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import bench  # noqa: E402
import sisr_amd  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="qrcan")
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--no-dp", action="store_true")
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    if not a.no_dp:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29541")
        dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=dev)
    name, params, tflop = bench.WORKLOADS[a.workload]
    params = dict(params)
    torch.manual_seed(8)
    h = sisr_amd.available_models[name](device=0, model_save_dir="/tmp", eval_mode=False, scale=4, lr=1e-4,
                                        scheduler="cosine_annealing_warm_restarts",
                                        scheduler_params={"t_mult": 1, "restart_period": 125000, "lr_min": 1e-7}, **params)
    if not a.no_dp:
        h.set_multi_gpu()
    h.use_graph = True
    g = torch.Generator().manual_seed(8)
    B = a.batch
    x, y = torch.rand(B, 3, 128, 128, generator=g).to(dev), torch.rand(B, 3, 512, 512, generator=g).to(dev)
    kw = {"extra_channels": (torch.rand(B, 10, 1, 1, generator=g) * 0.4).to(dev)} if "metadata" in params else {}
    marks = []
    orig_replay, orig_step = torch.cuda.CUDAGraph.replay, h.optimizer.step

    def ev():
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        return e

    def replay(self):
        e0 = ev()
        orig_replay(self)
        marks.append(["replay", e0, ev()])

    def step(*args, **kwargs):
        r = orig_step(*args, **kwargs)
        marks.append(["update", ev()])
        return r

    for _ in range(a.warmup):
        h.train_step(x, y, **kw)
    torch.cuda.synchronize()
    torch.cuda.CUDAGraph.replay, h.optimizer.step = replay, step
    import time
    t0 = time.perf_counter()
    for _ in range(a.steps):
        h.train_step(x, y, **kw)
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / a.steps * 1e3
    torch.cuda.CUDAGraph.replay = orig_replay
    rep, tail, bound = [], [], []
    for i in range(0, len(marks) - 2, 2):
        (_, r0, r1), (_, u), (_, n0, _) = marks[i], marks[i + 1], marks[i + 2]
        rep.append(r0.elapsed_time(r1))
        tail.append(r1.elapsed_time(u))
        bound.append(u.elapsed_time(n0))
    med = lambda v: sorted(v)[len(v) // 2]  # noqa: E731
    print(json.dumps({"workload": a.workload, "batch": B, "dp": not a.no_dp, "ms_per_step_wall": round(wall, 3),
                      "patches_per_s": round(B * 1e3 / wall, 2), "replay_ms": round(med(rep), 3), "tail_ms": round(med(tail), 3),
                      "boundary_ms": round(med(bound), 3), "tail_max": round(max(tail), 3), "boundary_max": round(max(bound), 3)}))
    if h.reducer is not None:
        h.remove_multi_gpu()
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
