"""BASELINE config 0 through our own train entry point vs the reference's TrainingHandler (fixture G5).

CPU variant: the handler's parameter holders are driven by the ORACLE forward (test-only monkeypatch), which
isolates everything around the kernels -- TOML plumbing, dataset ordering, Python/torch RNG call order,
collation, Adam/scheduler, Y-PSNR validation, summary.csv, checkpoints -- and pins it to the reference run.
GPU variant (pytest -m gpu): the same experiment on the HIP kernels.
"""
import copy
import os

import numpy as np
import pytest
import torch

import sisr_amd
from conftest import GOLDEN, golden_json
from oracle import sisr_oracle as O


def _config(name, tmp_path):
    cfg = copy.deepcopy(golden_json("g5_train_sisr")[name]["config"])
    cfg["experiment_save_loc"] = str(tmp_path)
    for part in ("training_sets", "eval_sets"):
        for d in cfg["data"][part].values():
            d["lr"] = d["lr"].replace("SET5", os.path.join(GOLDEN, "set5"))
            d["hr"] = d["hr"].replace("SET5", os.path.join(GOLDEN, "set5"))
    return cfg


def _oracle_drive(handler, name):
    net = handler.net
    cfg = dict(num_blocks=2, scale=4, res_scale=0.1)
    if name == "qedsr":
        cfg["q_layer_nonlinearity"] = False
        net.forward = lambda x, metadata=None: O.qedsr(dict(net.state_dict(keep_vars=True)), x, metadata, **cfg)
    else:
        net.forward = lambda x: O.edsr(dict(net.state_dict(keep_vars=True)), x, **cfg)
    handler.criterion = torch.nn.L1Loss()


@pytest.mark.parametrize("name", ["edsr", "qedsr"])
def test_train_loop_matches_reference_with_oracle_net(name, tmp_path, monkeypatch):
    ref = golden_json("g5_train_sisr")[name]["summary"]
    cfg = _config(name, tmp_path)
    real_init = sisr_amd.cli.ModelInterface.__init__

    def patched(self, *a, **k):
        real_init(self, *a, **k)
        _oracle_drive(self.model, name)
    monkeypatch.setattr(sisr_amd.cli.ModelInterface, "__init__", patched)
    total = sisr_amd.cli.train_sisr(cfg)
    for key in ("train-loss", "val-loss", "val-PSNR", "learning-rate"):
        np.testing.assert_allclose(total[key], ref[key], rtol=2e-5, atol=2e-6, err_msg=key)
    assert list(total["epoch"]) == [0, 1]
    exp = os.path.join(str(tmp_path), cfg["experiment"])
    assert os.path.isfile(os.path.join(exp, "config.toml"))
    assert os.path.isfile(os.path.join(exp, "result_outputs", "summary.csv"))
    assert os.path.isfile(os.path.join(exp, "saved_models", "train_model_1"))
    # overwrite guard (ref: models/__init__.py:175-179)
    with pytest.raises(RuntimeError, match="overwriting existing data"):
        sisr_amd.cli.train_sisr(_config(name, tmp_path))


def test_batch_dict_schema():
    d = os.path.join(GOLDEN, "set5")
    ds = sisr_amd.data.SuperResImages(os.path.join(d, "lr_random_blur"), os.path.join(d, "hr"), split="all", scale=4,
                                      degradation_metadata_file=os.path.join(d, "lr_random_blur", "degradation_metadata.csv"),
                                      random_crop=16, random_augments=True)
    assert len(ds) == 5 and ds.metadata_keys == ["blur_kernel"] * 10
    loader = torch.utils.data.DataLoader(ds, batch_size=2)
    b = next(iter(loader))
    assert set(b) == {"lr", "hr", "tag", "hr_tag", "mask", "halfway_data", "metadata", "metadata_keys", "blur_kernels"}
    assert b["lr"].shape == (2, 3, 16, 16) and b["hr"].shape == (2, 3, 64, 64) and b["lr"].dtype == torch.float32
    assert b["metadata"].dtype == torch.float64 and b["metadata"].shape == (2, 10)
    assert len(b["metadata_keys"]) == 10 and b["metadata_keys"][0] == ("blur_kernel", "blur_kernel")
    assert b["tag"] == ["baby.png", "bird.png"]
    with pytest.raises(NotImplementedError):  # options that stay outside the HIP hot-path scope fail loudly
        sisr_amd.data.SuperResImages(os.path.join(d, "lr_random_blur"), os.path.join(d, "hr"), split="all",
                                     mask_data="somewhere")


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["edsr", "qedsr"])
def test_train_and_eval_entry_points_on_hip(name, tmp_path):
    ref = golden_json("g5_train_sisr")[name]["summary"]
    cfg = _config(name, tmp_path)
    cfg["training"]["gpu"] = "single"
    cfg["training"]["sp_gpu"] = 0
    total = sisr_amd.cli.train_sisr(cfg)
    np.testing.assert_allclose(total["train-loss"], ref["train-loss"], rtol=2e-4, atol=2e-5)
    np.testing.assert_allclose(total["val-loss"], ref["val-loss"], rtol=5e-4, atol=5e-5)
    np.testing.assert_allclose(total["val-PSNR"], ref["val-PSNR"], rtol=0, atol=5e-3)  # dB
    # eval_sisr on the epoch-1 checkpoint reproduces the last validation PSNR
    d = os.path.join(GOLDEN, "set5")
    df, avg = sisr_amd.cli.eval_sisr(model_and_epoch=[[cfg["experiment"], "1"]], model_loc=str(tmp_path), gpu=True,
                                     hr_dir=os.path.join(d, "hr"), lr_dir=os.path.join(d, "lr_random_blur"),
                                     full_directory=True, scale=4, out_loc=str(tmp_path), results_name="ev")
    assert len(df) == 5 and abs(float(avg["PSNR"].iloc[0]) - ref["val-PSNR"][1]) < 5e-3
    assert os.path.isfile(os.path.join(str(tmp_path), "ev", "standard_metrics", "average_metrics.csv"))


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["edsr", "qedsr"])
def test_device_tiles_reproduce_the_reference_run(name, tmp_path):
    """`[data] device_tiles = true`: the same experiment with the training batches cut on the GPU from a device-
    resident copy of the images must land on the reference's summary.csv too (same shuffle, same crops / flips)."""
    ref = golden_json("g5_train_sisr")[name]["summary"]
    cfg = _config(name, tmp_path)
    cfg["training"]["gpu"] = "single"
    cfg["training"]["sp_gpu"] = 0
    cfg["data"]["device_tiles"] = True
    total = sisr_amd.cli.train_sisr(cfg)
    np.testing.assert_allclose(total["train-loss"], ref["train-loss"], rtol=2e-4, atol=2e-5)
    np.testing.assert_allclose(total["val-loss"], ref["val-loss"], rtol=5e-4, atol=5e-5)
    np.testing.assert_allclose(total["val-PSNR"], ref["val-PSNR"], rtol=0, atol=5e-3)


@pytest.mark.gpu
def test_two_rank_cli_training_reproduces_the_reference_run(tmp_path):
    """`gpu = 'multi'` through the train entry point with TWO ranks (both on cuda:0, gloo transport) on the reference's
    own example experiment (G5: EDSR, batch 2, five Set5 images -> batches of 2, 2 and a ragged 1 per epoch, two epochs):
    every global batch is cut across the ranks (1 + 1, 1 + 1, 1 + 0), gradients are averaged with the ragged-batch
    weights, every rank walks the same shuffle in both epochs, rank 0 validates and writes -- and summary.csv is the
    reference's single-process one."""
    import json
    import subprocess
    import sys
    ref = golden_json("g5_train_sisr")["edsr"]["summary"]
    cfg = _config("edsr", tmp_path)
    cfg["training"]["gpu"] = "multi"
    cfg["training"]["sp_gpu"] = 0
    cfg_path = os.path.join(str(tmp_path), "cfg.json")
    with open(cfg_path, "w") as f:
        json.dump(cfg, f)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (f"import sys, json; sys.path.insert(0, {root!r}); import sisr_amd\n"
            f"total = sisr_amd.cli.train_sisr(json.load(open({cfg_path!r})))\n"
            "import torch.distributed as dist\n"
            f"\nif dist.get_rank() == 0: json.dump({{k: [float(x) for x in v] for k, v in total.items()}}, open({cfg_path!r} + '.out', 'w'))\n"
            "dist.barrier(); dist.destroy_process_group()\n")
    env = dict(os.environ, SISR_DIST_BACKEND="gloo", SISR_BENCH_SHARE_GPU="1", MASTER_ADDR="127.0.0.1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    prog = os.path.join(str(tmp_path), "run.py")
    with open(prog, "w") as f:
        f.write(code)
    res = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", "29777", prog], env=env, capture_output=True,
                         text=True, timeout=900)
    assert res.returncode == 0, res.stderr[-3000:]
    total = json.load(open(cfg_path + ".out"))
    np.testing.assert_allclose(total["train-loss"], ref["train-loss"], rtol=2e-4, atol=2e-5)
    np.testing.assert_allclose(total["val-loss"], ref["val-loss"], rtol=5e-4, atol=5e-5)
    np.testing.assert_allclose(total["val-PSNR"], ref["val-PSNR"], rtol=0, atol=5e-3)


def test_interp_input_keeps_hr_at_lr_size():
    """ref: sr_tools/data_handler.py:473-476: with input = 'interp' (SPARNet: LR images stored already interpolated) the HR
    image is cropped to the LR image's own size, not scale times it."""
    d = os.path.join(GOLDEN, "set5")
    ds = sisr_amd.data.SuperResImages(os.path.join(d, "hr"), os.path.join(d, "hr"), split="all", scale=4, input="interp")
    item = ds[0]
    assert item["lr"].shape == item["hr"].shape and item["lr"].shape[0] == 3
    with pytest.raises(RuntimeError):
        sisr_amd.data.SuperResImages(os.path.join(d, "hr"), os.path.join(d, "hr"), split="all", input="bicubic")
