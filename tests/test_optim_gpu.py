"""optim.FlatAdam (one-launch Adam over flat arenas) against torch.optim.Adam on a real MI355X (pytest -m gpu)."""
import copy

import numpy as np
import pytest
import torch

import sisr_amd

pytestmark = pytest.mark.gpu


def _nets():
    torch.manual_seed(3)
    a = sisr_amd.architectures.QEDSR(num_features=64, num_blocks=2, scale=2, input_para=10).to("cuda:0")
    b = copy.deepcopy(a)
    return a, b


def test_flat_adam_matches_torch_adam_and_keeps_its_state_dict_schema():
    """Seven steps with changing gradients (one parameter without a gradient in steps 3-4, as torch skips it): parameters
    and both moment estimates agree with torch.optim.Adam to fp32 rounding, state_dict() has torch's keys / shapes /
    per-parameter step counts, and a state dict round-trips through load_state_dict into a fresh FlatAdam."""
    na, nb = _nets()
    fa = sisr_amd.optim.FlatAdam(list(na.parameters()), lr=1e-3, betas=(0.9, 0.999))
    ta = torch.optim.Adam(list(nb.parameters()), lr=1e-3, betas=(0.9, 0.999))
    sched_a = torch.optim.lr_scheduler.CosineAnnealingWarmRestarts(fa, T_0=3, T_mult=1, eta_min=1e-7)
    sched_b = torch.optim.lr_scheduler.CosineAnnealingWarmRestarts(ta, T_0=3, T_mult=1, eta_min=1e-7)
    g = torch.Generator().manual_seed(9)
    pa, pb = list(na.parameters()), list(nb.parameters())
    assert all(p.data_ptr() >= fa.flat_p.data_ptr() and p.data_ptr() < fa.flat_p.data_ptr() + 4 * fa.total for p in pa)
    for step in range(7):
        for i, (x, y) in enumerate(zip(pa, pb)):
            if i == 5 and step in (3, 4):
                x.grad = y.grad = None
                continue
            gr = (torch.randn(x.shape, generator=g) * 10 ** float(torch.randint(-4, 1, (1,), generator=g))).cuda()
            x.grad, y.grad = gr.clone(), gr.clone()
        fa.step()
        ta.step()
        sched_a.step()
        sched_b.step()
    for x, y in zip(pa, pb):
        np.testing.assert_allclose(x.detach().cpu().numpy(), y.detach().cpu().numpy(), rtol=2e-6, atol=1e-7)  # ~ulps of the weights
    sa, sb = fa.state_dict(), ta.state_dict()
    assert sa["param_groups"][0].keys() == sb["param_groups"][0].keys() and sa["state"].keys() == sb["state"].keys()
    for k in sb["state"]:
        assert sa["state"][k].keys() == sb["state"][k].keys()
        assert float(sa["state"][k]["step"]) == float(sb["state"][k]["step"]), k
        for key in ("exp_avg", "exp_avg_sq"):
            want = sb["state"][k][key].cpu().numpy()
            np.testing.assert_allclose(sa["state"][k][key].cpu().numpy(), want, rtol=2e-6, atol=1e-6 * np.abs(want).max())
    # round trip into a fresh optimiser over a fresh copy of the net, then one more identical step on both
    nc = copy.deepcopy(nb)
    with torch.no_grad():
        for z, y in zip(nc.parameters(), nb.parameters()):
            z.copy_(y)
    fc = sisr_amd.optim.FlatAdam(list(nc.parameters()), lr=1e-3)
    fc.load_state_dict(sb)
    for z, y in zip(nc.parameters(), pb):
        gr = torch.randn(z.shape, generator=g).cuda() * 1e-2
        z.grad, y.grad = gr.clone(), gr.clone()
    fc.step()
    ta.step()
    for z, y in zip(nc.parameters(), pb):
        np.testing.assert_allclose(z.detach().cpu().numpy(), y.detach().cpu().numpy(), rtol=2e-6, atol=1e-7)


def test_handlers_train_with_flat_adam_and_conv_gradients_land_in_its_arena():
    h = sisr_amd.available_models["edsr"](device=0, model_save_dir="/tmp", eval_mode=False, scale=4, lr=1e-4, num_blocks=2)
    assert isinstance(h.optimizer, sisr_amd.optim.FlatAdam)
    x, y = torch.rand(2, 3, 16, 16), torch.rand(2, 3, 64, 64)
    h.run_train(x, y)
    w = h.net.body[0].body[0].weight
    assert w.grad.data_ptr() == h.optimizer.grad_views[w].data_ptr()  # written there by the weight-gradient kernel
    sd = h.save_model("x", 0, extract_state_only=True)
    assert sorted(sd["optimizer"]["state"][0].keys()) == ["exp_avg", "exp_avg_sq", "step"]
