"""CPU: the training-step oracle (oracle/train.py: autograd of the eager port + restated Adam) against the gradients and
the two Adam steps captured from the reference module itself (tests/golden/train_r1.npz, make_golden.py capture_train)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import train as otrain  # noqa: E402
from helpers import load_train_case  # noqa: E402


def _close(a, b, tol):
    scale = max(float(np.abs(b).max()), 1e-30)
    return float(np.abs(a - b).max()) <= tol * scale


def test_gradients_and_adam_match_reference():
    g, params = load_train_case("train_r1")
    tr = otrain.EagerTrainer(params, g["aabb"], float(g["stepSize"]), g["near_far"], float(g["distance_scale"]), float(g["thr"]))
    rays, tgt = torch.from_numpy(g["rays"]), torch.from_numpy(g["rgb_train"])
    S = int(g["S"])
    groups = otrain.param_groups(params)
    lr = {k: lr0 for names, lr0 in groups for k in names}
    m = {k: torch.zeros_like(v) for k, v in tr.p.items()}
    v = {k: torch.zeros_like(p) for k, p in tr.p.items()}
    for it in range(int(g["steps"])):
        grads, rgb_loss, rgb_map, _ = tr.gradients(rays, tgt, S, torch.from_numpy(g[f"jitter{it}"]), bool(g[f"white{it}"]), it)
        assert abs(rgb_loss - float(g[f"rgb_loss{it}"])) < 1e-6
        np.testing.assert_allclose(rgb_map.numpy(), g[f"rgb_map{it}"], rtol=0, atol=2e-6)
        for k in params:
            assert _close(grads[k].numpy(), g[f"grad{it}.{k}"], 2e-4), (it, k)
        with torch.no_grad():
            for k, p in tr.p.items():
                new_p, m[k], v[k] = otrain.adam_update(p, grads[k], m[k], v[k], it + 1, lr[k])
                p.copy_(new_p)
        lr = {k: x * float(g["lr_factor"]) for k, x in lr.items()}
    for k in params:
        assert _close(tr.p[k].detach().numpy(), g[f"after.{k}"], 2e-5), k
        assert not np.array_equal(g[f"after.{k}"], params[k]), k         # every group really moved


def test_restated_adam_equals_torch_optim():
    torch.manual_seed(0)
    p0 = torch.randn(257)
    ref = torch.nn.Parameter(p0.clone())
    opt = torch.optim.Adam([ref], lr=0.02, betas=(0.9, 0.99))
    p, m, v = p0.clone(), torch.zeros(257), torch.zeros(257)
    for t in range(1, 6):
        g = torch.randn(257) * 10 ** (t - 3)
        ref.grad = g.clone()
        opt.step()
        p, m, v = otrain.adam_update(p, g, m, v, t, 0.02)
        np.testing.assert_allclose(p.numpy(), ref.detach().numpy(), rtol=2e-6, atol=1e-7)
