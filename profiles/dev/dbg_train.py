import os, sys
ROOT = "/root/repo"
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from helpers import field_for_case, load_train_case
import ngf_amd
from ngf_amd import train
g, params = load_train_case("train_r1")
f = field_for_case(g, params, None)
S = int(g["S"])
tr = train.Trainer(f, batch_size=g["rays"].shape[0], max_samples=S)
rays, tgt = torch.from_numpy(g["rays"]).cuda(), torch.from_numpy(g["rgb_train"]).cuda()
loss = tr.backward(rays, tgt, S, white_bg=True, iteration=0, jitter=torch.from_numpy(g["jitter0"]))
print("loss", loss.item(), float(g["rgb_loss0"]), "active", tr.last_active)
for k, name in enumerate(train.PARAM_NAMES):
    got = tr.gradient(k).cpu().numpy(); want = g[f"grad0.{name}"].copy()
    if k < 3: want = want - 8e-5 * np.sign(params[name]) / params[name].size
    d = np.abs(got - want)
    i = np.unravel_index(d.argmax(), d.shape)
    print(f"{name:28s} rel {d.max()/np.abs(want).max():.3e}  max|want| {np.abs(want).max():.3e} at {i} got {got[i]:.6e} want {want[i]:.6e}")
    if k < 3:
        for c0, c1 in ((0, 16), (16, 64)):
            dd = d[:, c0:c1]; print(f"      channels {c0}:{c1} rel {dd.max()/np.abs(want[:, c0:c1]).max():.3e}")
