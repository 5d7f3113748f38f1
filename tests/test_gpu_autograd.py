"""GPU: ``field(rays, is_train=True)`` is differentiable -- the reference's own training loop (TriPlane/main.py:272-299) runs UNCHANGED on the
drop-in field: its ``torch.mean((rgb_map - rgb_train) ** 2)``, its ``density_L1``, ``total_loss.backward()`` and
``torch.optim.Adam(field.get_optparam_groups(...), betas=(0.9, 0.99))``.  Checked against what the reference module itself produced for the
same two iterations (tests/golden/train_r1.npz: rgb_map, every gradient INCLUDING the L1 term, the parameters after two Adam steps), and
against the fused device trainer (ngf_amd.train.Trainer) that the same kernels serve.  C ABI: ngf_train_forward / ngf_train_backward_grad."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import field_for_case, load_case, load_train_case  # noqa: E402
import ngf_amd  # noqa: E402,F401
from ngf_amd import train  # noqa: E402

pytestmark = pytest.mark.gpu
GRAD_TOL = 1e-4


def rel(a, b):
    return float(np.abs(a - b).max()) / max(float(np.abs(b).max()), 1e-30)


def _adam(which):
    from ngf_amd import optim
    return {"torch": torch.optim.Adam, "ngf": optim.Adam}[which]


@pytest.mark.parametrize("opt", ["torch", "ngf"])
def test_the_references_training_loop_runs_unchanged_and_matches_its_gradients(opt):
    """The loop body below is main.py:272-299 line for line (names kept); only the two random draws of the forward are pinned to the captured
    ones (jitter=, coin=: keyword extras of the drop-in's forward) so that the reference's numbers can be compared.  ``opt = ngf``: the one
    changed line is the optimizer's class (ngf_amd.optim.Adam, same constructor): the same trajectory from one fused C-ABI call per step."""
    g, params = load_train_case("train_r1")
    field = field_for_case(g, params, None)
    nSamples = int(g["S"])
    rays_train, rgb_train = torch.from_numpy(g["rays"]).cuda(), torch.from_numpy(g["rgb_train"]).cuda()
    grad_vars = field.get_optparam_groups(0.02, 1e-3)                       # main.py:234 (args.lr_init, args.lr_basis)
    optimizer = _adam(opt)(grad_vars, betas=(0.9, 0.99))                    # main.py:241
    lr_factor = float(g["lr_factor"])
    L1_reg_weight = 8e-5
    PSNRs = []
    for iteration in range(int(g["steps"])):
        output = field(rays_train, is_train=True, white_bg=bool(g[f"white{iteration}"]), N_samples=nSamples, iteration=iteration,
                       jitter=torch.from_numpy(g[f"jitter{iteration}"]), coin=0.7)
        rgb_map = output['rgb_map']
        assert rgb_map.requires_grad and not output['depth_map'].requires_grad

        rgb_loss = torch.mean((rgb_map - rgb_train) ** 2)
        total_loss = rgb_loss
        if L1_reg_weight > 0:
            loss_reg_L1 = field.density_L1()
            total_loss += L1_reg_weight * loss_reg_L1

        optimizer.zero_grad()
        total_loss.backward()
        if iteration == 0:
            np.testing.assert_allclose(rgb_map.detach().cpu().numpy(), g["rgb_map0"], rtol=1e-4, atol=2e-6)
            assert abs(float(total_loss.detach()) - float(g["total_loss0"])) < 2e-6
            sd = dict(field.named_parameters())
            for name in train.PARAM_NAMES:
                got = sd[name].grad.cpu().numpy()
                assert rel(got, g[f"grad0.{name}"]) < GRAD_TOL, (name, rel(got, g[f"grad0.{name}"]))
        optimizer.step()

        rgb_loss = rgb_loss.detach().item()
        PSNRs.append(-10.0 * np.log(rgb_loss) / np.log(10.0))
        for param_group in optimizer.param_groups:
            param_group['lr'] = param_group['lr'] * lr_factor
    assert abs(rgb_loss - float(g["rgb_loss1"])) < 2e-3 * float(g["rgb_loss1"])
    sd = field.state_dict()
    for name in train.PARAM_NAMES:                                          # the trajectory, as test_gradients_and_two_adam_steps_match_reference holds it
        d = np.abs(sd[name].cpu().numpy() - g[f"after.{name}"])
        assert not np.array_equal(sd[name].cpu().numpy(), params[name])
        assert np.median(d) < 1e-5 and np.mean(d > 1e-3) < 0.02, (name, float(np.median(d)), float(np.mean(d > 1e-3)))
    # the eval launch sees the optimizer's in-place updates (torch.optim: Parameter._version moved; ngf_amd.optim: it invalidates the image itself)
    with torch.no_grad():
        out = field(rays_train, N_samples=nSamples, iteration=30001)
        fresh = field_for_case(g, {k: v.cpu().numpy() for k, v in field.state_dict().items()}, None)(rays_train, N_samples=nSamples, iteration=30001)
    assert torch.isfinite(out["rgb_map"]).all() and not out["rgb_map"].requires_grad
    assert torch.equal(out["rgb_map"], fresh["rgb_map"])


def test_fused_adam_follows_torch_adam_step_for_step():
    """ngf_amd.optim.Adam against torch.optim.Adam on two copies of one field, the reference's loop for six iterations (gauge planes join at
    iteration 3: gauge_start): parameters and optimizer state agree after every step; the state dict of one loads into the other; a parameter
    that is not the field's (an extra tensor in its own group) takes torch's path inside the same step; the differentiable engine's packed
    planes stay current without a re-pack (the next forward's rgb_map equals a freshly built field's)."""
    from ngf_amd import optim, synth
    g, params = load_train_case("train_r1")
    S = int(g["S"])
    rays, tgt = torch.from_numpy(g["rays"]).cuda(), torch.from_numpy(g["rgb_train"]).cuda()
    fa, fb = field_for_case(g, params, None), field_for_case(g, params, None)
    fa.gauge_start = fb.gauge_start = 3
    ea, eb = torch.nn.Parameter(torch.ones(5, device="cuda")), torch.nn.Parameter(torch.ones(5, device="cuda"))
    oa = torch.optim.Adam(fa.get_optparam_groups(0.02, 1e-3) + [{'params': [ea], 'lr': 0.1}], betas=(0.9, 0.99))
    ob = optim.Adam(fb.get_optparam_groups(0.02, 1e-3) + [{'params': [eb], 'lr': 0.1}], betas=(0.9, 0.99))
    for it in range(6):
        jit = torch.from_numpy(synth.hash_uniform(77, it, (rays.shape[0],)))
        for f, o, e in ((fa, oa, ea), (fb, ob, eb)):
            out = f(rays, is_train=True, white_bg=True, N_samples=S, iteration=it, jitter=jit)
            total = torch.mean((out["rgb_map"] - tgt) ** 2) + 8e-5 * f.density_L1() + (e ** 2).sum()
            o.zero_grad()
            total.backward()
            if f is fb:
                # the density / gauge gradients are sums of float atomics (order-dependent last bits) and Adam divides by sqrt(v): two runs of ONE
                # optimizer drift apart by ~1e-4 x lr where a gradient nearly cancels.  This test is about the update's arithmetic, so both
                # optimizers get the same gradient tensors (the trajectory against the reference's own is the loop test above)
                for (na, pa), (nb, pb) in zip(fa.named_parameters(), fb.named_parameters()):
                    assert (pa.grad is None) == (pb.grad is None), na
                    if pa.grad is not None:
                        assert float((pa.grad - pb.grad).abs().max()) <= 2e-3 * max(float(pa.grad.abs().max()), 1e-30), na      # (the density bias's is a cancelling sum of 600 k atomics: 1e-4 of itself between two runs)
                        pb.grad.copy_(pa.grad)
            o.step()
            for gr in o.param_groups:
                gr['lr'] = gr['lr'] * 0.999
        assert torch.allclose(ea, eb)
        for (na, pa), (nb, pb) in zip(fa.named_parameters(), fb.named_parameters()):
            d = float((pa.detach() - pb.detach()).abs().max())
            assert d <= 1e-6 * max(float(pa.detach().abs().max()), 1e-3) + 1e-8, (it, na, d)      # the same arithmetic on the same gradients: rounding of lr / (1 - beta1^t) only
            if pa.grad is None:
                assert pb.grad is None and pb not in ob.state, na
            else:
                sa, sb = oa.state[pa], ob.state[pb]
                assert float(sa['step']) == float(sb['step'])
                assert float((sa['exp_avg'] - sb['exp_avg']).abs().max()) <= 1e-5 * max(float(sa['exp_avg'].abs().max()), 1e-12), (it, na)
                assert float((sa['exp_avg_sq'] - sb['exp_avg_sq']).abs().max()) <= 1e-5 * max(float(sa['exp_avg_sq'].abs().max()), 1e-20), (it, na)
    # no re-pack happened on the fused side and none was needed: the engine renders the current parameters
    jit = torch.from_numpy(synth.hash_uniform(77, 99, (rays.shape[0],)))
    fb.zero_grad()
    got = fb(rays, is_train=True, white_bg=True, N_samples=S, iteration=9, jitter=jit)["rgb_map"].detach()
    sdb = {k: v.cpu().numpy() for k, v in fb.state_dict().items()}
    fc = field_for_case(g, sdb, None)
    fc.gauge_start = 3
    want = fc(rays, is_train=True, white_bg=True, N_samples=S, iteration=9, jitter=jit)["rgb_map"].detach()
    assert torch.equal(got, want)
    # checkpoints move between the two classes
    oc = torch.optim.Adam(fb.get_optparam_groups(0.02, 1e-3) + [{'params': [eb], 'lr': 0.1}], betas=(0.9, 0.99))
    oc.load_state_dict(ob.state_dict())
    od = optim.Adam(fa.get_optparam_groups(0.02, 1e-3) + [{'params': [ea], 'lr': 0.1}], betas=(0.9, 0.99))
    od.load_state_dict(oa.state_dict())
    assert float(oc.state[fb.plane_xy]['step']) == 6 and float(od.state[fa.gauge_xy]['step']) == 3


@pytest.mark.parametrize("name", ["triplane_r1_train_white", "triplane_r1_train_black"])
def test_differentiable_forward_matches_the_references_train_mode_forward(name):
    """The trainer's forward kernels (what the autograd path renders with) against the reference's own is_train=True forward: rgb_map and
    depth_map of the captured cases, and the fused eval launch on the same jitter."""
    g, params, step, mask = load_case(name)
    S, wb = int(g["S"]), bool(int(g["white_bg"]))
    f = field_for_case(g, params, mask)
    rays = torch.from_numpy(g["rays"]).cuda()
    kw = dict(white_bg=wb, is_train=True, N_samples=S, iteration=30001 if int(g["gauge_on"]) else -1, jitter=torch.from_numpy(g["jitter"]),
              coin=float(g["coin"]))
    f.gauge_start = 0
    out = f(rays, **kw)
    assert out["rgb_map"].requires_grad
    with torch.no_grad():
        fused = f(rays, **kw)
    assert not fused["rgb_map"].requires_grad
    rgb, depth = out["rgb_map"].detach().cpu().numpy(), out["depth_map"].cpu().numpy()
    np.testing.assert_allclose(rgb, g["rgb_map"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(depth, g["depth_map"], rtol=1e-4, atol=5e-5)
    np.testing.assert_allclose(rgb, fused["rgb_map"].cpu().numpy(), rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(depth, fused["depth_map"].cpu().numpy(), rtol=1e-4, atol=5e-5)


def test_arbitrary_loss_two_batches_in_one_graph_and_the_fused_trainer_agree():
    """d loss / d rgb_map is whatever the caller's loss makes it (here an L1 photometric loss over TWO batches rendered before one backward):
    the second forward takes the engine's buffers, so the first batch's backward renders it again (stale ticket) -- gradients equal the sum
    of the two batches done one at a time.  With an MSE loss the autograd path reproduces the fused trainer's gradients."""
    g, params, step, mask = load_case("triplane_r1_gauge")
    from ngf_amd import synth
    rays = torch.from_numpy(g["rays"]).cuda()
    n, S = rays.shape[0], 48
    tgt = torch.from_numpy(synth.hash_uniform(31, 1, (n, 3))).cuda()
    j1, j2 = (torch.from_numpy(synth.hash_uniform(31, k, (n,))) for k in (2, 3))
    f = field_for_case(g, params, None)
    ps = dict(f.named_parameters())

    def grads_of(loss):
        f.zero_grad(set_to_none=True)
        loss.backward()
        return {k: (None if p.grad is None else p.grad.detach().clone()) for k, p in ps.items()}

    l1 = lambda j: (f(rays, is_train=True, N_samples=S, iteration=5, jitter=j)["rgb_map"] - tgt).abs().mean()      # noqa: E731
    ga, gb = grads_of(l1(j1)), grads_of(l1(j2))
    both = grads_of(l1(j1) + l1(j2))
    for k in train.PARAM_NAMES:
        want = ga[k] + gb[k]
        assert float((both[k] - want).abs().max()) <= 2e-5 * max(float(want.abs().max()), 1e-30), k
    # a LARGER batch between a forward and its backward rebuilds the engine: the pending batch is rendered again on the new one
    out_small = f(rays[:64], is_train=True, N_samples=S, iteration=5, jitter=j1[:64])["rgb_map"]
    small_alone = grads_of((f(rays[:64], is_train=True, N_samples=S, iteration=5, jitter=j1[:64])["rgb_map"] - tgt[:64]).abs().mean())
    big = torch.cat([rays, rays], 0)
    out_big = f(big, is_train=True, N_samples=S + 8, iteration=5, jitter=torch.cat([j1, j2]))["rgb_map"]        # outgrows max_rays and max_samples
    got = grads_of((out_small - tgt[:64]).abs().mean())
    for k in train.PARAM_NAMES:
        assert float((got[k] - small_alone[k]).abs().max()) <= 2e-5 * max(float(small_alone[k].abs().max()), 1e-30), k
    del out_big
    # MSE: the fused trainer's gradient of the same batch
    tr = train.Trainer(field_for_case(g, params, None), batch_size=n, max_samples=S, chunk_samples=0)
    tr.backward(rays, tgt, S, white_bg=True, iteration=5, jitter=j1)
    mse = grads_of(torch.mean((f(rays, is_train=True, N_samples=S, iteration=5, jitter=j1)["rgb_map"] - tgt) ** 2))
    for k, name in enumerate(train.PARAM_NAMES):
        a, b = mse[name], tr.gradient(k)
        assert float((a - b).abs().max()) <= 2e-5 * max(float(b.abs().max()), 1e-30), name
    tr.release()


def test_the_references_loop_goes_through_alpha_mask_shrink_and_up_sampling():
    """The structural steps of the reference's loop (TriPlane/main.py:329-356) on the AUTOGRAD path: ``updateAlphaMask`` installs a mask,
    ``shrink`` crops the planes into new Parameters and moves the aabb (the mask keeps its own), ``up_sampling`` replaces the planes again,
    and each time the reference builds a new ``torch.optim.Adam``.  ``forward`` rebuilds the field's gradient engine behind the scenes; after
    every step the reference loop's gradients (MSE + density_L1) equal autograd of the eager port built from the field's CURRENT state."""
    from oracle import train as otrain
    from ngf_amd import synth
    g, params, step, mask = load_case("triplane_r1_gauge")
    field = field_for_case(g, params, None)
    with torch.no_grad():       # the seeded fog fills the box: carve an occupied column (x, y in the middle half, every z) so that shrink has something to crop
        w = field.density_decoder.weight[0, :16]                       # the xy plane's share of the linear density decoder (bias 10, then - 10)
        H, W = field.plane_xy.shape[-2:]
        yy, xx = torch.meshgrid(torch.arange(H, device="cuda"), torch.arange(W, device="cuda"), indexing="ij")
        inside = ((yy - (H - 1) / 2).abs() < H / 4) & ((xx - (W - 1) / 2).abs() < W / 4)
        field.plane_xy[0, :16] = torch.where(inside, -2.0, -25.0)[None] * torch.sign(w)[:, None, None] / w.abs().sum()      # pre-softplus -2 (alpha ~0.2 per step) / -25 (empty)
        field.plane_yz[0, :16] = 0.0
        field.plane_xz[0, :16] = 0.0
    rays_np = synth.lookat_rays(36, 36)
    rays_train = torch.from_numpy(rays_np).cuda()
    n, nSamples = rays_np.shape[0], 80
    rgb_np = synth.hash_uniform(57, 1, (n, 3))
    rgb_train = torch.from_numpy(rgb_np).cuda()
    L1_reg_weight = 8e-5

    def one_iteration(tag, iteration, optimizer):
        jit_np = synth.hash_uniform(57, 10 + iteration, (n,))
        cur = {name: p.detach().cpu().numpy().copy() for name, p in zip(train.PARAM_NAMES, train._train_params(field))}
        am = None
        if field.alphaMask is not None:
            am = (field.alphaMask.alpha_volume[0, 0].float().cpu().numpy(), field.alphaMask.aabb.cpu().numpy())
        orc = otrain.EagerTrainer(cur, field.aabb.cpu().numpy(), float(field.stepSize), tuple(field.near_far), float(field.distance_scale),
                                  float(field.rayMarch_weight_thres), alpha_mask=am)
        want, want_loss, want_rgb, aux = orc.gradients(torch.from_numpy(rays_np), torch.from_numpy(rgb_np), nSamples, torch.from_numpy(jit_np), True, iteration)
        assert int(aux["active"].sum()) > 50, tag                       # the colour path carries gradient in every phase
        # main.py:272-296
        output = field(rays_train, is_train=True, white_bg=True, N_samples=nSamples, iteration=iteration, jitter=torch.from_numpy(jit_np), coin=0.7)
        rgb_map = output['rgb_map']
        rgb_loss = torch.mean((rgb_map - rgb_train) ** 2)
        mse = float(rgb_loss.detach())                                  # (the reference's `total_loss +=` below adds the L1 term into rgb_loss in place)
        total_loss = rgb_loss
        loss_reg_L1 = field.density_L1()
        total_loss += L1_reg_weight * loss_reg_L1
        optimizer.zero_grad()
        total_loss.backward()
        np.testing.assert_allclose(rgb_map.detach().cpu().numpy(), want_rgb.numpy(), rtol=1e-4, atol=2e-6, err_msg=tag)
        assert abs(mse - want_loss) < 2e-6, tag
        sd = dict(field.named_parameters())
        for name in train.PARAM_NAMES:
            got = sd[name].grad.cpu().numpy()
            assert got.shape == cur[name].shape and rel(got, want[name].numpy()) < GRAD_TOL, (tag, name, rel(got, want[name].numpy()))
        optimizer.step()

    new_opt = lambda: torch.optim.Adam(field.get_optparam_groups(0.02, 1e-3), betas=(0.9, 0.99))      # noqa: E731  (main.py:241, :354-356)
    optimizer = new_opt()
    one_iteration("fresh field", 0, optimizer)
    one_iteration("after an Adam step", 1, optimizer)
    new_aabb = field.updateAlphaMask((48, 48, 48))                       # main.py:334
    one_iteration("alpha mask", 2, optimizer)
    shape0 = tuple(field.plane_xy.shape)
    field.shrink(new_aabb)                                               # main.py:336
    assert tuple(field.plane_xy.shape) != shape0
    optimizer = new_opt()            # (the reference keeps its old optimizer until the next up-sampling: the cropped planes get no updates and no zero_grad in
                                     # between -- train.fit mirrors that; a new one here, so that .grad is this iteration's alone)
    one_iteration("shrunk planes, mask on the old box", 3, optimizer)
    res = [int(r) + 12 for r in field.gridSize]
    field.up_sampling(res)                                               # main.py:350
    optimizer = new_opt()                                                # main.py:354-356
    one_iteration("up-sampled planes", 4, optimizer)
    one_iteration("and one more step on them", 5, optimizer)


def test_gauge_off_frozen_parameters_and_the_guards():
    g, params = load_train_case("train_r1")
    f = field_for_case(g, params, None)
    f.gauge_start = 10
    S = int(g["S"])
    rays, tgt = torch.from_numpy(g["rays"]).cuda(), torch.from_numpy(g["rgb_train"]).cuda()
    f.rgb_decoder.basis.weight.requires_grad_(False)
    out = f(rays, is_train=True, N_samples=S, iteration=0, jitter=torch.from_numpy(g["jitter0"]))
    torch.mean((out["rgb_map"] - tgt) ** 2).backward()
    # iteration < gauge_start: compute_gauge is not evaluated (Field.py:58,73) -> .grad stays None and torch.optim skips the gauge planes
    assert f.gauge_xy.grad is None and f.rgb_decoder.basis.weight.grad is None and f.plane_xy.grad is not None
    # in-place write between forward and backward: autograd refuses (saved tensors' version counters), as it does for the reference
    out = f(rays, is_train=True, N_samples=S, iteration=0, jitter=torch.from_numpy(g["jitter0"]))
    with torch.no_grad():
        f.plane_xy.mul_(1.0)
    with pytest.raises(RuntimeError, match="modified by an inplace operation"):
        out["rgb_map"].sum().backward()
    # no graph is built where the reference builds none
    with torch.no_grad():
        assert not f(rays, is_train=True, N_samples=S, iteration=0)["rgb_map"].requires_grad
    assert not f(rays, is_train=False, N_samples=S, iteration=0)["rgb_map"].requires_grad
    # the InfoInv tree has no backward (out of scope): it raises instead of returning pixels without a graph
    gi, pi, _, mi = load_case("infoinv_r1_on")
    fi = field_for_case(gi, pi, mi)
    with pytest.raises(NotImplementedError):
        fi(torch.from_numpy(gi["rays"]).cuda(), is_train=True, N_samples=int(gi["S"]))
    with torch.no_grad():
        assert torch.isfinite(fi(torch.from_numpy(gi["rays"]).cuda(), is_train=True, N_samples=int(gi["S"]))["rgb_map"]).all()


def test_full_size_batch_through_the_autograd_path_matches_the_fused_trainer():
    """The reference's training shape (4096 rays x 884 samples, 256^2 planes): gradients of the two-call form against the fused step's."""
    from helpers import big_case
    from ngf_amd import synth
    g, params, step = big_case("triplane", "R1")
    f = field_for_case(g, params, None)
    S = int(f.nSamples)
    frame = synth.lookat_rays(800, 800)
    pick = (synth.hash_uniform(9, 1, (4096,)) * np.float32(frame.shape[0])).astype(np.int64)
    rays = torch.from_numpy(frame[pick]).cuda()
    tgt = torch.from_numpy(synth.hash_uniform(9, 2, (4096, 3))).cuda()
    jit = torch.from_numpy(synth.hash_uniform(9, 3, (4096,)))
    out = f(rays, is_train=True, N_samples=S, iteration=7, jitter=jit)
    loss = torch.mean((out["rgb_map"] - tgt) ** 2)
    loss.backward()
    tr = train.Trainer(field_for_case(g, params, None), batch_size=4096, max_samples=S)
    l2 = tr.backward(rays, tgt, S, white_bg=True, iteration=7, jitter=jit)
    assert abs(loss.item() - l2.item()) < 1e-7 and f._grad_engine.last_active == tr.last_active
    sd = dict(f.named_parameters())
    for k, name in enumerate(train.PARAM_NAMES):
        a, b = sd[name].grad, tr.gradient(k)
        assert float((a - b).abs().max()) <= 2e-5 * max(float(b.abs().max()), 1e-30), name
    tr.release()


def test_density_l1_matches_torchs_expression_and_gradient():
    """TriPlane.density_L1 (Field.py:149-152) on the device: ngf_planes_l1 / ngf_planes_l1_backward behind one autograd node against the
    reference's torch expression -- the value to float32 rounding of a 4 M-term mean, the gradient sign(p) * upstream / numel bit for bit
    (zeros get a zero gradient like torch.sgn); rectangular planes whose size is not a multiple of four; a frozen plane gets no gradient."""
    g, params, step, mask = load_case("triplane_r1_gauge")
    f = field_for_case(g, params, None)
    with torch.no_grad():
        f.plane_xy[0, 3, 5:9, :] = 0.0
    ref = torch.mean(torch.abs(f.plane_xy)) + torch.mean(torch.abs(f.plane_yz)) + torch.mean(torch.abs(f.plane_xz))
    got = f.density_L1()
    assert got.shape == ref.shape and got.requires_grad
    assert abs(float(got) - float(ref)) <= 2e-6 * float(ref)
    w = torch.tensor(8e-5, device="cuda")
    want = torch.autograd.grad(ref * w, [f.plane_xy, f.plane_yz, f.plane_xz])
    have = torch.autograd.grad(got * w, [f.plane_xy, f.plane_yz, f.plane_xz])
    for a, b in zip(have, want):
        assert torch.equal(a, b)
    assert int((have[0] == 0).sum()) == 4 * f.plane_xy.shape[3]
    f.plane_yz.requires_grad_(False)
    (f.density_L1() * 3.0).backward()
    assert f.plane_yz.grad is None and f.plane_xy.grad is not None
    assert torch.allclose(f.plane_xy.grad, want[0] * (3.0 / 8e-5), rtol=1e-6, atol=0)


def test_a_differentiable_forward_beyond_grad_max_pairs_is_chunked_and_gives_the_same_gradients():
    """``field(rays, is_train=True)`` under autograd keeps activation rows that grow with n x S.  Beyond ``field.grad_max_pairs`` the
    call is cut into ray chunks (one autograd node each, the engine holds one chunk, earlier chunks are rendered again in the backward): same
    rgb_map bit for bit, the same gradients up to the order of their sums, the engine sized for a chunk and not for the batch."""
    g, params, step, mask = load_case("triplane_r1_gauge")
    from ngf_amd import synth
    rays = torch.from_numpy(g["rays"]).cuda()
    n, S = rays.shape[0], 48
    tgt = torch.from_numpy(synth.hash_uniform(41, 1, (n, 3))).cuda()
    jit = torch.from_numpy(synth.hash_uniform(41, 2, (n,)))
    fa, fb = field_for_case(g, params, None), field_for_case(g, params, None)
    fb.grad_max_pairs = S * 37                       # 37 rays per chunk
    oa = fa(rays, is_train=True, N_samples=S, iteration=5, jitter=jit)
    ob = fb(rays, is_train=True, N_samples=S, iteration=5, jitter=jit)
    assert torch.equal(oa["rgb_map"], ob["rgb_map"]) and torch.equal(oa["depth_map"], ob["depth_map"])
    assert fb._grad_engine.max_rays == 37 and fa._grad_engine.max_rays == n
    ((oa["rgb_map"] - tgt) ** 2).mean().backward()
    ((ob["rgb_map"] - tgt) ** 2).mean().backward()
    for (na, pa), (nb, pb) in zip(fa.named_parameters(), fb.named_parameters()):
        assert float((pa.grad - pb.grad).abs().max()) <= 2e-5 * max(float(pa.grad.abs().max()), 1e-30), na
