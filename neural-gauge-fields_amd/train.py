"""One TriPlane training iteration on the device -- host-side mirror of the body of the
reference's training loop (TriPlane/main.py:264-299):

    output = field(rays_train, is_train=True, white_bg=white_bg, N_samples=nSamples, iteration=iteration)
    total_loss = mean((rgb_map - rgb_train)**2) + L1_reg_weight * field.density_L1()
    optimizer.zero_grad(); total_loss.backward(); optimizer.step()          # Adam(get_optparam_groups, betas=(0.9, 0.99))
    for g in optimizer.param_groups: g['lr'] *= lr_factor

``Trainer.step(rays_train, rgb_train, iteration)`` does exactly that through libngf_hip.so (csrc/ngf_train.hpp); the field's
nn.Parameters are updated in place, like torch.optim does, so ``field(...)`` / ``field.save(...)`` see the new values.
Random inputs of the reference (the per-ray jitter torch.rand_like of sample_ray and the white-background coin of
FieldBase.py:299) are drawn here with torch's generators, or passed in for parity tests.  There is no CPU path.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib

PARAM_NAMES = ("plane_xy", "plane_yz", "plane_xz", "gauge_xy", "gauge_yz", "gauge_xz", "density_decoder.weight",
               "density_decoder.bias", "rgb_decoder.basis.weight", "rgb_decoder.mlp.0.weight", "rgb_decoder.mlp.0.bias",
               "rgb_decoder.mlp.2.weight", "rgb_decoder.mlp.2.bias", "rgb_decoder.mlp.4.weight", "rgb_decoder.mlp.4.bias")
L1_REG_WEIGHT = 8e-5            # main.py:262


class TrainDesc(C.Structure):
    _fields_ = [
        ("aabb", C.c_float * 6), ("near_", C.c_float), ("far_", C.c_float), ("step", C.c_float), ("distance_scale", C.c_float),
        ("weight_thres", C.c_float),
        ("plane", C.c_void_p * 3), ("plane_h", C.c_int32 * 3), ("plane_w", C.c_int32 * 3),
        ("gauge", C.c_void_p * 3), ("gauge_h", C.c_int32 * 3), ("gauge_w", C.c_int32 * 3),
        ("dens_w", C.c_void_p), ("dens_b", C.c_void_p), ("basis", C.c_void_p), ("w1", C.c_void_p), ("b1", C.c_void_p),
        ("w2", C.c_void_p), ("b2", C.c_void_p), ("w3", C.c_void_p), ("b3", C.c_void_p),
        ("exp_avg", C.c_void_p * 15), ("exp_avg_sq", C.c_void_p * 15),
        ("mask_bits", C.c_void_p), ("mask_d", C.c_int32), ("mask_h", C.c_int32), ("mask_w", C.c_int32), ("mask_aabb", C.c_float * 6),
        ("max_rays", C.c_int64), ("max_samples", C.c_int32), ("chunk_samples", C.c_int64),
    ]


def _bind(L):
    if getattr(L, "_ngf_train_bound", False):
        return
    L.ngf_trainer_create.argtypes = [C.POINTER(TrainDesc), C.POINTER(C.c_void_p), C.c_void_p]
    L.ngf_trainer_destroy.argtypes = [C.c_void_p]
    L.ngf_trainer_bytes.restype = C.c_int64
    L.ngf_trainer_bytes.argtypes = [C.c_void_p]
    L.ngf_train_backward2.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_int32,
                                      C.POINTER(C.c_int64), C.c_void_p]
    L.ngf_train_forward.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p,
                                    C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.c_void_p]
    L.ngf_train_backward_grad.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]
    L.ngf_train_get_grad.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]
    L.ngf_train_get_active.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]
    L.ngf_train_overflow_count.argtypes = [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.c_void_p]
    L.ngf_train_adam.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, C.c_void_p]
    L.ngf_train_adam_all.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_float, C.c_float, C.c_void_p]
    L.ngf_train_params_changed.argtypes = [C.c_void_p]
    L.ngf_train_get_grads.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    L.ngf_train_adam_ext.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_float, C.c_void_p]
    if L.ngf_sizeof_train_desc() != C.sizeof(TrainDesc):
        raise RuntimeError("libngf_hip.so ABI mismatch (ngf_train_desc layout)")
    L._ngf_train_bound = True


def _train_params(field):
    """The fifteen parameters in ngf_train_desc order (PARAM_NAMES), checked for what the kernels read: contiguous float32 device tensors."""
    # (module, attribute) of every name resolved once per field; a call is fifteen dictionary look-ups, and the checks run again only when one of
    # the Parameter OBJECTS changed (up_sampling / shrink / load replace them) -- this sits in front of every differentiable forward
    slots = getattr(field, '_tp_slots', None)
    if slots is None:
        slots = []
        for name in PARAM_NAMES:
            *path, attr = name.split('.')
            m = field
            for part in path:
                m = getattr(m, part)
            slots.append((m, attr))
        field._tp_slots = slots
    params = [m._parameters[attr] for m, attr in slots]
    last = getattr(field, '_tp_checked', None)
    if last is None or any(a is not b for a, b in zip(params, last)):
        for name, p in zip(PARAM_NAMES, params):
            if not (p.is_cuda and p.dtype == torch.float32 and p.is_contiguous()):
                raise RuntimeError(f"parameter {name} must be a contiguous float32 device tensor")
        field._tp_checked = params
    return params


def _train_desc(f, params, exp_avg, exp_avg_sq, max_rays, max_samples, chunk_samples):
    """ngf_train_desc of field ``f``; ``exp_avg`` / ``exp_avg_sq`` None = a trainer without optimiser (ngf_train_forward / _backward_grad only)."""
    d = TrainDesc()
    dev = torch.device(f.device)
    d.aabb = (C.c_float * 6)(*f._aabb_host())
    d.near_, d.far_ = float(f.near_far[0]), float(f.near_far[1])
    d.step = float(f.stepSize)
    d.distance_scale = float(f.distance_scale)
    d.weight_thres = float(np.float32(f.rayMarch_weight_thres))
    for k in range(3):
        d.plane[k] = params[k].data_ptr()
        d.plane_h[k], d.plane_w[k] = params[k].shape[2], params[k].shape[3]
        d.gauge[k] = params[3 + k].data_ptr()
        d.gauge_h[k], d.gauge_w[k] = params[3 + k].shape[2], params[3 + k].shape[3]
    (d.dens_w, d.dens_b, d.basis, d.w1, d.b1, d.w2, d.b2, d.w3, d.b3) = [p.data_ptr() for p in params[6:]]
    if exp_avg is not None:
        for k in range(15):
            d.exp_avg[k] = exp_avg[k].data_ptr()
            d.exp_avg_sq[k] = exp_avg_sq[k].data_ptr()
    keep = []
    if f.alphaMask is not None:
        bits = f.alphaMask.packed_bits_device().to(dev)
        keep.append(bits)
        d.mask_bits = bits.data_ptr()
        shp = f.alphaMask.alpha_volume.shape
        d.mask_d, d.mask_h, d.mask_w = int(shp[-3]), int(shp[-2]), int(shp[-1])
        d.mask_aabb = (C.c_float * 6)(*f.alphaMask.aabb.reshape(-1).tolist())
    d.max_rays, d.max_samples, d.chunk_samples = int(max_rays), int(max_samples), int(chunk_samples)
    return d, keep


def _field_key(f, params):
    return (tuple((p.data_ptr(), tuple(p.shape)) for p in params), float(f.stepSize), f._aabb_host(),
            None if f.alphaMask is None else f.alphaMask.alpha_volume.data_ptr(),
            tuple(f.near_far), float(f.distance_scale), float(f.rayMarch_weight_thres))


class Trainer:
    """Adam state + the device trainer of one TriPlane field.

    ``lr_init`` / ``lr_basis`` / ``lr_decay_*`` / ``n_iters`` are the flags of TriPlane/opt.py; ``batch_size`` and
    ``max_samples`` size the scratch buffers (args.batch_size and the largest N_samples that will be passed)."""

    def __init__(self, field, batch_size=4096, max_samples=None, lr_init=0.02, lr_basis=1e-3, lr_decay_iters=-1,
                 lr_decay_target_ratio=0.1, n_iters=30000, L1_reg_weight=L1_REG_WEIGHT, betas=(0.9, 0.99), eps=1e-8, chunk_samples=None,
                 frozen=(), state_from=None, speculative=False, check_every=16):
        """``chunk_samples`` sizes the activation rows (2.4 KB per active sample), ngf_train_desc.chunk_samples of include/ngf.h.
        ``None`` (default): rows for a third of the batch's (ray, sample) pairs, at least 262 144 -- 3.4 GiB for 4096 rays x 884 samples instead
        of the 9.0 GiB of rows for every pair.  What happens when a batch has more active samples than rows is ``speculative``'s choice:

        * ``speculative=False`` (DEFAULT, fail-safe): the step reads the active count on the host (ONE stream synchronisation per step -- the
          reference's loop synchronises every iteration as well, ``rgb_loss.detach().item()``, main.py:297) and works a longer list through
          chunk by chunk: every step applies a complete gradient, nothing is ever skipped.
        * ``speculative=True`` (opt-in, no host round trip at all): a batch that does not fit is flagged ON THE DEVICE -- its Adam update leaves
          parameters and moments untouched (a truncated gradient is never applied) and its loss is NaN.  ``optimizer_step`` looks at the flag
          counter every ``check_every`` steps (one synchronisation each; ``check_rows()`` does it on demand), warns, doubles the rows and takes
          the skipped steps back out of its step counters and learning-rate decay.  Steps taken between an overflow and its detection ran with
          counters advanced by the skipped ones (bias correction / lr decay off by that many steps); use the default when that matters.

        An explicit ``chunk_samples``: ``0`` = rows for every pair (never overflows, no synchronisation); ``> 0`` = that many rows, host count
        (the default's path); ``< 0`` = speculative with that many rows."""
        self.field = field
        self.dev = torch.device(field.device)
        if self.dev.type != "cuda":
            raise RuntimeError("ngf_amd.train.Trainer runs on the GPU only (device='cuda'); there is no CPU path")
        self.L = _lib.lib()
        _bind(self.L)
        self.params = _train_params(field)
        self.exp_avg = [torch.zeros_like(p) for p in self.params]
        self.exp_avg_sq = [torch.zeros_like(p) for p in self.params]
        self.steps = [0] * 15                       # torch.optim.Adam keeps one step counter per parameter
        self.frozen = set(int(k) for k in frozen)   # parameter indices optimizer_step leaves alone
        # get_optparam_groups (Field.py:34-46)
        net = lr_basis
        self.lr = [lr_init] * 3 + [net * 0.1] * 3 + [net] * 9
        if lr_decay_iters > 0:
            self.lr_factor = lr_decay_target_ratio ** (1 / lr_decay_iters)
        else:
            self.lr_factor = lr_decay_target_ratio ** (1 / n_iters)
        self.betas, self.eps, self.l1 = betas, eps, L1_reg_weight
        self.batch_size = int(batch_size)
        self.max_samples = int(max_samples if max_samples is not None else field.nSamples)
        pairs = self.batch_size * self.max_samples
        if chunk_samples is None:
            rows = (max(1 << 18, -(-pairs // 3)) + 15) // 16 * 16
            chunk_samples = 0 if rows >= pairs else (-rows if speculative else rows)
        self.chunk_samples = int(chunk_samples)
        self.check_every = max(1, int(check_every))
        self._since_check = 0
        self._applied = []                          # speculative rows: (parameter indices, lr factor) of the optimizer steps since the last check
        self._overflows_seen = 0
        self._h = None
        if state_from is not None:                  # carry the optimiser state of parameters that kept their shape
            for k in range(15):
                if state_from.exp_avg[k].shape == self.exp_avg[k].shape:
                    self.exp_avg[k].copy_(state_from.exp_avg[k])
                    self.exp_avg_sq[k].copy_(state_from.exp_avg_sq[k])
                    self.steps[k] = state_from.steps[k]
            self.lr = list(state_from.lr)
        self._loss = torch.zeros((2,), dtype=torch.float64, device=self.dev)      # [sum of squared residuals, their mean]
        self._active = torch.zeros((1,), dtype=torch.int32, device=self.dev)
        self._gauge_on = True          # set by backward(); optimizer_step() before any backward() updates nothing but must not raise
        self._build()

    def _build(self):
        d, self._keep = _train_desc(self.field, self.params, self.exp_avg, self.exp_avg_sq, self.batch_size, self.max_samples, self.chunk_samples)
        out = C.c_void_p()
        with torch.cuda.device(self.dev):
            _lib.check(self.L.ngf_trainer_create(C.byref(d), C.byref(out), C.c_void_p(torch.cuda.current_stream().cuda_stream)))
        self.release()
        self._h = out
        self._shape_key = self._key()
        self._versions = self._param_versions()

    def _key(self):
        return _field_key(self.field, self.params)

    def _param_versions(self):
        """torch's in-place version counters of the six plane / gauge-plane tensors (the ones the trainer keeps packed copies of).
        The raw-pointer Adam kernel does not bump them -- and it writes the packed copy itself -- so a counter that moved since the
        last backward / optimizer_step means somebody else wrote the tensor in place: load_state_dict, a torch optimizer driving a
        ``frozen=`` parameter, ``.mul_()``.  (``.data`` writes and raw-pointer writes bump nothing: ``params_changed()`` is for them.)"""
        return tuple(int(p._version) for p in self.params[:6])

    def release(self):
        if getattr(self, "_h", None) is not None:
            self.L.ngf_trainer_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.release()
        except Exception:
            pass

    def params_changed(self):
        """Tell the trainer that plane / gauge-plane values were written by something other than ``optimizer_step`` (an in-place
        edit, ``load_state_dict``): its channel-last copies are rebuilt on the next ``backward``."""
        _lib.check(self.L.ngf_train_params_changed(self._h))

    def scratch_bytes(self) -> int:
        return int(self.L.ngf_trainer_bytes(self._h))

    def overflows(self):
        """(steps whose batch had more active samples than the trainer keeps activation rows for -- their updates were skipped on the
        device --, the row count).  Synchronises with the device."""
        cnt, rows = C.c_int64(0), C.c_int64(0)
        with torch.cuda.device(self.dev):
            _lib.check(self.L.ngf_train_overflow_count(self._h, C.byref(cnt), C.byref(rows), C.c_void_p(torch.cuda.current_stream().cuda_stream)))
        return int(cnt.value), int(rows.value)

    def check_rows(self, grow=True) -> int:
        """Speculative rows: report batches that did not fit since the last check (their steps changed nothing on the device) and, with
        ``grow``, rebuild the device trainer with twice the rows (the Adam state is the caller's tensors and stays).  The skipped steps are
        taken back out of the per-parameter step counters and the learning-rate decay (``optimizer_step`` advanced them on the host before
        the device's verdict was known).  Returns the number of new overflows.  Synchronises."""
        cnt, rows = self.overflows()
        new = cnt - self._overflows_seen
        self._overflows_seen = cnt
        self._since_check = 0
        if new > 0:
            # WHICH of the steps since the last check were skipped is not recorded (one counter on the device): take the most recent ones
            for idx, factor in self._applied[-new:]:
                for k in idx:
                    self.steps[k] -= 1
                self.lr = [x / factor for x in self.lr]
        self._applied = []
        if new > 0 and grow and self.chunk_samples < 0:
            import warnings
            warnings.warn(f"ngf_amd.train.Trainer: {new} step(s) had more active samples than the {rows} activation rows and were skipped; "
                          f"doubling the rows")
            self.release()
            self.chunk_samples = -min(self.batch_size * self.max_samples, 2 * rows)
            if -self.chunk_samples >= self.batch_size * self.max_samples:
                self.chunk_samples = 0
            self._build()
            self._overflows_seen = 0
        return new

    @property
    def last_active(self) -> int:
        """Active (weight > thr) samples of the last backward; reading it copies the count from the trainer and synchronises with the
        device (the copy is a launch of its own: it is not part of the step)."""
        n = getattr(self, "_last_n", 0)
        if n <= 0:
            return 0
        with torch.cuda.device(self.dev):
            st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
            _lib.check(self.L.ngf_train_get_active(self._h, n, self._active.data_ptr(), st))
        return int(self._active.item())

    # ---------------------------------------------------------------------------------------------------------------------
    @torch.no_grad()
    def backward(self, rays_train, rgb_train, N_samples=-1, white_bg=True, iteration=0, jitter=None, coin=None, keep_loss=False):
        """forward(is_train=True) + backward of the rgb MSE; returns the rgb loss as a 0-dim float64 device tensor.
        ``jitter`` [n] and ``coin`` (a float in [0,1)) replace torch.rand_like / torch.rand((1,)) for parity tests.

        ALIASING: by default the returned tensor is a VIEW of the trainer's persistent loss buffer (the step's last kernel writes the mean
        there; a torch op here would be one more launch per step) -- the next ``backward`` overwrites it in place.  Read it (``.item()``,
        ``float()``) before the next step, or pass ``keep_loss=True`` to get a fresh tensor per step as the reference's loss is
        (main.py:277), e.g. for a loss history or deferred ``.item()`` calls."""
        if self._key() != self._shape_key:
            raise RuntimeError("the field's parameters were re-allocated (up_sampling / shrink / load): build a new Trainer")
        if self._param_versions() != self._versions:      # an in-place write from outside: the packed copies are stale
            self.params_changed()
            self._versions = self._param_versions()
        rays = rays_train.to(device=self.dev, dtype=torch.float32).contiguous()
        tgt = rgb_train.to(device=self.dev, dtype=torch.float32).contiguous()
        n = rays.shape[0]
        if rays.dim() != 2 or rays.shape[1] != 6 or tuple(tgt.shape) != (n, 3):
            raise ValueError(f"rays_train must be [n,6] and rgb_train [n,3], got {tuple(rays.shape)} / {tuple(tgt.shape)}")
        S = int(N_samples) if N_samples > 0 else int(self.field.nSamples)
        if jitter is None:
            jitter = torch.rand((n,), device=self.dev)                 # FieldBase.py:129-130
        jitter = jitter.to(device=self.dev, dtype=torch.float32).contiguous()
        if not white_bg:
            c = float(torch.rand((1,))) if coin is None else float(coin)   # FieldBase.py:299
            white_bg = c < 0.5
        gauge_on = int(iteration >= self.field.gauge_start)
        # no host pointer for the active count: the call stays asynchronous (the colour kernels read the count on the device);
        # ``last_active`` fetches it lazily
        with torch.cuda.device(self.dev):
            st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
            _lib.check(self.L.ngf_train_backward2(self._h, rays.data_ptr(), tgt.data_ptr(), jitter.data_ptr(), n, S, int(bool(white_bg)), gauge_on,
                                                  self._loss.data_ptr(), int(self._loss.numel()), None, st))
        self._last_n = n
        self._gauge_on = gauge_on
        # a view: the division is done by the step's last kernel (a torch op here is one more launch per step)
        return self._loss[1].clone() if keep_loss else self._loss[1]

    @torch.no_grad()
    def gradient(self, which) -> torch.Tensor:
        """The gradient of one parameter (index or state_dict name) in its reference layout, after ``backward``.
        Planes exclude the L1 term (it is added inside the Adam kernel)."""
        k = PARAM_NAMES.index(which) if isinstance(which, str) else int(which)
        out = torch.empty_like(self.params[k])
        with torch.cuda.device(self.dev):
            _lib.check(self.L.ngf_train_get_grad(self._h, k, out.data_ptr(), C.c_void_p(torch.cuda.current_stream().cuda_stream)))
        return out

    @torch.no_grad()
    def optimizer_step(self):
        """optimizer.step() + the lr decay of main.py:298-299.  Gauge planes without a gradient (iteration <
        gauge_start) are skipped like torch.optim skips parameters whose .grad is None."""
        counts = (C.c_int32 * 15)()
        idx = []
        for k in range(15):
            if (3 <= k < 6 and not self._gauge_on) or k in self.frozen:
                continue                              # count 0 = skipped
            self.steps[k] += 1
            counts[k] = self.steps[k]
            idx.append(k)
        lrs = (C.c_float * 15)(*[float(x) for x in self.lr])
        with torch.cuda.device(self.dev):
            st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
            _lib.check(self.L.ngf_train_adam_all(self._h, counts, lrs, float(self.betas[0]), float(self.betas[1]), float(self.eps), float(self.l1), st))
        self.field.invalidate()      # parameters changed behind torch's back: the eval image is re-packed on the next render
        self.lr = [x * self.lr_factor for x in self.lr]
        if self.chunk_samples < 0:   # speculative rows: the device may have refused this update -- look every check_every steps
            self._applied.append((idx, self.lr_factor))
            self._since_check += 1
            if self._since_check >= self.check_every:
                self.check_rows()

    def step(self, rays_train, rgb_train, iteration, N_samples=-1, white_bg=True, jitter=None, coin=None, keep_loss=False):
        """One iteration of main.py:264-299.  Returns the rgb loss (0-dim float64 device tensor; ``.item()`` for PSNR) -- a view of the
        trainer's loss buffer that the next step overwrites unless ``keep_loss=True`` (see ``backward``)."""
        loss = self.backward(rays_train, rgb_train, N_samples, white_bg, iteration, jitter, coin, keep_loss=keep_loss)
        self.optimizer_step()
        return loss


class RenderGrad:
    """The device side of a DIFFERENTIABLE ``field(rays, is_train=True)`` (TriPlane/models/FieldBase.py:251-312 under autograd, as the reference's
    training loop uses it, TriPlane/main.py:272-296): a device trainer without optimiser state, driven in two calls -- ``forward`` renders the
    batch in training mode with the trainer's kernels and keeps its per-sample buffers, ``backward`` takes d loss / d rgb_map and returns the
    gradients of the fifteen parameters in their reference layouts.  ``Base.forward`` owns one per field (``_TrainRender`` below is the
    torch.autograd.Function around it); the caller's own loss, ``density_L1`` and ``torch.optim.Adam`` do the rest, unchanged.

    Activation rows (2.4 KB each), never a truncated gradient either way:
    * when the device has the memory for it (rows for EVERY (ray, sample) pair of the largest batch take less than a quarter of the free HBM: 8.7 GiB
      at the reference's 4096 x 884 on a 288 GB MI355X) the rows cover every pair and the forward NEVER waits for the host -- the colour kernels read
      the active count on the device.  Round 6: the host round trip in the middle of the forward left the GPU idle while Python caught up (the
      loop reads its loss every iteration, so every iteration starts with an empty queue); without it the reference's loop runs ~0.2 ms faster;
    * otherwise a third of the pairs (at least 262 144) and the active count read on the host in ``forward`` -- one stream synchronisation per step, like
      the reference's own ``rgb_mask.any()`` (FieldBase.py:291): a batch with more active samples than rows is worked through chunk by chunk.
    ``field.grad_rows = "all" | "third"`` forces one or the other."""

    def __init__(self, field, max_rays, max_samples):
        self.field = field
        self.dev = torch.device(field.device)
        if self.dev.type != "cuda":
            raise RuntimeError("a differentiable field(..., is_train=True) renders on the GPU only (device='cuda'); there is no CPU path")
        self.L = _lib.lib()
        _bind(self.L)
        self.params = _train_params(field)
        self.max_rays, self.max_samples = int(max_rays), int(max_samples)
        pairs = self.max_rays * self.max_samples
        rows = min(pairs, max(1 << 18, -(-pairs // 3)))
        self.chunk_samples = (rows + 15) // 16 * 16
        want = getattr(field, 'grad_rows', None)
        with torch.cuda.device(self.dev):
            free = torch.cuda.mem_get_info()[0]
        self.host_count = not (want == "all" or (want != "third" and pairs * 2600 * 4 <= free))
        if not self.host_count:
            self.chunk_samples = 0                  # ngf_train_desc: rows for every pair -- no host count, no chunks
        d, self._keep = _train_desc(field, self.params, None, None, self.max_rays, self.max_samples, self.chunk_samples)
        out = C.c_void_p()
        with torch.cuda.device(self.dev):
            _lib.check(self.L.ngf_trainer_create(C.byref(d), C.byref(out), C.c_void_p(torch.cuda.current_stream().cuda_stream)))
        self._h = out
        self.key = _field_key(field, self.params)
        self._versions = tuple(int(p._version) for p in self.params[:6])
        self._last_active, self._last_n = 0, 0
        self._active_dev = torch.zeros((1,), dtype=torch.int32, device=self.dev)
        from . import optim
        optim.register(field, self.params)          # ngf_amd.optim.Adam finds the engine behind a parameter (fused update, no re-pack)

    @property
    def last_active(self) -> int:
        """Active samples of the last forward (read from the device on demand when the forward did not bring it to the host)."""
        if not self.host_count and self._last_n > 0 and self._h is not None:
            with torch.cuda.device(self.dev):
                _lib.check(self.L.ngf_train_get_active(self._h, self._last_n, self._active_dev.data_ptr(), C.c_void_p(torch.cuda.current_stream().cuda_stream)))
            return int(self._active_dev.item())
        return self._last_active

    def fits(self, n, S, params=None):
        return n <= self.max_rays and S <= self.max_samples and self.key == _field_key(self.field, _train_params(self.field) if params is None else params)

    def release(self):
        if getattr(self, "_h", None) is not None:
            self.L.ngf_trainer_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.release()
        except Exception:
            pass

    def forward(self, rays, jitter, S, white_bg, gauge_on):
        """-> (rgb_map [n,3], depth_map [n], ticket).  The planes' packed copies follow torch's in-place version counters (optimizer.step(),
        load_state_dict); ``field.invalidate()`` marks them stale for writes that bypass the counters."""
        v = tuple(int(p._version) for p in self.params[:6])
        if v != self._versions or getattr(self.field, "_grad_stale", False):
            _lib.check(self.L.ngf_train_params_changed(self._h))
            self._versions = v
            self.field._grad_stale = False
        n = rays.shape[0]
        rgb = torch.empty((n, 3), device=self.dev, dtype=torch.float32)
        depth = torch.empty((n,), device=self.dev, dtype=torch.float32)
        n_active, ticket = C.c_int64(0), C.c_int64(0)
        with torch.cuda.device(self.dev):
            st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
            _lib.check(self.L.ngf_train_forward(self._h, rays.data_ptr(), jitter.data_ptr(), n, int(S), int(bool(white_bg)), int(bool(gauge_on)),
                                                rgb.data_ptr(), depth.data_ptr(), C.byref(n_active) if self.host_count else None, C.byref(ticket), st))
        self._last_active, self._last_n = int(n_active.value), n
        return rgb, depth, int(ticket.value)

    def backward(self, ticket, d_rgb, want):
        """``want[k]``: return parameter k's gradient (else None).  None = the ticket is stale (NGF_E_STALE: another forward used the buffers) -- re-run forward; every other failure raises."""
        with torch.cuda.device(self.dev):
            st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
            rc = self.L.ngf_train_backward_grad(self._h, int(ticket), d_rgb.data_ptr(), st)
            if rc == _lib.E_STALE:
                return None
            _lib.check(rc)              # anything else is an error of this call (a HIP / launch failure must not be retried away)
            grads = [torch.empty_like(self.params[k]) if want[k] else None for k in range(15)]
            ptrs = (C.c_void_p * 15)(*[None if g is None else g.data_ptr() for g in grads])
            _lib.check(self.L.ngf_train_get_grads(self._h, ptrs, st))          # one call: three tiled transposes, the gauge planes, one launch for the nine MLP tensors
        return grads


class _TrainRender(torch.autograd.Function):
    """``rgb_map, depth_map = field(rays, is_train=True)`` as one autograd node over the fifteen parameters.  depth_map is not differentiable
    (the reference computes it under torch.no_grad(), FieldBase.py:304-306); rays carry no gradient (the reference's are data)."""

    @staticmethod
    def forward(ctx, field, eng, rays, jitter, S, white_bg, gauge_on, *params):
        rgb, depth, ticket = eng.forward(rays, jitter, S, white_bg, gauge_on)
        ctx.field, ctx.engine, ctx.ticket = field, eng, ticket
        ctx.cfg = (int(S), bool(white_bg), bool(gauge_on))
        ctx.save_for_backward(rays, jitter, *params)          # saved tensors: autograd refuses a backward after an in-place write to any of them
        ctx.mark_non_differentiable(depth)
        return rgb, depth

    @staticmethod
    def backward(ctx, d_rgb, _d_depth):
        saved = ctx.saved_tensors
        rays, jitter = saved[0], saved[1]
        S, white_bg, gauge_on = ctx.cfg
        # the field's CURRENT engine: the one of the forward (the common case: its identity is the whole check), unless a larger batch or a
        # re-allocation had it rebuilt in between (then this batch is rendered again below)
        eng = ctx.engine
        if getattr(ctx.field, '_grad_engine', None) is not eng or eng._h is None:
            eng = ctx.field._render_grad_engine(rays.shape[0], S)
        if any(a.data_ptr() != b.data_ptr() or a.shape != b.shape for a, b in zip(saved[2:], eng.params)):
            raise RuntimeError("the field's parameter tensors were re-allocated (up_sampling / shrink / load) between forward and backward")
        d = d_rgb.to(dtype=torch.float32).contiguous()
        want = [bool(w) for w in ctx.needs_input_grad[7:]]
        if not gauge_on:                                   # compute_gauge was not evaluated (Field.py:58,73): the gauge planes are not in the graph
            want[3:6] = [False, False, False]
        grads = eng.backward(ctx.ticket, d, want) if eng is ctx.engine else None
        if grads is None:                                  # another forward went through the engine since (or the engine is a new one): render this batch again, then its backward
            _, _, ticket = eng.forward(rays, jitter, S, white_bg, gauge_on)
            grads = eng.backward(ticket, d, want)
            if grads is None:
                raise RuntimeError(_lib.lib().ngf_last_error().decode())
        return (None,) * 7 + tuple(grads)


class SimpleSampler:
    """TriPlane/utils.py:15-29: epoch-wise random permutation, ``batch`` indices per call."""

    def __init__(self, total, batch):
        self.total, self.batch, self.curr, self.ids = total, batch, total, None

    def nextids(self):
        self.curr += self.batch
        if self.curr + self.batch > self.total:
            self.ids = torch.LongTensor(np.random.permutation(self.total))
            self.curr = 0
        return self.ids[self.curr:self.curr + self.batch]


def fit(field, allrays, allrgbs, args, white_bg=True, on_iteration=None):
    """The optimisation loop of ``train`` (TriPlane/main.py:243-330) on top of ``Trainer`` -- no dataset, logging, checkpoint
    or visualisation code (those stay with the caller; ``on_iteration(iteration, rgb_loss)`` is the hook for them).

    ``args``: the reference's namespace (``opt.config_parser``): batch_size, n_iters, lr_init, lr_basis, lr_decay_iters,
    lr_decay_target_ratio, N_voxel_init, N_voxel_final, upsamp_list, update_AlphaMask_list, nSamples, step_ratio.
    Returns the per-iteration PSNR list.  Two behaviours of the reference that are easy to miss are kept:
    * after ``shrink`` (first alpha-mask update) the reference's optimizer still holds the pre-shrink plane tensors and is
      only rebuilt at the next up-sampling, so the cropped planes receive no updates in between -> the plane groups are
      frozen until the next rebuild here too;
    * every up-sampling resets the learning rates to their initial values and starts fresh Adam moments (main.py:316-324).
    """
    from . import geometry
    dev = torch.device(field.device)
    upsamp_list = list(args.upsamp_list or [])
    mask_list = list(args.update_AlphaMask_list or [])
    n_voxel_list = (torch.round(torch.exp(torch.linspace(np.log(args.N_voxel_init), np.log(args.N_voxel_final), len(upsamp_list)))).long()).tolist() \
        if upsamp_list else []
    reso_cur = [int(v) for v in field.gridSize]
    nSamples = min(int(args.nSamples), geometry.cal_n_samples(reso_cur, args.step_ratio))
    allrays, allrgbs = field.filtering_rays(allrays, allrgbs, bbox_only=True)
    sampler = SimpleSampler(allrays.shape[0], args.batch_size)

    def new_trainer(frozen=(), state_from=None, l1=L1_REG_WEIGHT):
        return Trainer(field, batch_size=args.batch_size, max_samples=nSamples, lr_init=args.lr_init, lr_basis=args.lr_basis,
                       lr_decay_iters=args.lr_decay_iters, lr_decay_target_ratio=args.lr_decay_target_ratio, n_iters=args.n_iters,
                       L1_reg_weight=l1, frozen=frozen, state_from=state_from)

    l1 = L1_REG_WEIGHT
    trainer = new_trainer()
    PSNRs = []
    for iteration in range(args.n_iters):
        ids = sampler.nextids()
        rays_train, rgb_train = allrays[ids].to(dev), allrgbs[ids].to(dev)
        rgb_loss = trainer.step(rays_train, rgb_train, iteration, N_samples=nSamples, white_bg=white_bg).item()
        while rgb_loss != rgb_loss and trainer.chunk_samples < 0 and trainer.check_rows() > 0:
            # speculative rows (opt-in): the batch did not fit, the device skipped its update and marked the loss NaN; check_rows doubled the
            # rows and took the step back out of the counters -- take it again
            rgb_loss = trainer.step(rays_train, rgb_train, iteration, N_samples=nSamples, white_bg=white_bg).item()
        PSNRs.append(-10.0 * np.log(rgb_loss) / np.log(10.0))
        if on_iteration is not None:
            on_iteration(iteration, rgb_loss)
        if iteration in mask_list:
            new_aabb = field.updateAlphaMask((256, 256, 256))
            if iteration == mask_list[0]:
                field.shrink(new_aabb)
                l1 = 4e-5                                                    # main.py:306
                allrays, allrgbs = field.filtering_rays(allrays, allrgbs)
                sampler = SimpleSampler(allrgbs.shape[0], args.batch_size)
                old = trainer
                trainer = new_trainer(frozen=(0, 1, 2), state_from=old, l1=l1)
                old.release()
            else:
                old = trainer
                trainer = new_trainer(frozen=old.frozen, state_from=old, l1=l1)     # new mask -> new device image
                old.release()
        if iteration in upsamp_list:
            reso_cur = geometry.N_to_reso(n_voxel_list.pop(0), field.aabb.detach().cpu().numpy())
            nSamples = min(int(args.nSamples), geometry.cal_n_samples(reso_cur, args.step_ratio))
            field.up_sampling(reso_cur)
            trainer.release()
            trainer = new_trainer(l1=l1)                                       # fresh Adam, initial learning rates
    trainer.release()
    return PSNRs
