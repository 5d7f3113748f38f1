"""``ngf_amd.optim.Adam`` -- torch.optim.Adam for the reference's own training loop on the drop-in field (TriPlane/main.py:234-242,294-302):

    grad_vars = field.get_optparam_groups(args.lr_init, args.lr_basis)
    optimizer = Adam(grad_vars, betas=(0.9, 0.99))          # was: torch.optim.Adam(grad_vars, betas=(0.9, 0.99))
    ...
    optimizer.zero_grad(); total_loss.backward(); optimizer.step()
    for param_group in optimizer.param_groups: param_group['lr'] = param_group['lr'] * lr_factor

Same constructor, same ``param_groups`` / ``state`` / ``state_dict()`` (per parameter ``step``, ``exp_avg``, ``exp_avg_sq`` in the parameter's
layout, so a checkpoint moves between this class and torch's), same arithmetic -- but the update of a TriPlane field's fifteen parameters is ONE
C-ABI call (``ngf_train_adam_ext``, include/ngf.h): a pass per plane that reads ``p.grad`` and the moments, writes the parameter, the moments and
the differentiable forward's channel-last copy of the plane.  What leaves the step compared with torch.optim.Adam on the same field: ~20 foreach
launches over 52 MB of parameters and, at the next ``field(rays, is_train=True)``, the re-pack of three planes (torch's in-place update makes the
copies stale).  Parameters that do not belong to a field with a live differentiable engine -- and groups that ask for amsgrad, weight decay,
maximize, capturable or differentiable -- take torch's own path (``super().step()``), so the class is a superset, not a special case.
There is no CPU path for the fused part; on CPU tensors everything is torch's."""
from __future__ import annotations

import ctypes as C
import weakref

import torch

from . import _lib

# id(parameter) -> (weak parameter, weak field, index in train.PARAM_NAMES); filled by train.RenderGrad when a field's differentiable engine is
# built.  (Keyed by id: a tensor cannot key a WeakKeyDictionary -- its == is elementwise.)
_OWNERS = {}


def register(field, params):
    for key in [i for i, (wp, _, _) in _OWNERS.items() if wp() is None]:
        del _OWNERS[key]
    for k, p in enumerate(params):
        _OWNERS[id(p)] = (weakref.ref(p), weakref.ref(field), k)


class Adam(torch.optim.Adam):
    def _plan(self):
        """[(field, engine, [(group, param, k)])] for the parameters the fused call can take today; everything else is torch's."""
        by_field = {}
        for g in self.param_groups:
            if g.get('amsgrad') or g.get('maximize') or g.get('weight_decay', 0) != 0 or g.get('capturable') or g.get('differentiable'):
                continue
            for p in g['params']:
                if p.grad is None or not p.is_cuda:
                    continue
                own = _OWNERS.get(id(p))
                if own is None or own[0]() is not p:
                    continue
                field, k = own[1](), own[2]
                eng = None if field is None else getattr(field, '_grad_engine', None)
                if eng is None or eng._h is None or eng.params[k] is not p:
                    continue
                g_ = p.grad
                if g_.is_sparse or g_.dtype != torch.float32 or not g_.is_contiguous() or g_.shape != p.shape:
                    continue
                by_field.setdefault(id(field), (field, eng, []))[2].append((g, p, k))
        return list(by_field.values())

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        taken = []
        for field, eng, items in self._plan():
            grads, m, v = (C.c_void_p * 15)(), (C.c_void_p * 15)(), (C.c_void_p * 15)()
            counts, lrs = (C.c_int32 * 15)(), (C.c_float * 15)()
            beta = None
            ok = True
            for g, p, k in items:          # one call takes one pair of betas / eps: the reference's groups share them (main.py:242)
                b = (float(g['betas'][0]), float(g['betas'][1]), float(g['eps']))
                if beta is None:
                    beta = b
                ok = ok and b == beta and not isinstance(g['lr'], torch.Tensor)
            if not ok:
                continue
            for g, p, k in items:
                st = self.state[p]
                if len(st) == 0:                        # torch.optim.Adam._init_group
                    st['step'] = torch.tensor(0.0, dtype=torch.float32)
                    st['exp_avg'] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st['exp_avg_sq'] = torch.zeros_like(p, memory_format=torch.preserve_format)
                if not (st['exp_avg'].is_contiguous() and st['exp_avg_sq'].is_contiguous()):
                    ok = False
                    break
            if not ok:
                continue
            for g, p, k in items:
                st = self.state[p]
                st['step'] += 1
                counts[k] = int(st['step'])
                lrs[k] = float(g['lr'])
                grads[k], m[k], v[k] = p.grad.data_ptr(), st['exp_avg'].data_ptr(), st['exp_avg_sq'].data_ptr()
                taken.append(p)
            with torch.cuda.device(eng.dev):
                _lib.check(eng.L.ngf_train_adam_ext(eng._h, grads, m, v, counts, lrs, beta[0], beta[1], beta[2],
                                                    C.c_void_p(torch.cuda.current_stream().cuda_stream)))
            # the parameters changed behind torch's version counters: the eval image is rebuilt at the next render; the differentiable
            # engine's packed planes were written by the call itself and stay current
            field._handle_key = None
        if taken:
            rest = any(p.grad is not None and not any(p is q for q in taken) for g in self.param_groups for p in g['params'])
            if rest:
                held = [(p, p.grad) for p in taken]
                for p in taken:
                    p.grad = None
                try:
                    super().step()
                finally:
                    for p, gr in held:
                        p.grad = gr
        else:
            super().step()
        return loss
