"""Operator surface of the reference, on the MI355X backend.

Mirrors, name for name and argument for argument:
  * the pybind modules `hgnnaggr` (HyperGsys/source/hgnnaggr/hgnnaggr.cc:122-151)
    and `unignnaggr` (HyperGsys/source/unignnaggr/unignnaggr.cc:81-102) --
    exposed as `hypergef_amd.hgnnaggr` / `hypergef_amd.unignnaggr` and, after
    `hypergef_amd.install_dropin()`, as top-level `hgnnaggr` / `unignnaggr`;
  * the Python wrappers `HGNNAggr`, `UniGNNConvdeg`, `UniGNNConv`
    (HyperGsys/source/python/hgnnaggr.py:6-7, unignnconv.py:6-10).

The four schedule tensors (`balan_key, balan_row, group_st, group_ed`) stay in
the signatures.  The default kernels derive their own wave64 schedule from
`csrptr_t` / `indices_t` (cached per hypergraph); with
`set_variant("push_groups")` the reference's own scheme runs on exactly the
tasks those tensors describe.

Backward: the reference returns `forward(grad_out)` for every sum operator
(hgnnaggr.cc:51-64, unignnaggr.cc:36-49, 65-78).  That equals the true adjoint
only when degV is absent; `set_backward("adjoint")` selects
H diag(degE W) H^T diag(degV) grad instead.  Default: "reference".
"""
import torch

from . import _lib
from .plan import (cached_plan, linear_fusion_pays, linear_rows, linear_supported, linear_wgrad,
                   wgrad_supported, _check_feat, _check_index, _ptr, _stream_handle)

import contextlib
import dataclasses
import os as _os
import threading


@dataclasses.dataclass(frozen=True)
class Options:
    """How one operator call runs: kernel family, backward rule, linear folding.

    variant:     'auto' | 'pull' | 'fused' | 'push_atomic' (one task per hyperedge) | 'push_groups'
                 (the caller's group_* tensors drive the push kernel).
    backward:    'reference' (forward(grad_out), hgnnaggr.cc:51-64) | 'adjoint' (the exact transpose).
    fuse_linear: 'auto' folds a layer's projection into the aggregation where that is faster
                 (plan.linear_fusion_pays), 'always' wherever the kernel takes the widths, 'never' not.
    linear_math: 'f32' = the folded projection on fp32 MFMA; 'bf16x6' = at F_in = 128 each fp32 product as six bf16
                 products on the bf16 MFMA (HG_LIN_BF16X6, include/hg_aggr.h): the same error bound, 3/8 of the matrix-pipe
                 cycles.  Forward of the fused layer only; every other width and kernel stays fp32 MFMA.

    Every operator resolves its options per call -- an explicit `options=` argument, else the innermost
    `with ops.options(...)` block of the calling thread, else the process defaults -- and an autograd node keeps
    the options of its forward for its backward, whichever thread runs it.  Two models in one process can
    therefore differ; nothing is shared but the defaults."""
    variant: str = "auto"
    backward: str = "reference"
    fuse_linear: str = "auto"
    linear_math: str = "f32"

    _CHOICES = {"variant": ("auto", "pull", "fused", "push_atomic", "push_groups"),
                "backward": ("reference", "adjoint"), "fuse_linear": ("auto", "always", "never"),
                "linear_math": ("f32", "bf16x6")}

    def __post_init__(self):
        for k, allowed in Options._CHOICES.items():
            if getattr(self, k) not in allowed:
                raise ValueError("%s must be one of %s, got %r" % (k, ", ".join(allowed), getattr(self, k)))

    def replace(self, **kw):
        return dataclasses.replace(self, **kw)


_DEFAULTS_LOCK = threading.Lock()
_DEFAULTS = Options(fuse_linear=_os.environ["HG_FUSE_LINEAR"]
                    if _os.environ.get("HG_FUSE_LINEAR") in ("auto", "always", "never") else "auto",  # A/B runs of the drivers
                    linear_math=_os.environ["HG_LINEAR_MATH"]
                    if _os.environ.get("HG_LINEAR_MATH") in ("f32", "bf16x6") else "f32")
_TLS = threading.local()


def current_options():
    """The options a call made here, now, without an explicit `options=` would run with."""
    stack = getattr(_TLS, "stack", None)
    return stack[-1] if stack else _DEFAULTS


@contextlib.contextmanager
def options(**kw):
    """`with ops.options(variant="pull", backward="adjoint"):` -- per-thread, nestable overrides."""
    stack = getattr(_TLS, "stack", None)
    if stack is None:
        stack = _TLS.stack = []
    stack.append(current_options().replace(**kw))
    try:
        yield stack[-1]
    finally:
        stack.pop()


def _opt(o):
    if o is None:
        return current_options()
    if not isinstance(o, Options):
        raise TypeError("options must be an ops.Options")
    return o


def _set_default(**kw):
    global _DEFAULTS
    with _DEFAULTS_LOCK:
        _DEFAULTS = _DEFAULTS.replace(**kw)


def set_variant(name):
    """Process default of Options.variant (see there)."""
    _set_default(variant=name)


def set_backward(mode):
    """Process default of Options.backward."""
    _set_default(backward=mode)


def set_fuse_linear(mode):
    """Process default of Options.fuse_linear."""
    _set_default(fuse_linear=mode)


def set_linear_math(mode):
    """Process default of Options.linear_math."""
    _set_default(linear_math=mode)


def _flat(t):
    # degE / degV / W are read as flat arrays: [M] and [M,1] both valid
    return None if t is None else t.reshape(-1)


def _forward(sched, csrptr_t, indices_t, node_feat, degE, degV, W, opt):
    _check_feat(node_feat, "node_feat")
    _check_index(csrptr_t, "csrptr_t")
    _check_index(indices_t, "indices_t")
    if node_feat.dim() != 2:
        raise ValueError("node_feat must be [N, F]")
    N, F = node_feat.shape
    degE, degV, W = _flat(degE), _flat(degV), _flat(W)
    variant = opt.variant
    if variant == "push_groups":
        key, row, st, ed = sched
        for n, t in (("balan_key", key), ("balan_row", row), ("group_st", st), ("group_ed", ed)):
            _check_index(t, n)
        M = csrptr_t.numel() - 1
        for name, t, n in (("degE", degE, M), ("degV", degV, N), ("W", W, M)):
            if t is not None:
                _check_feat(t, name, device=node_feat.device)
                if t.numel() != n:
                    raise ValueError("%s must have %d elements" % (name, n))
        Y = torch.empty((N, F), dtype=torch.float32, device=node_feat.device)
        with torch.cuda.device(node_feat.device):
            _lib.check(_lib.lib().hg_aggr_push_groups_f32(
                N, M, F, row.numel(), _ptr(key), _ptr(row), _ptr(st), _ptr(ed),
                _ptr(csrptr_t), _ptr(indices_t), _ptr(node_feat), _ptr(degE), _ptr(degV), _ptr(W),
                _ptr(Y), _stream_handle(node_feat.device)))
        return Y
    plan = cached_plan(N, csrptr_t, indices_t)
    return plan.aggregate(csrptr_t, indices_t, node_feat, degE, degV, W, variant=variant)


class _SumAggr(torch.autograd.Function):
    """One autograd node for all three sum operators (degE/degV/W optional)."""

    @staticmethod
    def forward(ctx, balan_key, balan_row, group_st, group_ed, csrptr_t, indices_t, node_feat,
                degE, degV, W, opt=None):
        opt = ctx.opt = _opt(opt)
        out = _forward((balan_key, balan_row, group_st, group_ed), csrptr_t, indices_t, node_feat,
                       degE, degV, W, opt)
        # The reference saves its inputs the same way (hgnnaggr.cc:44-46); node_feat is not needed
        # (the operator is linear).  save_for_backward makes autograd raise if one of them is
        # modified in place between forward and backward.
        ctx.save_for_backward(balan_key, balan_row, group_st, group_ed, csrptr_t, indices_t, degE, degV, W)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        grad_out = grad_out.contiguous()
        balan_key, balan_row, group_st, group_ed, csrptr_t, indices_t, degE, degV, W = ctx.saved_tensors
        sched = (balan_key, balan_row, group_st, group_ed)
        if ctx.opt.backward == "reference" or degV is None:
            g = _forward(sched, csrptr_t, indices_t, grad_out, degE, degV, W, ctx.opt)
        else:
            g = _forward(sched, csrptr_t, indices_t, grad_out * degV.reshape(-1, 1), degE, None, W, ctx.opt)
        return (None,) * 6 + (g, None, None, None, None)


def _rows_times(A, B, mode="auto"):
    """A . B for tall-skinny A [N, K], B [K, F]: the library's MFMA rows kernel where it takes the
    widths (1.4-1.5x rocBLAS at K <= 64), torch otherwise.  Backward-pass GEMMs use it."""
    K, F = B.shape
    if mode != "never" and A.is_cuda and linear_supported(K, F) and A.shape[0] >= 4096:
        return linear_rows(A.contiguous(), B.t().contiguous())
    return A @ B


def _pad16(t):
    pad = (-t.shape[1]) % 16
    return t if pad == 0 else torch.nn.functional.pad(t, (0, pad))


def _wgrad(A, B, mode="auto"):
    """A^T . B over the vertices (the linear's weight gradient): the library's streaming MFMA kernel
    where it takes the widths -- rocBLAS needs 1.2 ms for [64 x 693 k] x [693 k x 64], 17x the time of
    reading the operands -- torch otherwise."""
    if mode != "never" and A.is_cuda and A.shape[0] >= 4096:
        Fa, Fb = A.shape[1], B.shape[1]
        if wgrad_supported(Fa, Fb):
            return linear_wgrad(A.contiguous(), B.contiguous())
        Pa, Pb = Fa + (-Fa) % 16, Fb + (-Fb) % 16
        if wgrad_supported(Pa, Pb):  # e.g. the class-count layer: pad to the next 16 columns, cut the result
            return linear_wgrad(_pad16(A).contiguous(), _pad16(B).contiguous())[:Fa, :Fb]
    return A.t() @ B


class _LinearFn(torch.autograd.Function):
    """torch.nn.functional.linear with this library's weight-gradient kernel in the backward
    (x^T-style contraction over the vertices; see _wgrad)."""

    @staticmethod
    def forward(ctx, x, weight, bias, opt=None):
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        ctx.mode = _opt(opt).fuse_linear  # the owning module's pinned Options, else the caller's (thread, process)
        return torch.nn.functional.linear(x, weight, bias)

    @staticmethod
    def backward(ctx, grad):
        x, weight = ctx.saved_tensors
        grad = grad.contiguous()
        gx = _rows_times(grad, weight, ctx.mode) if ctx.needs_input_grad[0] else None
        gw = _wgrad(grad, x, ctx.mode) if ctx.needs_input_grad[1] else None
        gb = grad.sum(0) if (ctx.has_bias and ctx.needs_input_grad[2]) else None
        return gx, gw, gb, None


class Linear(torch.nn.Linear):
    """Drop-in nn.Linear (same parameters, same state_dict) for the [N, F] activations of these
    models: identical forward, weight gradient on hg_linear_wgrad_f32.  `options` (an ops.Options, or None for the
    caller's) pins what the backward may use, like the conv modules' own: a model built with
    Options(fuse_linear="never") keeps its linears off the library's MFMA kernels too."""

    def __init__(self, in_features, out_features, bias=True, device=None, dtype=None, options=None):
        super().__init__(in_features, out_features, bias=bias, device=device, dtype=dtype)
        if options is not None and not isinstance(options, Options):
            raise TypeError("options must be an ops.Options")
        self.options = options

    def forward(self, x):
        if x.dim() == 2 and x.is_cuda and x.dtype == torch.float32:
            return _LinearFn.apply(x, self.weight, self.bias, self.options)
        return super().forward(x)


class _SumAggrLinear(torch.autograd.Function):
    """Aggr(X . Wlin^T) as one node: the layer's bias-free nn.Linear followed by the sum
    aggregation (HyperGsysHGNN.forward, model/ugsys/hgnn.py:22-23; HyperGsysUinGINConv.forward,
    unigin.py:20-21).  Forward runs hg_aggr_linear_f32 -- aggregate the F_in-wide rows, multiply
    each finished row by Wlin^T on the matrix cores -- when the widths allow, else the two steps.
    Backward is the two-step one: dZ = aggregation backward of grad_out (reference or adjoint
    mode, as _SumAggr), dX = dZ . Wlin, dWlin = dZ^T . X."""

    @staticmethod
    def forward(ctx, csrptr_t, indices_t, node_feat, weight, degE, degV, W, opt=None):
        opt = ctx.opt = _opt(opt)
        _check_feat(node_feat, "node_feat")
        _check_feat(weight, "weight", device=node_feat.device)
        _check_index(csrptr_t, "csrptr_t")
        _check_index(indices_t, "indices_t")
        if node_feat.dim() != 2 or weight.dim() != 2 or weight.shape[1] != node_feat.shape[1]:
            raise ValueError("node_feat must be [N, F_in] and weight [F_out, F_in]")
        degE, degV, W = _flat(degE), _flat(degV), _flat(W)
        N, F_in = node_feat.shape
        F_out = weight.shape[0]
        variant = opt.variant
        plan = cached_plan(N, csrptr_t, indices_t)
        mode = opt.fuse_linear
        fuse = (mode == "always" and linear_supported(F_in, F_out)) or \
               (mode == "auto" and linear_fusion_pays(F_in, F_out))
        if fuse and variant in ("auto", "pull", "fused"):
            out = plan.aggregate_linear(csrptr_t, indices_t, node_feat, weight.detach().contiguous(),
                                        degE, degV, W, variant=variant, math=opt.linear_math)
        else:  # project, then aggregate at F_out (own MFMA rows kernel where it takes the widths)
            wd = weight.detach().contiguous()
            Z = linear_rows(node_feat, wd) if linear_supported(F_in, F_out) and mode != "never" \
                else torch.nn.functional.linear(node_feat, wd)
            out = _SumAggrLinear._aggr(csrptr_t, indices_t, Z, degE, degV, W, opt)
        ctx.save_for_backward(node_feat, weight, csrptr_t, indices_t, degE, degV, W)
        return out

    @staticmethod
    def _aggr(csrptr_t, indices_t, feat, degE, degV, W, opt):
        # this operator has no group_* tensors: "push_groups" falls back to the plan's own schedule
        variant = opt.variant if opt.variant != "push_groups" else "auto"
        plan = cached_plan(feat.shape[0], csrptr_t, indices_t)
        return plan.aggregate(csrptr_t, indices_t, feat.contiguous(), degE, degV, W, variant=variant)

    @staticmethod
    def backward(ctx, grad_out):
        grad_out = grad_out.contiguous()
        node_feat, weight, csrptr_t, indices_t, degE, degV, W = ctx.saved_tensors
        opt = ctx.opt
        if opt.backward == "reference" or degV is None:
            dZ = _SumAggrLinear._aggr(csrptr_t, indices_t, grad_out, degE, degV, W, opt)
        else:
            dZ = _SumAggrLinear._aggr(csrptr_t, indices_t, grad_out * degV.reshape(-1, 1), degE, None, W, opt)
        gx = _rows_times(dZ, weight, opt.fuse_linear) if ctx.needs_input_grad[2] else None
        gw = _wgrad(dZ, node_feat, opt.fuse_linear) if ctx.needs_input_grad[3] else None
        return None, None, gx, gw, None, None, None, None


class _AggrResLinear(torch.autograd.Function):
    """Y = act((ca * Aggr(X) + cb * R) . M^T): a whole UniGNN layer as one node
    (hg_aggr_linear_res_f32).  UniGCNII (model/ugsys/unigcnii.py:19-21 + the relu of model/gnn.py:199):
    ca = 1 - alpha, cb = alpha, R = X0, M = (1 - beta) I + beta W.  UniGIN (unigin.py:20-22):
    ca = 1, cb = 1 + eps, R = X, M = W.  Falls back to the same formula in torch ops around the
    plain aggregation where the MFMA epilogue does not take the widths.  Backward: dP = dY (masked by
    the relu), dT = dP . M, dM = dP^T . T, dX = ca * (aggregation backward of dT), dR = cb * dT,
    dcb = <dT, R> (cb may be a tensor, e.g. 1 + eps)."""

    @staticmethod
    def forward(ctx, csrptr_t, indices_t, node_feat, M, R, cb, degE, degV, W, ca, relu, need_t, opt=None):
        opt = ctx.opt = _opt(opt)
        _check_feat(node_feat, "node_feat")
        _check_index(csrptr_t, "csrptr_t")
        _check_index(indices_t, "indices_t")
        degE, degV, W = _flat(degE), _flat(degV), _flat(W)
        N, F_in = node_feat.shape
        F_out = M.shape[0]
        # cb is a Python float except where it is learned (UniGIN's 1 + eps): then it stays on the device -- the
        # kernel reads it there (hg_aggr_linear_res_dev_f32), nothing is read back, and the step can be captured
        if R is None:
            cbf = 0.0
        elif isinstance(cb, torch.Tensor):
            cbf = cb.detach().to(torch.float32).reshape(1).contiguous()
        else:
            cbf = float(cb)
        variant = opt.variant if opt.variant != "push_groups" else "auto"
        Md = M.detach().contiguous()
        Rd = None if R is None else R.detach().contiguous()
        mode = opt.fuse_linear
        fuse = (mode == "always" and linear_supported(F_in, F_out)) or \
               (mode == "auto" and linear_fusion_pays(F_in, F_out))
        if fuse and variant in ("auto", "pull", "fused"):
            plan = cached_plan(N, csrptr_t, indices_t)
            T = torch.empty_like(node_feat) if need_t else None
            out = plan.aggregate_linear(csrptr_t, indices_t, node_feat.detach(), Md, degE, degV, W, variant=variant,
                                        residual=Rd, ca=ca, cb=cbf, relu=relu, t_out=T, math=opt.linear_math)
        else:
            T = _SumAggrLinear._aggr(csrptr_t, indices_t, node_feat.detach(), degE, degV, W, opt) * ca
            if Rd is not None:
                T = T + Rd * cbf
            out = T @ Md.t()
            if relu:
                out = torch.relu(out)
        ctx.consts = (ca, cbf, relu)
        ctx.save_for_backward(M, R, T, out, csrptr_t, indices_t, degE, degV, W)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        M, R, T, out, csrptr_t, indices_t, degE, degV, W = ctx.saved_tensors
        ca, cbf, relu = ctx.consts
        dP = grad_out.contiguous()
        if relu:
            dP = torch.ops.aten.threshold_backward(dP, out, 0.0)  # relu's own backward: one vectorised kernel
        opt = ctx.opt
        need_gcb = R is not None and ctx.needs_input_grad[5]
        # dX wants ca * (dP . M): the factor rides in the [F_out, F_in] matrix (one tiny kernel) instead of an [N, F] pass over
        # the product, unless the unscaled product is needed for d cb (UniGIN, where ca = 1 anyway)
        fold = ca != 1.0 and not need_gcb
        dT = _rows_times(dP, M * ca if fold else M, opt.fuse_linear)
        gM = _wgrad(dP, T, opt.fuse_linear) if ctx.needs_input_grad[3] else None
        gx = None
        if ctx.needs_input_grad[2]:
            g_in = dT if (fold or ca == 1.0) else dT * ca  # UniGIN: ca = 1 -- no [N, F] kernel for a multiplication by one
            if opt.backward == "reference" or degV is None:
                gx = _SumAggrLinear._aggr(csrptr_t, indices_t, g_in, degE, degV, W, opt)
            else:
                gx = _SumAggrLinear._aggr(csrptr_t, indices_t, g_in * degV.reshape(-1, 1), degE, None, W, opt)
        gR = None
        if R is not None and ctx.needs_input_grad[4]:
            cbr = cbf / ca if fold else cbf  # dT already carries ca
            gR = dT if (not isinstance(cbr, torch.Tensor) and cbr == 1.0) else dT * cbr
        gcb = (dT * R).sum() if need_gcb else None
        return None, None, gx, gM, gR, gcb, None, None, None, None, None, None, None


def aggr_res_linear(csrptr_t, indices_t, node_feat, M, residual=None, ca=1.0, cb=0.0, degE=None, degV=None, W=None,
                    relu=False, options=None):
    """act((ca * Aggr(node_feat) + cb * residual) . M^T) in one pass where the widths allow
    (include/hg_aggr.h, hg_aggr_linear_res_f32).  cb may be a tensor (its gradient is returned)."""
    # a Python-number cb goes through as it is (no host-to-device copy, no sync to read it back)
    cb_a = cb if isinstance(cb, torch.Tensor) else float(cb)
    # T (the rows before the product) is written only if a backward pass can follow: Function.forward
    # itself always runs with grad mode off and needs_input_grad set, so decide here
    need_t = torch.is_grad_enabled() and any(
        isinstance(t, torch.Tensor) and t.requires_grad for t in (node_feat, M, residual, cb_a))
    return _AggrResLinear.apply(csrptr_t, indices_t, node_feat, M, residual, cb_a, degE, degV, W, float(ca),
                                bool(relu), need_t, _opt(options))


def hgnnaggr_linear(csrptr_t, indices_t, node_feat, weight, degE=None, degV=None, W=None, options=None):
    """Aggr(node_feat . weight^T) with the projection folded into the aggregation
    (include/hg_aggr.h, hg_aggr_linear_f32).  degE / degV / W optional: all three = hgnnaggr,
    degE + degV = unignnaggrdeg, none = unignnaggr."""
    return _SumAggrLinear.apply(csrptr_t, indices_t, node_feat, weight, degE, degV, W, _opt(options))


# ---- module `hgnnaggr` (hgnnaggr.cc:122-151) ---------------------------------

def hgnnaggr(balan_key, balan_row, group_st, group_ed, csrptr_t, indices_t, node_feat, degE, degV, W, options=None):
    """hgnnaggr with fused degE and degV.  (`options`: this backend's per-call Options; the reference's ten
    positional arguments are unchanged.)"""
    return _sum_aggr(balan_key, balan_row, group_st, group_ed, csrptr_t, indices_t, node_feat, degE, degV, W, options)


def _sum_aggr(balan_key, balan_row, group_st, group_ed, csrptr_t, indices_t, node_feat, degE, degV, W, options):
    opt = _opt(options)
    # nothing to differentiate (inference, or a feature tensor outside the graph): the operator itself, without the
    # autograd node -- a third of the host cost of a launch-bound call
    if not (torch.is_grad_enabled() and isinstance(node_feat, torch.Tensor) and node_feat.requires_grad):
        return _forward((balan_key, balan_row, group_st, group_ed), csrptr_t, indices_t, node_feat, degE, degV, W, opt)
    return _SumAggr.apply(balan_key, balan_row, group_st, group_ed, csrptr_t, indices_t, node_feat, degE, degV, W, opt)


def _edge_sizes(csrptr_t):
    return (csrptr_t[1:] - csrptr_t[:-1]).to(torch.float32)


class _MeanF1(torch.autograd.Function):
    """first hop = mean (HGNNAggr_MeanF1, hgnnaggr.cc:66-90; kernels hgnnaggr_cuda.cu:86-142):
    the hyperedge sum is scaled by degE*W/|e| (one factor, as the reference computes it), the
    second hop is the ordinary one.  Backward = the same operator on grad_out (the reference's
    backward kernel differs only in where it divides by |e|)."""

    @staticmethod
    def forward(ctx, csrptr_t, indices_t, node_feat, degE, degV, W):
        _check_index(csrptr_t, "csrptr_t")
        s = _flat(degE) * _flat(W) / _edge_sizes(csrptr_t)
        s = torch.where(torch.isfinite(s), s, torch.zeros_like(s))  # empty hyperedge: never read
        ctx.save_for_backward(csrptr_t, indices_t, s, _flat(degV))
        plan = cached_plan(node_feat.shape[0], csrptr_t, indices_t)
        return plan.aggregate(csrptr_t, indices_t, node_feat, s, _flat(degV), None)

    @staticmethod
    def backward(ctx, grad_out):
        csrptr_t, indices_t, s, degV = ctx.saved_tensors
        plan = cached_plan(grad_out.shape[0], csrptr_t, indices_t)
        g = plan.aggregate(csrptr_t, indices_t, grad_out.contiguous(), s, degV, None)
        return None, None, g, None, None, None


class _MaxF1(torch.autograd.Function):
    """first hop = per-column max with arg-max table (HGNNAggr_MaxF1, hgnnaggr.cc:92-120;
    kernels hgnnaggr_cuda.cu:144-208).  Loop bounds use M, not the reference's N (defect D2)."""

    @staticmethod
    def forward(ctx, csrptr_t, indices_t, node_feat, degE, degV, W):
        _check_feat(node_feat, "node_feat")
        _check_index(csrptr_t, "csrptr_t")
        _check_index(indices_t, "indices_t")
        N, F = node_feat.shape
        M = csrptr_t.numel() - 1
        degE, degV, W = _flat(degE), _flat(degV), _flat(W)
        dev = node_feat.device
        Xe = torch.empty((M, F), dtype=torch.float32, device=dev)
        record = torch.empty((M, F), dtype=torch.int32, device=dev)
        with torch.cuda.device(dev):
            _lib.check(_lib.lib().hg_gather_max_f32(M, F, _ptr(csrptr_t), _ptr(indices_t), _ptr(node_feat),
                                                    _ptr(degE), _ptr(W), _ptr(Xe), _ptr(record),
                                                    _stream_handle(dev)))
        plan = cached_plan(N, csrptr_t, indices_t)
        out = plan.gather_rows(1, csrptr_t, indices_t, Xe, degV, None)
        ctx.save_for_backward(csrptr_t, indices_t, degE, degV, W, record)
        ctx.mark_non_differentiable(record)
        return out, record

    @staticmethod
    def backward(ctx, grad_out, _grad_record):
        csrptr_t, indices_t, degE, degV, W, record = ctx.saved_tensors
        grad_out = grad_out.contiguous()
        N, F = grad_out.shape
        M = csrptr_t.numel() - 1
        plan = cached_plan(N, csrptr_t, indices_t)
        T = plan.gather_rows(0, csrptr_t, indices_t, grad_out, degE, W)  # (sum grad) * degE * W
        g = torch.empty((N, F), dtype=torch.float32, device=grad_out.device)
        with torch.cuda.device(grad_out.device):
            _lib.check(_lib.lib().hg_scatter_record_f32(N, M, F, _ptr(T), _ptr(record), _ptr(degV), _ptr(g),
                                                        _stream_handle(grad_out.device)))
        return None, None, g, None, None, None


def hgnnaggr_mean(csrptr_t, indices_t, node_feat, degE, degV, W):
    """hgnnaggr with f1 mean (hgnnaggr.cc:131-136)."""
    return _MeanF1.apply(csrptr_t, indices_t, node_feat, degE, degV, W)


def hgnnaggr_max(csrptr_t, indices_t, node_feat, degE, degV, W):
    """hgnnaggr with f1 max (hgnnaggr.cc:138-144): returns [out, record_table]."""
    return list(_MaxF1.apply(csrptr_t, indices_t, node_feat, degE, degV, W))


# ---- module `unignnaggr` (unignnaggr.cc:81-102) ------------------------------

def unignnaggrdeg(balan_key, balan_row, group_st, group_ed, csrptr_t, indices_t, node_feat, degE, degV, options=None):
    return _sum_aggr(balan_key, balan_row, group_st, group_ed, csrptr_t, indices_t, node_feat, degE, degV, None, options)


def unignnaggr(balan_key, balan_row, group_st, group_ed, csrptr_t, indices_t, node_feat, options=None):
    return _sum_aggr(balan_key, balan_row, group_st, group_ed, csrptr_t, indices_t, node_feat, None, None, None, options)


# the names the reference's Python wrapper actually calls (unignnconv.py:7,10);
# the reference module does not export them (defect D4), this one does
unignnconvdeg = unignnaggrdeg
unignnconv = unignnaggr


# ---- wrappers (source/python/hgnnaggr.py, unignnconv.py) ----------------------

def HGNNAggr(hyperg, in_feat, degE, degV, Wdiag, first_aggr="sum", options=None):
    """first_aggr is accepted and ignored, as in the reference (hgnnaggr.py:6-7);
    it has a default so the reference test's 5-argument call works (hgnn_test.py:89)."""
    return hgnnaggr(hyperg.group_key, hyperg.group_row, hyperg.group_start, hyperg.group_end,
                    hyperg.H_T_csrptr, hyperg.H_T_colind, in_feat, degE, degV, Wdiag, options=options)


def HGNNAggrLinear(hyperg, in_feat, weight, degE, degV, Wdiag, options=None):
    """HGNNAggr(hyperg, in_feat . weight^T, ...) in one pass."""
    return hgnnaggr_linear(hyperg.H_T_csrptr, hyperg.H_T_colind, in_feat, weight, degE, degV, Wdiag, options=options)


def UniGNNConvLinear(dl, in_feat, weight, options=None):
    """UniGNNConv(dl, in_feat . weight^T) in one pass."""
    return hgnnaggr_linear(dl.H_T_csrptr, dl.H_T_colind, in_feat, weight, options=options)


def UniGNNConvdeg(dl, in_feat, degE, degV, options=None):
    return unignnaggrdeg(dl.group_key, dl.group_row, dl.group_start, dl.group_end,
                         dl.H_T_csrptr, dl.H_T_colind, in_feat, degE, degV, options=options)


def UniGNNConv(dl, in_feat, options=None):
    return unignnaggr(dl.group_key, dl.group_row, dl.group_start, dl.group_end,
                      dl.H_T_csrptr, dl.H_T_colind, in_feat, options=options)
