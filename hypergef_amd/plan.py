"""Plan objects: this backend's schedule for one hypergraph (include/hg_aggr.h).

The plan plays the role the reference's `group_key/group_row/group_start/
group_end` tensors play for its kernels (HyperGsys/hypergraph.py:96-101): it is
built once per hypergraph from `H_T_csrptr` / `H_T_colind` and reused by every
aggregation call.  The reference operators receive those two tensors on every
call, so a small cache keyed on them keeps the call signature unchanged.
"""
import collections
import ctypes
import threading

import numpy as np
import torch

from . import _lib


_CACHE_LOCK = threading.Lock()


def _ptr(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _stream_handle(device):
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def make_opts(short_max=0, split_len=0, panel_rows=0, panel_nnz=0, xcd_remap=True, host_only=False,
              t_big=0, fused_tile_bytes=0, dfs_order=False, hub_pass=True, fused_steps=0,
              row_stream=True):
    flags = 0
    if dfs_order:
        flags |= _lib.HG_PLAN_DFS_ORDER
    if not hub_pass:
        flags |= _lib.HG_PLAN_NO_HUB_PASS
    if not row_stream:
        flags |= _lib.HG_PLAN_NO_ROW_STREAM
    if host_only:
        flags |= _lib.HG_PLAN_HOST_ONLY
    if not xcd_remap:
        flags |= _lib.HG_PLAN_NO_XCD_REMAP
    return _lib.PlanOpts(short_max, split_len, panel_rows, panel_nnz, flags, t_big, fused_tile_bytes, fused_steps)


class Plan:
    """Owns an `hg_plan*`."""

    def __init__(self, handle, device=None):
        self._h = handle
        self.device = device
        info = _lib.PlanInfo()
        _lib.check(_lib.lib().hg_plan_get_info(self._h, ctypes.byref(info)))
        self.info = info.as_dict()
        self.N, self.M, self.nnz = info.N, info.M, info.nnz

    @classmethod
    def from_host(cls, N, M, csrptr_t, colind_t, opts=None, device=None):
        """csrptr_t / colind_t: host int32 arrays of H_T.  With a host-only
        `opts` nothing touches the GPU (used by the CPU tests)."""
        csrptr_t = np.ascontiguousarray(csrptr_t, dtype=np.int32)
        colind_t = np.ascontiguousarray(colind_t, dtype=np.int32)
        if csrptr_t.shape[0] != M + 1:
            raise ValueError("csrptr_t must have M + 1 entries")
        if int(csrptr_t[-1]) != colind_t.shape[0]:
            raise ValueError("csrptr_t[M] must equal len(colind_t)")
        h = ctypes.c_void_p()
        ctx = torch.cuda.device(device) if device is not None else _NullCtx()
        with ctx:
            _lib.check(_lib.lib().hg_plan_create_host(
                ctypes.byref(h), N, M, csrptr_t.ctypes.data_as(ctypes.c_void_p),
                colind_t.ctypes.data_as(ctypes.c_void_p),
                ctypes.byref(opts) if opts is not None else None))
        return cls(h, device)

    @classmethod
    def from_tensors(cls, N, csrptr_t, colind_t, opts=None):
        """csrptr_t / colind_t: int32 device tensors (the reference's
        `hyperg.H_T_csrptr` / `hyperg.H_T_colind`)."""
        _check_index(csrptr_t, "csrptr_t")
        _check_index(colind_t, "indices_t")
        M = csrptr_t.numel() - 1
        h = ctypes.c_void_p()
        with torch.cuda.device(csrptr_t.device):
            _lib.check(_lib.lib().hg_plan_create_device(
                ctypes.byref(h), N, M, colind_t.numel(), _ptr(csrptr_t), _ptr(colind_t),
                ctypes.byref(opts) if opts is not None else None, _stream_handle(csrptr_t.device)))
        return cls(h, csrptr_t.device)

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            try:
                _lib.lib().hg_plan_destroy(h)
            except Exception:
                pass

    def vertex_csr(self):
        """Host copy of the derived H CSR: (ptr_v [N+1], ind_v [nnz])."""
        ptr_v = np.empty(self.N + 1, np.int32)
        ind_v = np.empty(self.nnz, np.int32)
        _lib.check(_lib.lib().hg_plan_get_vertex_csr(
            self._h, ptr_v.ctypes.data_as(ctypes.c_void_p), ind_v.ctypes.data_as(ctypes.c_void_p)))
        return ptr_v, ind_v

    def schedule(self, hop):
        """Host copy of hop's schedule: dict of int32 [n,4] arrays (see hg_aggr.h)."""
        out = {k: np.zeros((self.info[k][hop], 4), np.int32) for k in ("panels", "tasks", "fixups")}
        _lib.check(_lib.lib().hg_plan_get_schedule(
            self._h, hop, *(out[k].ctypes.data_as(ctypes.c_void_p) for k in ("panels", "tasks", "fixups"))))
        return out

    def prepare(self, F):
        """Build the feature-width dependent (fused) schedule now; returns its shape."""
        info = _lib.FusedInfo()
        _lib.check(_lib.lib().hg_plan_prepare(self._h, F, ctypes.byref(info)))
        for k in (F, ("lin", F)):
            self.__dict__.setdefault("_ws_bytes", {}).pop(k, None)
        return info.as_dict()

    def auto_variant(self, F):
        """Name of the kernel family `variant="auto"` runs for feature width F."""
        v = _lib.lib().hg_plan_auto_variant(self._h, F)
        if v < 0:
            _lib.check(v)
        return {code: name for name, code in _lib.VARIANTS.items()}[v]

    def tune(self, csrptr_t, colind_t, X, degE=None, degV=None, W=None, iters=20):
        """Timed choice of what variant="auto" runs for X's width (hg_plan_tune_f32): the fused schedule and the
        pull variant with each of its kernels per hop (three on launch-bound graphs) are run `iters` times each on these tensors and the fastest is
        pinned -- the counterpart of the reference's tuner (HyperGAggr_tune, hgnnAgg.cuh:1115-1157), worth calling
        for a single dataset-sized hypergraph.  Returns {"variant", "pull_hop_kernels", "us": {...}}."""
        _check_feat(X, "node_feat")
        F = X.shape[1]
        self.prepare(F)
        flat = lambda t: None if t is None else t.reshape(-1)
        degE, degV, W = flat(degE), flat(degV), flat(W)
        Y = torch.empty((self.N, F), dtype=torch.float32, device=X.device)
        workspace, nbytes = self._workspace(F, X.device)
        info = _lib.TuneInfo()
        with torch.cuda.device(X.device):
            _lib.check(_lib.lib().hg_plan_tune_f32(
                self._h, F, _ptr(csrptr_t), _ptr(colind_t), _ptr(X), _ptr(degE), _ptr(degV), _ptr(W), _ptr(Y),
                _ptr(workspace), nbytes, int(iters), _stream_handle(X.device), ctypes.byref(info)))
        if hasattr(self, "_auto"):
            self._auto.pop(F, None)  # the cached answer of auto_variant may have changed
        for k in (F, ("lin", F)):
            self.__dict__.setdefault("_ws_bytes", {}).pop(k, None)
        names = {code: name for name, code in _lib.VARIANTS.items()}
        kinds = ("stream", "panels", "tasks")  # per hop: streaming row gather, row panels + wave tasks, latency schedule
        labels = ("fused", "pull") + tuple("pull/%s+%s" % (kinds[c % 3], kinds[c // 3]) for c in range(1, 9))
        return {"variant": names[info.variant], "pull_hop_kernels": info.pull_hop_kernels,
                "us": {l: float(u) for l, u in zip(labels, info.us) if u >= 0}}

    def _ensure_fused(self, F):
        if not hasattr(self, "_prepared"):
            self._prepared = set()
        if F not in self._prepared:
            self.prepare(F)
            self._prepared.add(F)

    def workspace_bytes(self, F):
        # cached per width: asking costs a library call per aggregation otherwise; prepare / tune, which can build a
        # larger layout for the width, drop the entry
        cache = self.__dict__.setdefault("_ws_bytes", {})
        n = cache.get(F)
        if n is None:
            n = cache[F] = int(_lib.lib().hg_plan_workspace_bytes(self._h, F))
        return n

    def linear_workspace_bytes(self, F_in):
        """hg_aggr_linear_workspace_bytes, cached per width like workspace_bytes (the query runs the AUTO rule under the
        plan's locks and may build a schedule: not something to pay per launch-bound layer call)."""
        cache = self.__dict__.setdefault("_ws_bytes", {})
        n = cache.get(("lin", F_in))
        if n is None:
            n = cache[("lin", F_in)] = int(_lib.lib().hg_aggr_linear_workspace_bytes(self._h, F_in))
        return n

    def _workspace(self, F, device):
        nbytes = self.workspace_bytes(F)
        # torch's caching allocator returns 512-byte aligned blocks and is stream-ordered
        return torch.empty(max(nbytes, 256), dtype=torch.uint8, device=device), nbytes

    def aggregate(self, csrptr_t, colind_t, X, degE=None, degV=None, W=None, variant="auto",
                  out=None, workspace=None, bind_scales=True):
        """Y = degV . H (degE . W . (H^T X)) on X's device, current stream.
        bind_scales: see _bind_scales; pass False for scale vectors whose contents change
        without torch noticing (numpy / DLPack aliases, `.data` writes, other libraries)."""
        _check_feat(X, "node_feat")
        if X.shape[0] != self.N:
            raise ValueError("node_feat has %d rows, hypergraph has %d vertices" % (X.shape[0], self.N))
        F = X.shape[1]
        for name, t, n in (("degE", degE, self.M), ("degV", degV, self.N), ("W", W, self.M)):
            if t is not None:
                _check_feat(t, name, device=X.device)
                if t.numel() != n:
                    raise ValueError("%s must have %d elements, got %d" % (name, n, t.numel()))
        W = self._drop_unit_weights(W, bind_scales)
        stream = torch.cuda.current_stream(X.device)  # looked up once per call: it is a measurable share of a launch-bound call
        if (degE is not None or degV is not None or W is not None) and variant in ("auto", "fused"):
            self._bind_scales(F, degE, degV, W, X.device, bind_scales, stream)
        Y = out if out is not None else torch.empty((self.N, F), dtype=torch.float32, device=X.device)
        if variant == "fused":  # hg_plan_workspace_bytes sizes for what AUTO runs; a forced fused call needs its schedule first
            self._ensure_fused(F)
        own_ws = workspace is None
        if own_ws:
            workspace, nbytes = self._workspace(F, X.device)
        else:
            nbytes = workspace.numel() * workspace.element_size()
        with torch.cuda.device(X.device):
            args = (self._h, F, _ptr(csrptr_t), _ptr(colind_t), _ptr(X), _ptr(degE), _ptr(degV), _ptr(W), _ptr(Y))
            st = _lib.lib().hg_aggr_fused_f32(*args, _ptr(workspace), nbytes, _lib.VARIANTS[variant],
                                              ctypes.c_void_p(stream.cuda_stream))
            if st == _lib.HG_ERR_WORKSPACE and own_ws:
                # The cached size is stale: the call itself built a layout for this width that needs more (e.g. AUTO on an
                # unaligned view of X with N >= 2^24 takes another schedule than the one the size was asked for).  Nothing
                # was written; ask again and retry once.
                self.__dict__.setdefault("_ws_bytes", {}).pop(F, None)
                workspace, nbytes = self._workspace(F, X.device)
                st = _lib.lib().hg_aggr_fused_f32(*args, _ptr(workspace), nbytes, _lib.VARIANTS[variant],
                                                  ctypes.c_void_p(stream.cuda_stream))
            _lib.check(st)
        return Y

    def aggregate_linear(self, csrptr_t, colind_t, X, weight, degE=None, degV=None, W=None,
                         variant="auto", out=None, workspace=None, packed=None, residual=None, ca=1.0,
                         cb=0.0, relu=False, t_out=None, bind_scales=True, math="f32"):
        """Y[N, F_out] = Aggr(X) . weight^T in one pass (hg_aggr_linear_f32); weight = nn.Linear.weight,
        [F_out, F_in]; packed = pack_linear(weight) if the caller keeps it across calls.
        With residual / ca / cb / relu / t_out: Y = act((ca * Aggr(X) + cb * residual) . weight^T) and
        t_out receives the bracket (hg_aggr_linear_res_f32: one UniGCNII / UniGIN layer per pass).  cb may be a
        one-element float32 device tensor (a learned scalar such as UniGIN's 1 + eps): the kernel then reads it from
        device memory (hg_aggr_linear_res_dev_f32) -- no read-back to the host, and a captured hipGraph sees its updates.
        math = 'bf16x6': at F_in = 128 the fused panels' matrix phase computes each fp32 product as six bf16 products
        (HG_LIN_BF16X6, include/hg_aggr.h: fp32-equivalent error, 3/8 of the matrix-pipe cycles); 'f32': fp32 MFMA.
        Raises HgError(unsupported) for widths the MFMA epilogue does not take."""
        _check_feat(X, "node_feat")
        _check_feat(weight, "weight", device=X.device)
        if X.shape[0] != self.N:
            raise ValueError("node_feat has %d rows, hypergraph has %d vertices" % (X.shape[0], self.N))
        F_in = X.shape[1]
        if weight.dim() != 2 or weight.shape[1] != F_in:
            raise ValueError("weight must be [F_out, F_in = %d]" % F_in)
        F_out = weight.shape[0]
        if packed is None:
            packed = packed_linear_cached(weight, math == "bf16x6")
        elif packed.numel() not in (weight.numel(), _lib.lib().hg_linear_pack_floats(F_out, F_in, LIN_BF16X6)):
            raise ValueError("packed does not belong to this weight")
        if math not in ("f32", "bf16x6"):
            raise ValueError("math must be 'f32' or 'bf16x6', got %r" % (math,))
        flags = (LIN_RELU if relu else 0)
        if math == "bf16x6" and F_in == 128 and packed.numel() == _lib.lib().hg_linear_pack_floats(F_out, F_in, LIN_BF16X6):
            flags |= LIN_BF16X6
        weight = packed
        for name, t, n in (("degE", degE, self.M), ("degV", degV, self.N), ("W", W, self.M)):
            if t is not None:
                _check_feat(t, name, device=X.device)
                if t.numel() != n:
                    raise ValueError("%s must have %d elements, got %d" % (name, n, t.numel()))
        W = self._drop_unit_weights(W, bind_scales)
        if (degE is not None or degV is not None or W is not None) and variant in ("auto", "fused"):
            self._bind_scales(F_in, degE, degV, W, X.device, bind_scales)
        Y = out if out is not None else torch.empty((self.N, F_out), dtype=torch.float32, device=X.device)
        if variant == "fused":
            self._ensure_fused(F_in)
        own_ws = workspace is None
        if own_ws:
            workspace = torch.empty(max(self.linear_workspace_bytes(F_in), 256), dtype=torch.uint8, device=X.device)
        nbytes = workspace.numel() * workspace.element_size()
        with torch.cuda.device(X.device):
            for name, t in (("residual", residual), ("t_out", t_out)):
                if t is not None:
                    _check_feat(t, name, device=X.device)
                    if tuple(t.shape) != (self.N, F_in):
                        raise ValueError("%s must be [N, F_in]" % name)
            if isinstance(cb, torch.Tensor):
                _check_feat(cb, "cb", device=X.device)
                if cb.numel() != 1:
                    raise ValueError("a tensor cb must hold one element")
                fn, cb_arg = _lib.lib().hg_aggr_linear_res_dev_f32, _ptr(cb)
            else:
                fn, cb_arg = _lib.lib().hg_aggr_linear_res_f32, float(cb)
            head = (self._h, F_in, F_out, _ptr(csrptr_t), _ptr(colind_t), _ptr(X), _ptr(degE), _ptr(degV),
                    _ptr(W), _ptr(weight), _ptr(residual), float(ca), cb_arg, flags, _ptr(t_out), _ptr(Y))
            stream = _stream_handle(X.device)
            st = fn(*head, _ptr(workspace), nbytes, _lib.VARIANTS[variant], stream)
            if st == _lib.HG_ERR_WORKSPACE and own_ws:  # stale cached size: see aggregate()
                self.__dict__.setdefault("_ws_bytes", {}).pop(("lin", F_in), None)
                workspace = torch.empty(max(self.linear_workspace_bytes(F_in), 256), dtype=torch.uint8, device=X.device)
                nbytes = workspace.numel()
                st = fn(*head, _ptr(workspace), nbytes, _lib.VARIANTS[variant], stream)
            _lib.check(st)
        return Y

    def _drop_unit_weights(self, W, enable=True):
        """The reference's layers each carry their own all-ones `Wdiag` (model/ugsys/hgnn.py:12): multiplying by
        exactly 1.0f is the identity, so such a W is not passed on at all -- same bits, no multiply, and the
        layers of a model then share ONE bound scale set instead of re-binding at every call.  Checked once
        per tensor (address, torch version counter); `bind_scales=False` skips the shortcut as it skips binding."""
        if W is None or not enable or W.requires_grad:
            return W
        if not hasattr(self, "_unit_w"):
            self._unit_w, self._unit_w_misses = {}, 0
        key = (W.data_ptr(), W._version, W.numel())
        hit = self._unit_w.get(key)
        if hit is None:
            if self._unit_w_misses >= 16:  # weights that keep changing and are never all ones: stop looking
                return W                   # (the check reads one flag back from the device)
            if len(self._unit_w) > 64:
                self._unit_w.clear()
            hit = self._unit_w[key] = (bool((W == 1).all().item()), W)  # keeps W (and its address) alive
            self._unit_w_misses = 0 if hit[0] else self._unit_w_misses + 1
        return None if hit[0] else W

    def _bind_scales(self, F, degE, degV, W, device, enable=True, stream=None):
        """Degree / weight vectors are graph constants: pre-gather them into the fused
        schedule's panel order once (hg_plan_bind_scales) and again only when a tensor is
        replaced or modified in place (data_ptr / torch version counter).

        The rule for callers: the library compares addresses, this layer adds torch's version
        counter.  A write that bumps neither (`t.data.mul_()`, a numpy or DLPack alias, another
        library writing in place) is invisible -- call `plan.unbind()` after such a write, or
        pass `bind_scales=False` for vectors that change that way; the kernel then gathers
        degE / W / degV itself on every call (a few percent slower, always current).

        The gather kernel runs on the stream current at binding time; a later call on another
        stream waits for it through an event."""
        if not hasattr(self, "_bound"):
            self._bound, self._auto, self._bind_stats = {}, {}, {}
        if F not in self._auto:
            self._auto[F] = self.auto_variant(F)
        if self._auto[F] != "fused":
            return
        if not enable:
            self.unbind(F)
            return
        key = tuple(None if t is None else (t.data_ptr(), t._version) for t in (degE, degV, W))
        if stream is None:
            stream = torch.cuda.current_stream(device)
        hit = self._bound.get(F)
        if hit is not None and hit[0] == key:
            if hit[2] != stream.cuda_stream:
                stream.wait_event(hit[3])
            self._bind_stats[F][0] += 1
            return
        # Callers that alternate between scale sets at one width (layers with different W, an adjoint backward)
        # would re-gather at every call: after a few re-binds that were not followed by reuse, stop binding this
        # width -- the kernels then gather degE / W / degV themselves (a few percent, not a gather + a sync).
        st = self._bind_stats.setdefault(F, [0, 0])  # [reuses, binds]
        if st[1] >= 8 and st[0] < st[1]:
            if hit is not None:
                self.unbind(F)
            return
        st[1] += 1
        with torch.cuda.device(device):
            _lib.check(_lib.lib().hg_plan_bind_scales(self._h, F, _ptr(degE), _ptr(degV), _ptr(W),
                                                      _stream_handle(device)))
            ev = torch.cuda.Event()
            ev.record(stream)
        # keep the tensors (and so their addresses) alive; remember where the gather was enqueued
        self._bound[F] = (key, (degE, degV, W), stream.cuda_stream, ev)

    def unbind(self, F=None):
        """Forget the pre-gathered scale vectors (of feature width F, or all): the next call either
        binds afresh or, with bind_scales=False, reads degE / W / degV directly."""
        bound = getattr(self, "_bound", {})
        for f in ([F] if F is not None else list(bound)):
            if f in bound:
                dev = next((t.device for t in bound[f][1] if t is not None), self.device)
                with torch.cuda.device(dev):
                    _lib.check(_lib.lib().hg_plan_bind_scales(self._h, f, None, None, None, _stream_handle(dev)))
                del bound[f]

    def gather_rows(self, hop, csrptr_t, colind_t, src, scaleA=None, scaleB=None):
        """One hop: hop 0 = H^T src (rows = hyperedges), hop 1 = H src."""
        _check_feat(src, "src")
        F = src.shape[1]
        nrows = self.M if hop == 0 else self.N
        dst = torch.empty((nrows, F), dtype=torch.float32, device=src.device)
        workspace, nbytes = self._workspace(F, src.device)
        with torch.cuda.device(src.device):
            _lib.check(_lib.lib().hg_gather_rows_f32(
                self._h, hop, F, _ptr(csrptr_t), _ptr(colind_t), _ptr(src), _ptr(scaleA), _ptr(scaleB),
                _ptr(dst), _ptr(workspace), nbytes, _stream_handle(src.device)))
        return dst


def linear_supported(F_in, F_out):
    """Widths hg_aggr_linear_f32 takes (MFMA tiles: K in {32, 64, 128}, 16-column output tiles)."""
    return F_in in (32, 64, 128) and F_out > 0 and F_out % 16 == 0


def linear_fusion_pays(F_in, F_out):
    """Measured on MI355X (profiles/r01_linear_epilogue.md): folding the projection into the
    aggregation beats linear-then-aggregate 1.5-1.8x at 32 -> 32 / 64 -> 64, still 1.1x at 64 -> 16
    and 128 -> 128, and loses (0.86x) at 128 -> 64, where the aggregation would run at twice the
    width it needs.  The operator layer fuses only where it wins."""
    return linear_supported(F_in, F_out) and (F_in <= 64 or F_out >= F_in)


LIN_RELU, LIN_BF16X6 = 1, 2  # include/hg_aggr.h: flags of the linear epilogue


def pack_linear(weight, bf16x6=False):
    """nn.Linear.weight [F_out, F_in] -> the MFMA fragment order hg_aggr_linear_f32 reads
    (hg_linear_pack_ex_f32; one tiny kernel, two with the planes).  Re-pack after every weight update.
    bf16x6: at F_in = 128 the buffer also carries the weight's three bf16 planes (HG_LIN_BF16X6); such a packing serves
    both forms of the matrix phase (aggregate_linear(..., math=)), one without them the fp32 form only."""
    _check_feat(weight, "weight")
    F_out, F_in = weight.shape
    L = _lib.lib()
    flags = LIN_BF16X6 if bf16x6 else 0
    wfrag = torch.empty(max(L.hg_linear_pack_floats(F_out, F_in, flags), F_out * F_in), dtype=torch.float32,
                        device=weight.device)
    with torch.cuda.device(weight.device):
        _lib.check(L.hg_linear_pack_ex_f32(F_out, F_in, _ptr(weight), _ptr(wfrag), flags, _stream_handle(weight.device)))
    return wfrag


_PACK_CACHE = collections.OrderedDict()
_PACK_CACHE_MAX = 64


def packed_linear_cached(weight, bf16x6=False):
    """pack_linear(weight), kept until the weight is replaced or modified in place (data_ptr /
    torch version counter: an optimizer step bumps it): inference re-uses one packing per layer
    instead of launching the pack kernel on every forward."""
    bf16x6 = bool(bf16x6) and weight.shape[1] == 128  # other widths carry no planes: one entry
    key = (weight.data_ptr(), weight._version, tuple(weight.shape), str(weight.device), bf16x6)
    stream = torch.cuda.current_stream(weight.device)
    with _CACHE_LOCK:
        hit = _PACK_CACHE.get(key)
        if hit is not None:
            _PACK_CACHE.move_to_end(key)
    if hit is not None:
        if hit[2] != stream.cuda_stream:  # packed on another stream: order this one behind the pack kernel
            stream.wait_event(hit[3])
        return hit[0]
    packed = pack_linear(weight, bf16x6)
    with torch.cuda.device(weight.device):
        ev = torch.cuda.Event()
        ev.record(stream)
    with _CACHE_LOCK:
        # the weight stays alive: its address cannot be recycled under the key
        _PACK_CACHE[key] = (packed, weight, stream.cuda_stream, ev)
        while len(_PACK_CACHE) > _PACK_CACHE_MAX:
            _PACK_CACHE.popitem(last=False)
    return packed


def clear_pack_cache():
    """Forget every cached packing.  Needed around hipGraph captures that update weights: a replay rewrites the
    weights in place behind torch's version counter, which is part of the cache key."""
    with _CACHE_LOCK:
        _PACK_CACHE.clear()


def linear_rows(X, weight, packed=None, out=None):
    """X . weight^T on the library's own fp32-MFMA rows kernel (hg_linear_rows_f32): for the
    tall-skinny products of this path it is 1.1-1.5x rocBLAS at K <= 64 and on par at K = 128."""
    _check_feat(X, "X")
    _check_feat(weight, "weight", device=X.device)
    F_out, F_in = weight.shape
    if X.dim() != 2 or X.shape[1] != F_in:
        raise ValueError("X must be [rows, F_in = %d]" % F_in)
    if packed is None:
        packed = packed_linear_cached(weight)
    Y = out if out is not None else torch.empty((X.shape[0], F_out), dtype=torch.float32, device=X.device)
    with torch.cuda.device(X.device):
        _lib.check(_lib.lib().hg_linear_rows_f32(X.shape[0], F_in, F_out, _ptr(X), _ptr(packed), _ptr(Y),
                                                 _stream_handle(X.device)))
    return Y


def wgrad_supported(F_a, F_b):
    """hg_linear_wgrad_f32's shapes: at most sixteen 16 x 16 tiles (one workgroup's accumulators), or both widths
    multiples of 64 up to 512 (64 x 64 blocks of the output)."""
    if F_a <= 0 or F_b <= 0 or F_a % 16 or F_b % 16:
        return False
    if (F_a // 16) * (F_b // 16) <= 16:
        return True
    return F_a % 64 == 0 and F_b % 64 == 0 and F_a <= 512 and F_b <= 512


def linear_wgrad(A, B):
    """A^T . B for A [N, F_a], B [N, F_b] (the linear's weight gradient, hg_linear_wgrad_f32)."""
    _check_feat(A, "A")
    _check_feat(B, "B", device=A.device)
    if A.dim() != 2 or B.dim() != 2 or A.shape[0] != B.shape[0]:
        raise ValueError("A and B must be [N, F_a] and [N, F_b]")
    N, F_a = A.shape
    F_b = B.shape[1]
    nbytes = int(_lib.lib().hg_linear_wgrad_workspace_bytes(N, F_a, F_b))
    ws = torch.empty(max(nbytes, 256), dtype=torch.uint8, device=A.device)
    C = torch.empty((F_a, F_b), dtype=torch.float32, device=A.device)
    with torch.cuda.device(A.device):
        _lib.check(_lib.lib().hg_linear_wgrad_f32(N, F_a, F_b, _ptr(A), _ptr(B), _ptr(C), _ptr(ws), nbytes,
                                                  _stream_handle(A.device)))
    return C


class _NullCtx:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


def _check_index(t, name):
    # reference: assertTensor(..., torch::kInt32), hgnnaggr_cuda.cu:8-12 -- but raising, not aborting
    if not isinstance(t, torch.Tensor) or t.dtype != torch.int32:
        raise TypeError("%s must be an int32 tensor" % name)
    if not t.is_cuda:
        raise RuntimeError("%s must be on a GPU (no CPU fallback in this backend)" % name)
    if not t.is_contiguous():
        raise RuntimeError("%s must be contiguous" % name)


def _check_feat(t, name, device=None):
    if not isinstance(t, torch.Tensor) or t.dtype != torch.float32:
        raise TypeError("%s must be a float32 tensor" % name)
    if not t.is_cuda:
        raise RuntimeError("%s must be on a GPU (no CPU fallback in this backend)" % name)
    if not t.is_contiguous():
        raise RuntimeError("%s must be contiguous" % name)
    if device is not None and t.device != device:
        raise RuntimeError("%s is on %s, expected %s" % (name, t.device, device))


# ------------------------------------------------------------------ plan cache
_CACHE = collections.OrderedDict()
_CACHE_MAX = 32
_DEFAULT_OPTS = None


def set_default_opts(opts):
    """Plan options used by plans created through the cache; clears the cache."""
    global _DEFAULT_OPTS
    with _CACHE_LOCK:
        _DEFAULT_OPTS = opts
        _CACHE.clear()


def cached_plan(N, csrptr_t, colind_t):
    """Plan for the hypergraph held in these two device tensors.  The key pins
    storage address, length, device and torch's in-place version counter; the
    entry keeps the tensors alive so an address cannot be recycled under it."""
    key = (N, csrptr_t.data_ptr(), csrptr_t.numel(), csrptr_t._version,
           colind_t.data_ptr(), colind_t.numel(), colind_t._version, str(csrptr_t.device))
    with _CACHE_LOCK:
        hit = _CACHE.get(key)
        if hit is not None:
            _CACHE.move_to_end(key)
            return hit[0]
    plan = Plan.from_tensors(N, csrptr_t, colind_t, _DEFAULT_OPTS)
    with _CACHE_LOCK:
        _CACHE[key] = (plan, csrptr_t, colind_t)
        while len(_CACHE) > _CACHE_MAX:
            _CACHE.popitem(last=False)
    return plan


def clear_plan_cache():
    with _CACHE_LOCK:
        _CACHE.clear()
