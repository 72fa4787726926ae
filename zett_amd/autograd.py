"""Training-time use of the hypernetwork forward: the same forward, differentiable with respect to every
hypernetwork parameter.

Reference call sites: train.py:1007-1013 (train_step) and 1191-1197 (eval_step) evaluate
``state.apply_fn({"params": params["hypernet"]}, target_surface_forms, target_priors, source_embeddings, lang_index)``
inside the loss, under ``jax.value_and_grad``; the gradient flows from the predicted embeddings / biases into the
hypernetwork's parameters, the source embeddings and the language model are frozen.

First slice.  fp32 arithmetic, the reference's as-written dense ``[N, L', H]`` layout (every position computed, pads
masked as keys), one ``torch.autograd.Function`` for the whole hypernetwork: its forward keeps the activations the
backward needs, its backward is straight-line code.  Both are sequences of HIP launches through the C ABI
(include/zett_hip.h "training use", zett_amd/csrc/train_ops.hip): every dense contraction — forward, dgrad, wgrad — is
the library's TN MFMA GEMM (dgrad against the transposed weight, wgrad of the two transposed activations); LayerNorm,
the two GELUs, masked attention and the source-embedding gather have forward and backward row kernels.  torch holds the
tensors and the autograd tape, allocates, slices and concatenates (layout plumbing, parameter-sized glue); it computes
nothing of size O(rows x hidden).  There is no CPU path.

Two schedules, same results to fp32 round-off (tests/test_autograd_gpu.py holds both to float64 torch autograd):
``forward_train`` / ``backward_train`` is the reference's dense layout, every position computed; ``forward_packed`` /
``backward_packed`` (the default) is the inference path's schedule — pad skipping, the input projection once per distinct
referenced id, the position-0-only last layer: exact for gradients too (a position that cannot influence hidden[:, 0]
receives a zero gradient; a value computed once and used k times receives the sum of the k gradients), 3.3x fewer FLOPs on
the headline workload.  The dense contractions run in exact fp32 MFMA (default) or on 16-bit MFMA operands (bf16 / f16:
``model.train_precision``; operands converted — and for dgrad / wgrad transposed in the same pass — per launch, fp32
accumulation, everything else fp32).  bf16 is the 16-bit mode to train in: f16 operands are ~8x more accurate but have the
half range and there is no loss scaling here — a step whose gradients overflow raises RangeError instead of returning NaNs.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Optional, Tuple

import torch

from . import _lib
from .dims import PROJECTOR_LN_EPS, HypernetDims, weight_shapes

_SRC_DTYPES = {torch.float32: _lib.DTYPE_F32, torch.float16: _lib.DTYPE_F16, torch.bfloat16: _lib.DTYPE_BF16}
ACT_NONE, GELU_TANH, GELU_ERF = 0, 1, 2
K_STEP = 32          # contraction widths of the fp32 MFMA GEMM are multiples of this


NEVER_READ = frozenset({"model.embeddings.word_embeddings.weight"})     # inputs_embeds bypass the table (modeling_hypernet.py:225-229)


def _ptr(t: Optional[torch.Tensor]):
    return C.c_void_p(0 if t is None else t.data_ptr())


class Ops:
    """The HIP primitives on torch tensors (fp32, one cuda device, current stream)."""

    def __init__(self, device: torch.device, precision: str = "f32"):
        if device.type != "cuda":
            raise RuntimeError("zett_amd computes on MI355X only: the differentiable forward needs cuda (ROCm) tensors; there is no CPU path")
        if precision not in ("f32", "bf16", "f16"):
            raise ValueError("training precision must be f32, bf16 or f16")
        self.lib = _lib.load()
        self.device = device
        # arithmetic of the dense contractions (forward, dgrad, wgrad): exact fp32 MFMA, or 16-bit MFMA operands with fp32
        # accumulation and fp32 everything else (parameters, activations, LayerNorm / GELU / softmax, gradients)
        self.precision = precision
        self.prec = {"f32": None, "bf16": _lib.PREC_BF16, "f16": _lib.PREC_F16}[precision]
        self.lo_dtype = {"f32": None, "bf16": torch.bfloat16, "f16": torch.float16}[precision]
        self.kstep = K_STEP if self.prec is None else 64          # contraction widths are multiples of this

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def new(self, *shape) -> torch.Tensor:
        return torch.empty(shape, dtype=torch.float32, device=self.device)

    def _twin(self, x):
        """the 16-bit copy a producing kernel wrote next to an fp32 activation (LayerNorm forward), if it matches this arithmetic"""
        t = getattr(x, "_zett_lo", None)
        return t if t is not None and t.dtype == self.lo_dtype and t.shape[0] == x.shape[0] and t.shape[1] >= x.shape[1] and x.shape[1] % 64 == 0 else None

    def keep(self, x):
        """What the backward needs of an activation that is ONLY an operand there (the x of a Linear: its weight gradient
        transposes it, nothing reads its fp32 values again): in 16-bit arithmetic the 16-bit twin the producing kernel wrote —
        the fp32 tensor is then free as soon as the forward has passed it (r5: 7.5 GB of an 88 GB step at the full vocabulary)."""
        t = self._twin(x)
        return x if t is None else t

    # ---- dense contraction: y[M,N] = act(x[M,K] . w[N,K]^T + bias) + residual
    def to_lo(self, x):
        """fp32 [R, C] -> 16-bit [R, C'] (C' = C zero-padded to the 64-wide K step): an operand of zett_op_gemm_lo"""
        assert x.dim() == 2 and x.stride(1) == 1 and x.dtype == torch.float32
        r, c = x.shape
        cp = -(-c // 64) * 64
        out = torch.empty((r, cp), dtype=self.lo_dtype, device=self.device)
        _lib.check(self.lib.zett_op_convert_lo(self.prec, _ptr(x), x.stride(0), _ptr(out), cp, r, c, cp, self._stream()), "zett_op_convert_lo")
        return out

    def gemm(self, x, w, bias=None, act=ACT_NONE, residual=None, out=None):
        assert x.dim() == 2 and w.dim() == 2 and x.stride(1) == 1 and w.stride(1) == 1
        if self.prec is not None:
            xa = x if x.dtype == self.lo_dtype else self._twin(x)
            if xa is None:
                xa = self.to_lo(x)
            wa = w if w.dtype == self.lo_dtype else self.to_lo(w)
            assert xa.shape[1] == wa.shape[1]
            m, k, n = xa.shape[0], xa.shape[1], wa.shape[0]
            if out is None:
                out = self.new(m, n)
            assert out.shape == (m, n) and out.is_contiguous()
            _lib.check(self.lib.zett_op_gemm_lo(self.prec, _ptr(xa), xa.stride(0), _ptr(wa), wa.stride(0), m, n, k, _ptr(bias), act,
                                                _ptr(residual), 0 if residual is None else residual.stride(0), _ptr(out), n, self._stream()), "zett_op_gemm_lo")
            return out
        assert x.shape[1] == w.shape[1]
        m, k = x.shape
        n = w.shape[0]
        if out is None:
            out = self.new(m, n)
        assert out.shape == (m, n) and out.is_contiguous()
        _lib.check(self.lib.zett_op_gemm_f32(_ptr(x), x.stride(0), _ptr(w), w.stride(0), m, n, k, _ptr(bias), act,
                                             _ptr(residual), 0 if residual is None else residual.stride(0), _ptr(out), n, self._stream()), "zett_op_gemm_f32")
        return out

    def wgrad(self, dy_t, x_t):
        """dW[N, K] = dy_t[N, M'] . x_t[K, M']^T (M' = rows, zero-padded to the K step).  The output is N*K/65536 tiles of 256x256
        whatever the batch: for a narrow hypernetwork (N, K <= 2304: 9-54 tiles) one launch would leave most of the 256 CUs
        idle while each tile walks tens of thousands of rows.  Then the rows are cut into S slices, the S partial products run
        as concurrent launches on side streams (same kernel, same arithmetic per slice) and one deterministic column sum
        over the [S, N*K] partials adds them."""
        n, k, mp = dy_t.shape[0], x_t.shape[0], dy_t.shape[1]
        tiles = -(-n // 256) * -(-k // 256)
        s_max = min(16, 256 // max(tiles, 1), mp // 2048)
        if s_max < 2:
            return self.gemm(dy_t, x_t)
        per = -(-(mp // self.kstep) // s_max) * self.kstep             # slice width, a multiple of the K step
        bounds = [(a, min(a + per, mp)) for a in range(0, mp, per)]
        part = self.new(len(bounds), n, k)
        if not hasattr(self, "_side"):
            self._side = [torch.cuda.Stream(device=self.device) for _ in range(16)]
        main = torch.cuda.current_stream(self.device)
        for i, (a, b) in enumerate(bounds):
            st = self._side[i]
            st.wait_stream(main)
            with torch.cuda.stream(st):
                self.gemm(dy_t[:, a:b], x_t[:, a:b], out=part[i])
        for i in range(len(bounds)):
            main.wait_stream(self._side[i])
        return self._colsum_raw(part.view(len(bounds), n * k)).view(n, k)

    def transpose(self, x, pad_to=None):
        """[R, C] -> [C, R'] (R' = R zero-padded to the K step), in the operand type of the training GEMMs (fp32, or 16-bit:
        the conversion rides on the transposition; an activation that is already stored as a 16-bit operand is transposed as is)"""
        assert x.dim() == 2 and x.stride(1) == 1
        r, c = x.shape
        pad_to = pad_to or self.kstep
        rp = -(-r // pad_to) * pad_to
        if self.prec is not None:
            out = torch.empty((c, rp), dtype=self.lo_dtype, device=self.device)
            src = x if x.dtype == self.lo_dtype else self._twin(x)
            if src is not None:
                _lib.check(self.lib.zett_op_transpose_lo16(self.prec, _ptr(src), src.stride(0), _ptr(out), rp, r, c, rp, self._stream()), "zett_op_transpose_lo16")
            else:
                _lib.check(self.lib.zett_op_transpose_lo(self.prec, _ptr(x), x.stride(0), _ptr(out), rp, r, c, rp, self._stream()), "zett_op_transpose_lo")
            return out
        out = self.new(c, rp)
        _lib.check(self.lib.zett_op_transpose_f32(_ptr(x), x.stride(0), _ptr(out), rp, r, c, rp, self._stream()), "zett_op_transpose_f32")
        return out

    def _colsum_raw(self, x, out=None, accumulate=False):
        assert x.dim() == 2 and x.stride(1) == 1
        r, c = x.shape
        if out is None:
            out = self.new(c)
        _lib.check(self.lib.zett_op_colsum_f32(_ptr(x), x.stride(0), r, c, _ptr(out), int(accumulate), self._stream()), "zett_op_colsum_f32")
        return out

    def colsum(self, x, out=None, accumulate=False):
        """out[c] (+)= sum_r x[r, c].  The kernel gives one workgroup 64 columns and all rows: a tall, narrow matrix (80 k rows x
        768 columns) would run on 12 workgroups.  Tall contiguous inputs are therefore summed in two deterministic passes on
        the same kernel: viewed as [R/k, k*C] (a view row = k consecutive rows) the first pass yields k partial vectors on
        k*C/64 workgroups, the second adds them; the < k rows left over are accumulated on top."""
        r, c = x.shape
        k = min(r // 64, max(1, 65536 // c)) if r >= 128 else 1
        if k <= 1 or not x.is_contiguous():
            return self._colsum_raw(x, out, accumulate)
        r0 = (r // k) * k
        part = self._colsum_raw(x[:r0].view(r0 // k, k * c))
        out = self._colsum_raw(part.view(k, c), out, accumulate)
        if r0 < r:
            out = self._colsum_raw(x[r0:], out, True)
        return out

    def _ew(self, op, a, b=None, vec=None, vec2=None, s=None, rows=None, cols=None):
        ref = a if a is not None else None
        n = ref.numel() if ref is not None else rows * cols
        cols = cols if cols is not None else (ref.shape[-1] if ref is not None else 1)
        out = self.new(*(ref.shape if ref is not None else (rows, cols)))
        for t in (a, b):
            assert t is None or t.is_contiguous()
        _lib.check(self.lib.zett_op_elementwise_f32(op, _ptr(a), _ptr(b), _ptr(vec), _ptr(vec2), _ptr(s), _ptr(out), n, cols, self._stream()),
                   "zett_op_elementwise_f32")
        return out

    def add(self, a, b):
        return self._ew(0, a, b)

    def mul(self, a, b):
        return self._ew(1, a, b)

    def affine_cols(self, a, vec=None, vec2=None):          # a * vec[col] + vec2[col]
        return self._ew(2, a, vec=vec, vec2=vec2)

    def add_outer(self, a, s, vec):                          # a + s[row] * vec[col]
        return self._ew(3, a, vec=vec, s=s)

    def scale_rows(self, a, s):                              # a * s[row]
        return self._ew(4, a, s=s)

    def rowdot(self, a, w, b=None):
        assert a.dim() == 2 and a.stride(1) == 1
        out = self.new(a.shape[0])
        _lib.check(self.lib.zett_op_rowdot_f32(_ptr(a), a.stride(0), _ptr(w), _ptr(b), _ptr(out), a.shape[0], a.shape[1], self._stream()), "zett_op_rowdot_f32")
        return out

    def layernorm(self, x, gamma, beta, eps):
        """-> y, stats.  In 16-bit arithmetic y carries its 16-bit twin (y._zett_lo, written by the same kernel): the operand of
        the contraction that reads y next, and of the weight gradient that transposes it in the backward."""
        assert x.dim() == 2 and x.stride(1) == 1
        r, h = x.shape
        y, stats = self.new(r, h), self.new(r, 2)
        twin = torch.empty((r, h), dtype=self.lo_dtype, device=self.device) if self.prec is not None and h % 64 == 0 else None
        _lib.check(self.lib.zett_op_layernorm_fwd_f32(_ptr(x), x.stride(0), _ptr(gamma), _ptr(beta), float(eps), _ptr(y), _ptr(stats), r, h, _ptr(twin),
                                                      self.prec if twin is not None else 0, self._stream()), "zett_op_layernorm_fwd_f32")
        if twin is not None:
            y._zett_lo = twin
        return y, stats

    def layernorm_bwd(self, dy, x, stats, gamma, dy2=None):
        """-> dx, dgamma, dbeta for the gradient dy (+ dy2: the part arriving over the residual branch, added inside the kernel).
        The parameter gradients leave the kernel as per-workgroup partial sums (no [R, H] product is written)."""
        assert dy.is_contiguous() and x.stride(1) == 1 and (dy2 is None or (dy2.is_contiguous() and dy2.shape == dy.shape))
        r, h = x.shape
        n_part = max(1, min(r, 1024))
        dx, part = self.new(r, h), self.new(n_part, 2 * h)
        _lib.check(self.lib.zett_op_layernorm_bwd_f32(_ptr(dy), _ptr(dy2), _ptr(x), x.stride(0), _ptr(stats), _ptr(gamma), _ptr(dx), _ptr(part), n_part, r, h,
                                                      self._stream()), "zett_op_layernorm_bwd_f32")
        g = self._colsum_raw(part)
        return dx, g[:h], g[h:]

    def gelu(self, z, kind, operand=False):
        """operand = True (the value only feeds a contraction): in 16-bit arithmetic the result is written as that operand and
        its fp32 form is never stored"""
        if operand and self.prec is not None and z.shape[-1] % 64 == 0 and z.is_contiguous():
            h = torch.empty(z.shape, dtype=self.lo_dtype, device=self.device)
            _lib.check(self.lib.zett_op_gelu_fwd_lo(self.prec, _ptr(z), _ptr(h), z.numel(), kind, self._stream()), "zett_op_gelu_fwd_lo")
            return h
        h = self.new(*z.shape)
        _lib.check(self.lib.zett_op_gelu_fwd_f32(_ptr(z), _ptr(h), z.numel(), kind, self._stream()), "zett_op_gelu_fwd_f32")
        return h

    def gelu_bwd(self, z, dh, kind):
        assert dh.is_contiguous() and z.is_contiguous()
        dz = self.new(*z.shape)
        _lib.check(self.lib.zett_op_gelu_bwd_f32(_ptr(z), _ptr(dh), _ptr(dz), z.numel(), kind, self._stream()), "zett_op_gelu_bwd_f32")
        return dz

    def attention(self, q, k, v, mask, row_offset, n_rows, seq, heads, hidden, cls_only=False, operand=False):
        """q [Tq, *] (Tq = positions, or n_rows when cls_only), k / v [T, *] column views, mask uint8 [T] (key visible),
        row_offset int32 [n_rows + 1] or None (dense: seq positions per row) -> ctx [Tq, H], probs [n_rows, heads, seq, seq].
        operand = True (the context only feeds the output projection): in 16-bit arithmetic ctx is written as that operand."""
        assert k.stride(0) == v.stride(0) and q.stride(1) == 1 and k.stride(1) == 1 and v.stride(1) == 1
        d = hidden // heads
        lo = operand and self.prec is not None and hidden % 64 == 0
        ctx = torch.empty((q.shape[0], hidden), dtype=self.lo_dtype if lo else torch.float32, device=self.device)
        probs = torch.zeros((n_rows, heads, seq, seq), dtype=torch.float32, device=self.device)
        _lib.check(self.lib.zett_op_attention_fwd_f32(_ptr(q), q.stride(0), _ptr(k), _ptr(v), k.stride(0), _ptr(mask), _ptr(row_offset), n_rows, seq, heads, d,
                                                      int(cls_only), _ptr(None if lo else ctx), hidden, _ptr(probs), _ptr(ctx if lo else None),
                                                      self.prec if lo else 0, self._stream()), "zett_op_attention_fwd_f32")
        return ctx, probs

    def attention_bwd(self, dctx, q, k, v, probs, row_offset, n_rows, seq, heads, hidden, dq, dk, dv, cls_only=False):
        """writes dq (rows like q), dk / dv (rows like k; column views of one buffer with equal row stride)"""
        assert dctx.is_contiguous() and dk.stride(0) == dv.stride(0)
        d = hidden // heads
        _lib.check(self.lib.zett_op_attention_bwd_f32(_ptr(dctx), hidden, _ptr(q), q.stride(0), _ptr(k), _ptr(v), k.stride(0), _ptr(probs), _ptr(row_offset),
                                                      n_rows, seq, heads, d, int(cls_only), _ptr(dq), dq.stride(0), _ptr(dk), _ptr(dv), dk.stride(0),
                                                      self._stream()), "zett_op_attention_bwd_f32")

    def gather_rows(self, src, idx, a=None):
        """out[r] = (a[r] if a is given else 0) + src[idx[r]]"""
        assert src.stride(1) == 1 and idx.dtype == torch.int32 and (a is None or a.is_contiguous())
        out = self.new(idx.numel(), src.shape[1])
        _lib.check(self.lib.zett_op_gather_rows_f32(_ptr(a), _ptr(src), src.stride(0), _ptr(idx), _ptr(out), idx.numel(), src.shape[1], self._stream()),
                   "zett_op_gather_rows_f32")
        return out

    def scatter_add_rows(self, dst, idx, src):
        """dst[idx[r]] += src[r] (in place; dst must be initialised)"""
        assert dst.stride(1) == 1 and src.is_contiguous() and idx.dtype == torch.int32 and idx.numel() == src.shape[0]
        _lib.check(self.lib.zett_op_scatter_add_rows_f32(_ptr(dst), dst.stride(0), _ptr(idx), _ptr(src), idx.numel(), src.shape[1], self._stream()),
                   "zett_op_scatter_add_rows_f32")
        return dst

    def gather(self, ids, src, v0, fallback, sw, sb):
        t = ids.numel()
        x = self.new(t, src.shape[1])
        _lib.check(self.lib.zett_op_gather_fwd_f32(_ptr(ids), t, _ptr(src), _SRC_DTYPES[src.dtype], src.shape[1], v0, _ptr(fallback), _ptr(sw), _ptr(sb),
                                                   _ptr(x), self._stream()), "zett_op_gather_fwd_f32")
        return x

    def gather_bwd(self, ids, src, v0, dx, n_fallback):
        """-> dfallback [n_fallback, E_in], d in_scaler.w [E_in], d in_scaler.b [E_in]"""
        t, e_in = dx.shape
        dfb = torch.zeros((n_fallback, e_in), dtype=torch.float32, device=self.device)
        prod, keep = self.new(t, e_in), self.new(t, e_in)
        _lib.check(self.lib.zett_op_gather_bwd_f32(_ptr(ids), t, _ptr(src), _SRC_DTYPES[src.dtype], e_in, v0, _ptr(dx), _ptr(dfb), _ptr(prod), _ptr(keep),
                                                   self._stream()), "zett_op_gather_bwd_f32")
        return dfb, self.colsum(prod), self.colsum(keep)

    # ---- Linear backward on the same GEMM: dgrad against W^T, wgrad of the transposed activations
    def grad_operands(self, dy, act_z=None, act_kind=0):
        """16-bit arithmetic: one read of dy [M, N] -> (lo(dy) [M, N], lo(dy)^T [N, M'], column sums of dy [N]); with act_z
        (the pre-activation of the GELU whose output gradient dy is) the three are those of dy * gelu'(act_z)"""
        m, n = dy.shape
        mp = -(-m // self.kstep) * self.kstep
        bands = -(-m // 64)
        dy_lo = torch.empty((m, n), dtype=self.lo_dtype, device=self.device)
        dy_t = torch.empty((n, mp), dtype=self.lo_dtype, device=self.device)
        part = self.new(bands, n)
        assert act_z is None or (act_z.shape == dy.shape and act_z.stride(1) == 1)
        _lib.check(self.lib.zett_op_grad_operands_lo(self.prec, _ptr(dy), dy.stride(0), _ptr(act_z), 0 if act_z is None else act_z.stride(0), act_kind, m, n, mp,
                                                     _ptr(dy_lo), n, _ptr(dy_t), mp, _ptr(part), self._stream()), "zett_op_grad_operands_lo")
        return dy_lo, dy_t, self.colsum(part)

    def linear_bwd(self, dy, x, w, act_z=None, act_kind=0):
        """y = x w^T + b  ->  dx [M, K], dw [N, K], db [N].  act_z / act_kind: y went through a GELU and dy is the gradient of the
        GELU's output (the activation's backward is applied on the way: fused into the operand pass in 16-bit arithmetic)"""
        assert dy.is_contiguous() and dy.shape == (x.shape[0], w.shape[0])
        if w.shape[0] % self.kstep:
            raise NotImplementedError(f"the training GEMM contracts over multiples of {self.kstep}: a Linear with {w.shape[0]} outputs is not supported yet")
        if self.prec is not None and dy.shape[0] > 0:
            dy_lo, dy_t, db = self.grad_operands(dy, act_z, act_kind)
            return self.gemm(dy_lo, self.transpose(w)), self.wgrad(dy_t, self.transpose(x)), db
        if act_z is not None:
            dy = self.gelu_bwd(act_z, dy, act_kind)
        dx = self.gemm(dy, self.transpose(w))                           # A = dy [M, N], W-operand = w^T [K, N]
        dw = self.wgrad(self.transpose(dy), self.transpose(x))          # A = dy^T [N, M'], W-operand = x^T [K, M'] (M' = rows zero-padded to 32)
        return dx, dw, self.colsum(dy)


def _projector_fwd(ops: Ops, P, prefix, x):
    """ProjectorBlock (modeling_hypernet.py:22-40): LN_1e-6(gelu_t(W2 gelu_t(W1 x + b1) + b2) + x)"""
    z1 = ops.gemm(x, P[prefix + "dense1.weight"], P[prefix + "dense1.bias"])
    h1 = ops.gelu(z1, GELU_TANH, operand=True)
    z2 = ops.gemm(h1, P[prefix + "dense2.weight"], P[prefix + "dense2.bias"])
    h2 = ops.gelu(z2, GELU_TANH)
    s = ops.add(h2, x)
    y, st = ops.layernorm(s, P[prefix + "ln.weight"], P[prefix + "ln.bias"], PROJECTOR_LN_EPS)
    return y, dict(x=ops.keep(x), z1=z1, h1=h1, z2=z2, s=s, st=st)


def _projector_bwd(ops: Ops, P, G, prefix, saved, dy):
    ds, G[prefix + "ln.weight"], G[prefix + "ln.bias"] = ops.layernorm_bwd(dy, saved["s"], saved["st"], P[prefix + "ln.weight"])
    dh1, G[prefix + "dense2.weight"], G[prefix + "dense2.bias"] = ops.linear_bwd(ds, saved["h1"], P[prefix + "dense2.weight"], saved["z2"], GELU_TANH)
    dx, G[prefix + "dense1.weight"], G[prefix + "dense1.bias"] = ops.linear_bwd(dh1, saved["x"], P[prefix + "dense1.weight"], saved["z1"], GELU_TANH)
    return ops.add(dx, ds)          # the residual branch


def _heads_fwd(ops: Ops, dims: HypernetDims, P, S, cls, n, device):
    """CLS -> output heads, Rescalers, bias head (modeling_hypernet.py:231-267) -> (pred_in, pred_out | None, bias)"""
    e = dims.n_embd
    h_in, S["pb_out0"] = _projector_fwd(ops, P, "output_projection.0.", cls)
    S["h_in"] = ops.keep(h_in)
    pred = ops.gemm(h_in, P["output_projection.1.weight"], P["output_projection.1.bias"])
    S["pred_raw"] = pred
    pred_out = None
    if dims.single_head:
        if dims.rescale:
            scale = torch.cat([P["scaler.w"].reshape(-1)] + ([P["out_scaler.w"].reshape(-1)] if dims.separate_out else []))
            shift = torch.cat([P["scaler.b"].reshape(-1)] + ([P["out_scaler.b"].reshape(-1)] if dims.separate_out else []))
            pred = ops.affine_cols(pred, scale, shift)
        pred_in = pred[:, :e].contiguous()
        pred_out = pred[:, e:].contiguous() if dims.separate_out else None
    else:
        pred_in = ops.affine_cols(pred, P["scaler.w"].reshape(-1), P["scaler.b"].reshape(-1)) if dims.rescale else pred
        if dims.separate_out:
            h_out, S["pb_out1"] = _projector_fwd(ops, P, "output_projection_out.0.", cls)
            S["h_out"] = ops.keep(h_out)
            po = ops.gemm(h_out, P["output_projection_out.1.weight"], P["output_projection_out.1.bias"])
            S["pred_out_raw"] = po
            pred_out = ops.affine_cols(po, P["out_scaler.w"].reshape(-1), P["out_scaler.b"].reshape(-1)) if dims.rescale else po
    if dims.predict_bias:
        bias = ops.rowdot(cls, P["bias_projection.weight"].reshape(-1), P["bias_projection.bias"].reshape(-1))
    else:
        bias = torch.zeros((n,), dtype=torch.float32, device=device)
    return pred_in, pred_out, bias


def _heads_bwd(ops: Ops, dims: HypernetDims, P, S, G, n, d_in, d_out, d_bias):
    """Backward of _heads_fwd: fills G for the heads' parameters, returns the gradient of hidden[:, 0] [n, H]"""
    e = dims.n_embd
    cls = S["cls"]
    dev = cls.device
    zeros = lambda *shape: torch.zeros(shape, dtype=torch.float32, device=dev)
    d_in = zeros(n, e) if d_in is None else d_in.contiguous().float()
    if dims.separate_out:
        d_out = zeros(n, e) if d_out is None else d_out.contiguous().float()
    # ---- heads
    if dims.single_head:
        dpred = torch.cat([d_in, d_out], 1).contiguous() if dims.separate_out else d_in
        if dims.rescale:
            raw = S["pred_raw"]
            gw, gb = ops.colsum(ops.mul(dpred, raw)), ops.colsum(dpred)
            G["scaler.w"], G["scaler.b"] = gw[:e].reshape(1, -1).clone(), gb[:e].reshape(1, -1).clone()
            if dims.separate_out:
                G["out_scaler.w"], G["out_scaler.b"] = gw[e:].reshape(1, -1).clone(), gb[e:].reshape(1, -1).clone()
            scale = torch.cat([P["scaler.w"].reshape(-1)] + ([P["out_scaler.w"].reshape(-1)] if dims.separate_out else []))
            dpred = ops.affine_cols(dpred, scale, None)
        dh, G["output_projection.1.weight"], G["output_projection.1.bias"] = ops.linear_bwd(dpred, S["h_in"], P["output_projection.1.weight"])
        dcls = _projector_bwd(ops, P, G, "output_projection.0.", S["pb_out0"], dh)
    else:
        dpred = d_in
        if dims.rescale:
            G["scaler.w"] = ops.colsum(ops.mul(dpred, S["pred_raw"])).reshape(1, -1)
            G["scaler.b"] = ops.colsum(dpred).reshape(1, -1)
            dpred = ops.affine_cols(dpred, P["scaler.w"].reshape(-1), None)
        dh, G["output_projection.1.weight"], G["output_projection.1.bias"] = ops.linear_bwd(dpred, S["h_in"], P["output_projection.1.weight"])
        dcls = _projector_bwd(ops, P, G, "output_projection.0.", S["pb_out0"], dh)
        if dims.separate_out:
            dpo = d_out
            if dims.rescale:
                G["out_scaler.w"] = ops.colsum(ops.mul(dpo, S["pred_out_raw"])).reshape(1, -1)
                G["out_scaler.b"] = ops.colsum(dpo).reshape(1, -1)
                dpo = ops.affine_cols(dpo, P["out_scaler.w"].reshape(-1), None)
            dh2, G["output_projection_out.1.weight"], G["output_projection_out.1.bias"] = ops.linear_bwd(dpo, S["h_out"], P["output_projection_out.1.weight"])
            dcls = ops.add(dcls, _projector_bwd(ops, P, G, "output_projection_out.0.", S["pb_out1"], dh2))
    if dims.predict_bias:
        db = zeros(n) if d_bias is None else d_bias.contiguous().float()
        wb = P["bias_projection.weight"].reshape(-1)
        G["bias_projection.weight"] = ops.colsum(ops.scale_rows(cls, db)).reshape(1, -1)
        G["bias_projection.bias"] = db.view(1, -1).sum(1) if n == 0 else ops.colsum(db.view(-1, 1)).reshape(1)
        dcls = ops.add_outer(dcls, db, wb)
    return dcls


def forward_train(ops: Ops, dims: HypernetDims, ln_eps: float, P: Dict[str, torch.Tensor], ids: torch.Tensor, src: torch.Tensor, lang: int):
    """The as-written forward (modeling_hypernet.py:156-267) on HIP primitives, keeping what the backward needs.
    -> (pred_in, pred_out | None, bias), saved"""
    n, L = ids.shape
    lam = 1 if dims.embed_lang else 0
    Lp, H = L + lam, dims.hidden
    for name, width in (("n_embd", dims.n_embd), ("n_in_embd", dims.n_in_embd), ("hidden", H), ("intermediate", dims.intermediate)):
        if width % ops.kstep:
            raise NotImplementedError(f"{name} = {width}: the training GEMM contracts over multiples of {ops.kstep}")
    S = {}
    ids32 = ids.to(torch.int32).contiguous()
    S["ids"] = ids32
    sw = P["in_scaler.w"].reshape(-1) if dims.rescale else None
    sb = P["in_scaler.b"].reshape(-1) if dims.rescale else None
    x0 = ops.gather(ids32.view(-1), src, dims.original_vocab_size, P["fallback_embeddings.weight"], sw, sb)          # [N*L, E_in]
    y0 = ops.gemm(x0, P["input_projection.0.weight"], P["input_projection.0.bias"])
    t, S["pb_in"] = _projector_fwd(ops, P, "input_projection.1.", y0)
    S["x0"] = x0
    # lang token + RobertaEmbeddings (modeling_hypernet.py:192-229): x + type[0] + pos[p], LayerNorm
    type0 = P["model.embeddings.token_type_embeddings.weight"][0]
    pos = P["model.embeddings.position_embeddings.weight"]
    xin = ops.new(n, Lp, H)
    xin[:, :L] = t.view(n, L, H)                                           # layout plumbing (copies), not arithmetic
    if lam:
        xin[:, L] = P["lang_embeddings.weight"][lang] - (type0 + pos[L])   # parameter-sized glue ([H])
    posadd = (type0[None, :] + pos[:Lp]).contiguous().view(-1)             # [L'*H], parameter-sized
    emb = ops.affine_cols(xin.view(n, Lp * H), None, posadd).view(n * Lp, H)
    z, st = ops.layernorm(emb, P["model.embeddings.LayerNorm.weight"], P["model.embeddings.LayerNorm.bias"], ln_eps)
    S["emb"], S["emb_st"] = emb, st
    mask = torch.ones((n, Lp), dtype=torch.uint8, device=ids.device)
    mask[:, :L] = (ids != dims.pad_token_id).to(torch.uint8)
    S["mask"] = mask
    layers = []
    for l in range(dims.layers):
        p = f"model.encoder.layer.{l}."
        a = p + "attention.self."
        wqkv = torch.cat([P[a + "query.weight"], P[a + "key.weight"], P[a + "value.weight"]], 0)      # fused operand (parameter plumbing)
        bqkv = torch.cat([P[a + "query.bias"], P[a + "key.bias"], P[a + "value.bias"]], 0)
        qkv = ops.gemm(z, wqkv, bqkv)
        ctx, probs = ops.attention(qkv[:, :H], qkv[:, H:2 * H], qkv[:, 2 * H:], mask.view(-1), None, n, Lp, dims.heads, H, operand=True)
        s1 = ops.gemm(ctx, P[p + "attention.output.dense.weight"], P[p + "attention.output.dense.bias"], residual=z)
        z1, st1 = ops.layernorm(s1, P[p + "attention.output.LayerNorm.weight"], P[p + "attention.output.LayerNorm.bias"], ln_eps)
        u = ops.gemm(z1, P[p + "intermediate.dense.weight"], P[p + "intermediate.dense.bias"])
        g = ops.gelu(u, GELU_ERF, operand=True)
        s2 = ops.gemm(g, P[p + "output.dense.weight"], P[p + "output.dense.bias"], residual=z1)
        z2, st2 = ops.layernorm(s2, P[p + "output.LayerNorm.weight"], P[p + "output.LayerNorm.bias"], ln_eps)
        layers.append(dict(z=z, wqkv=wqkv, qkv=qkv, probs=probs, ctx=ctx, s1=s1, st1=st1, z1=z1, u=u, g=g, s2=s2, st2=st2))
        z = z2
    S["layers"] = layers
    cls = z.view(n, Lp, H)[:, 0].contiguous()                                # hidden[:, 0] (modeling_hypernet.py:234)
    S["cls"] = cls
    return _heads_fwd(ops, dims, P, S, cls, n, ids.device), S


def backward_train(ops: Ops, dims: HypernetDims, P: Dict[str, torch.Tensor], S, src: torch.Tensor, lang: int,
                   d_in: Optional[torch.Tensor], d_out: Optional[torch.Tensor], d_bias: Optional[torch.Tensor]) -> Dict[str, torch.Tensor]:
    """Gradients of every hypernetwork parameter, given the gradients of the three outputs."""
    G: Dict[str, torch.Tensor] = {}
    n, L = S["ids"].shape
    lam = 1 if dims.embed_lang else 0
    Lp, H, e = L + lam, dims.hidden, dims.n_embd
    dev = S["cls"].device
    zeros = lambda *shape: torch.zeros(shape, dtype=torch.float32, device=dev)
    dcls = _heads_bwd(ops, dims, P, S, G, n, d_in, d_out, d_bias)
    # ---- encoder (only position 0 of the last hidden state carries a gradient)
    dz = zeros(n, Lp, H)
    dz[:, 0] = dcls                                                       # layout plumbing
    dz, dz_res = dz.view(n * Lp, H), None                                 # dz_res: the part of the gradient that came over a residual branch
    for l in reversed(range(dims.layers)):
        p = f"model.encoder.layer.{l}."
        a = p + "attention.self."
        A = S["layers"][l]
        ds2, G[p + "output.LayerNorm.weight"], G[p + "output.LayerNorm.bias"] = ops.layernorm_bwd(dz, A["s2"], A["st2"], P[p + "output.LayerNorm.weight"], dy2=dz_res)
        dg, G[p + "output.dense.weight"], G[p + "output.dense.bias"] = ops.linear_bwd(ds2, A["g"], P[p + "output.dense.weight"])
        dz1, G[p + "intermediate.dense.weight"], G[p + "intermediate.dense.bias"] = ops.linear_bwd(dg, A["z1"], P[p + "intermediate.dense.weight"], A["u"], GELU_ERF)
        ds1, G[p + "attention.output.LayerNorm.weight"], G[p + "attention.output.LayerNorm.bias"] = \
            ops.layernorm_bwd(dz1, A["s1"], A["st1"], P[p + "attention.output.LayerNorm.weight"], dy2=ds2)        # + the residual of the FFN
        dctx, G[p + "attention.output.dense.weight"], G[p + "attention.output.dense.bias"] = ops.linear_bwd(ds1, A["ctx"], P[p + "attention.output.dense.weight"])
        dqkv = ops.new(n * Lp, 3 * H)
        qkv = A["qkv"]
        ops.attention_bwd(dctx, qkv[:, :H], qkv[:, H:2 * H], qkv[:, 2 * H:], A["probs"], None, n, Lp, dims.heads, H,
                          dqkv[:, :H], dqkv[:, H:2 * H], dqkv[:, 2 * H:])
        dzin, dwqkv, dbqkv = ops.linear_bwd(dqkv, A["z"], A["wqkv"])
        for i, name in enumerate(("query", "key", "value")):
            G[a + name + ".weight"] = dwqkv[i * H:(i + 1) * H].clone()
            G[a + name + ".bias"] = dbqkv[i * H:(i + 1) * H].clone()
        dz, dz_res = dzin, ds1                                            # residual of the attention block: added inside the next LayerNorm backward
    # ---- embeddings
    demb, G["model.embeddings.LayerNorm.weight"], G["model.embeddings.LayerNorm.bias"] = \
        ops.layernorm_bwd(dz, S["emb"], S["emb_st"], P["model.embeddings.LayerNorm.weight"], dy2=dz_res)
    per_pos = ops.colsum(demb.view(n, Lp * H)).view(Lp, H)                # sum over the rows, per position
    dpos = torch.zeros_like(P["model.embeddings.position_embeddings.weight"])
    dpos[:L] = per_pos[:L]                                                # (the language token's type / position terms cancel: :192-199)
    G["model.embeddings.position_embeddings.weight"] = dpos
    G["model.embeddings.token_type_embeddings.weight"] = ops.colsum(per_pos[:L].contiguous()).reshape(1, -1)
    if lam:
        dlang = torch.zeros_like(P["lang_embeddings.weight"])
        dlang[lang] = per_pos[L]
        G["lang_embeddings.weight"] = dlang
    dt = demb.view(n, Lp, H)[:, :L].contiguous().view(n * L, H)           # layout plumbing
    # ---- input projection
    dy0 = _projector_bwd(ops, P, G, "input_projection.1.", S["pb_in"], dt)
    dx0, G["input_projection.0.weight"], G["input_projection.0.bias"] = ops.linear_bwd(dy0, S["x0"], P["input_projection.0.weight"])
    dfb, dsw, dsb = ops.gather_bwd(S["ids"].view(-1), src, dims.original_vocab_size, dx0, P["fallback_embeddings.weight"].shape[0])
    G["fallback_embeddings.weight"] = dfb
    if dims.rescale:
        # x = w * src + b on source rows: d w = sum dx * src; gather_bwd's `prod` used the raw source row
        G["in_scaler.w"], G["in_scaler.b"] = dsw.reshape(1, -1), dsb.reshape(1, -1)
    return G


# ------------------------------------------------------------------------------------------------------------------------
# The same forward on the PACKED schedule of the inference path: three of its exact levers are exact for gradients too —
# a position that cannot influence hidden[:, 0] receives a zero gradient, and a value computed once and used k times gets
# the sum of the k gradients.
#   1. pad skipping: a row keeps position 0, its non-pad positions and the language token (all-masked rows keep all L);
#   2. the input projection once per DISTINCT referenced source id (gradients of the uses are scatter-added);
#   3. the last layer computes K / V for every kept position and Q / O-proj / FFN / LayerNorms for position 0 only.
# (Lever 4, layer 0's Q/K/V per distinct (id, position) pair, is left to the inference path.)
def plan_packed(ids: torch.Tensor, pad: int, lam: int):
    """Integer plumbing on the device: which positions a row keeps.  -> dict of int32 / uint8 tensors"""
    n, L = ids.shape
    vis = ids != pad
    uniform = (~vis).all(1) & (lam == 0)              # every key masked: uniform attention over ALL L positions (eager semantics)
    keep = vis | uniform[:, None]
    keep[:, 0] = True                                 # position 0 is the query that is read out, pad or not
    if lam:
        one = torch.ones((n, 1), dtype=torch.bool, device=ids.device)
        keep, vis = torch.cat([keep, one], 1), torch.cat([vis, one], 1)
    tok_row, tok_pos = keep.nonzero(as_tuple=True)    # row-major: sorted by row, then position
    counts = keep.sum(1)
    row_offset = torch.zeros(n + 1, dtype=torch.int32, device=ids.device)
    row_offset[1:] = counts.cumsum(0)
    return dict(tok_row=tok_row, tok_pos=tok_pos.to(torch.int32), tok_key=vis[tok_row, tok_pos].to(torch.uint8).contiguous(),
                row_offset=row_offset, cls=row_offset[:-1].contiguous(), n_tokens=int(tok_row.numel()), max_len=int(counts.max()))


def forward_packed(ops: Ops, dims: HypernetDims, ln_eps: float, P, ids: torch.Tensor, src: torch.Tensor, lang: int):
    n, L = ids.shape
    lam = 1 if dims.embed_lang else 0
    Lp, H = L + lam, dims.hidden
    for name, width in (("n_embd", dims.n_embd), ("n_in_embd", dims.n_in_embd), ("hidden", H), ("intermediate", dims.intermediate)):
        if width % ops.kstep:
            raise NotImplementedError(f"{name} = {width}: the training GEMM contracts over multiples of {ops.kstep}")
    S = dict(packed=True)
    plan = plan_packed(ids, dims.pad_token_id, lam)
    S["plan"] = plan
    T, seq = plan["n_tokens"], plan["max_len"]
    # ---- lever 2: the input projection per distinct referenced id
    is_tok = plan["tok_pos"] < L
    tok_ids = ids[plan["tok_row"][is_tok], plan["tok_pos"][is_tok].long()]
    uniq, inv = torch.unique(tok_ids, return_inverse=True)
    uniq32 = uniq.to(torch.int32).contiguous()
    S["uniq"] = uniq32
    sw = P["in_scaler.w"].reshape(-1) if dims.rescale else None
    sb = P["in_scaler.b"].reshape(-1) if dims.rescale else None
    x0 = ops.gather(uniq32, src, dims.original_vocab_size, P["fallback_embeddings.weight"], sw, sb)      # [D, E_in]
    y0 = ops.gemm(x0, P["input_projection.0.weight"], P["input_projection.0.bias"])
    table, S["pb_in"] = _projector_fwd(ops, P, "input_projection.1.", y0)
    S["x0"] = x0
    d_ids = table.shape[0]
    type0 = P["model.embeddings.token_type_embeddings.weight"][0]
    pos = P["model.embeddings.position_embeddings.weight"]
    slot = torch.full((T,), d_ids, dtype=torch.int32, device=ids.device)       # the language token takes the extra row
    slot[is_tok] = inv.to(torch.int32)
    S["slot"] = slot
    if lam:
        table = torch.cat([table, (P["lang_embeddings.weight"][lang] - (type0 + pos[L]))[None, :]], 0)      # parameter-sized glue
    posadd = (type0[None, :] + pos[:Lp]).contiguous()
    emb = ops.gather_rows(posadd, plan["tok_pos"], ops.gather_rows(table, slot))        # table[slot] + (type + position)
    z, st = ops.layernorm(emb, P["model.embeddings.LayerNorm.weight"], P["model.embeddings.LayerNorm.bias"], ln_eps)
    S["emb"], S["emb_st"] = emb, st
    off, key = plan["row_offset"], plan["tok_key"]
    layers = []
    for l in range(dims.layers):
        p = f"model.encoder.layer.{l}."
        a = p + "attention.self."
        last = l == dims.layers - 1
        wqkv = torch.cat([P[a + "query.weight"], P[a + "key.weight"], P[a + "value.weight"]], 0)
        bqkv = torch.cat([P[a + "query.bias"], P[a + "key.bias"], P[a + "value.bias"]], 0)
        A = dict(z=ops.keep(z), wqkv=wqkv)
        if not last:
            qkv = ops.gemm(z, wqkv, bqkv)
            ctx, probs = ops.attention(qkv[:, :H], qkv[:, H:2 * H], qkv[:, 2 * H:], key, off, n, seq, dims.heads, H, operand=True)
            res = z
            A.update(qkv=qkv)
        else:
            # ---- lever 3: keys / values for every kept position, the query (and everything behind it) for position 0 only
            kv = ops.gemm(z, wqkv[H:], bqkv[H:])
            zc = ops.gather_rows(z, plan["cls"])
            qc = ops.gemm(zc, wqkv[:H], bqkv[:H])
            ctx, probs = ops.attention(qc, kv[:, :H], kv[:, H:], key, off, n, seq, dims.heads, H, cls_only=True, operand=True)
            res = zc
            A.update(kv=kv, zc=zc, qc=qc)
        s1 = ops.gemm(ctx, P[p + "attention.output.dense.weight"], P[p + "attention.output.dense.bias"], residual=res)
        z1, st1 = ops.layernorm(s1, P[p + "attention.output.LayerNorm.weight"], P[p + "attention.output.LayerNorm.bias"], ln_eps)
        u = ops.gemm(z1, P[p + "intermediate.dense.weight"], P[p + "intermediate.dense.bias"])
        g = ops.gelu(u, GELU_ERF, operand=True)
        s2 = ops.gemm(g, P[p + "output.dense.weight"], P[p + "output.dense.bias"], residual=z1)
        z, st2 = ops.layernorm(s2, P[p + "output.LayerNorm.weight"], P[p + "output.LayerNorm.bias"], ln_eps)
        A.update(probs=probs, ctx=ctx, s1=s1, st1=st1, z1=ops.keep(z1), u=u, g=g, s2=s2, st2=st2)
        layers.append(A)
    S["layers"] = layers
    S["seq"] = seq
    cls = z                                            # [n, H]: the last layer ran on position 0 only
    S["cls"] = cls
    S["ids"] = ids
    return _heads_fwd(ops, dims, P, S, cls, n, ids.device), S


def backward_packed(ops: Ops, dims: HypernetDims, P, S, src, lang, d_in, d_out, d_bias):
    G: Dict[str, torch.Tensor] = {}
    ids = S["ids"]
    n, L = ids.shape
    lam = 1 if dims.embed_lang else 0
    Lp, H = L + lam, dims.hidden
    plan, seq = S["plan"], S["seq"]
    T = plan["n_tokens"]
    off = plan["row_offset"]
    dev = ids.device
    dz, dz_res = _heads_bwd(ops, dims, P, S, G, n, d_in, d_out, d_bias), None    # [n, H]: gradient of hidden[:, 0]
    # (r5) what a stage saved is released as soon as its backward has run, and a gradient as soon as it has been consumed: the
    # peak of a step is then the forward's activations plus ONE stage's temporaries, not plus every stage's
    for key in ("pb_out0", "pb_out1", "h_in", "h_out", "pred_raw", "pred_out_raw"):
        S.pop(key, None)
    d_in = d_out = d_bias = None
    for l in reversed(range(dims.layers)):
        p = f"model.encoder.layer.{l}."
        a = p + "attention.self."
        A = S["layers"][l]
        last = l == dims.layers - 1
        ds2, G[p + "output.LayerNorm.weight"], G[p + "output.LayerNorm.bias"] = ops.layernorm_bwd(dz, A["s2"], A["st2"], P[p + "output.LayerNorm.weight"], dy2=dz_res)
        dz = dz_res = None
        dg, G[p + "output.dense.weight"], G[p + "output.dense.bias"] = ops.linear_bwd(ds2, A["g"], P[p + "output.dense.weight"])
        del A["s2"], A["g"]
        dz1, G[p + "intermediate.dense.weight"], G[p + "intermediate.dense.bias"] = ops.linear_bwd(dg, A["z1"], P[p + "intermediate.dense.weight"], A["u"], GELU_ERF)
        dg = None
        del A["u"], A["z1"]
        ds1, G[p + "attention.output.LayerNorm.weight"], G[p + "attention.output.LayerNorm.bias"] = \
            ops.layernorm_bwd(dz1, A["s1"], A["st1"], P[p + "attention.output.LayerNorm.weight"], dy2=ds2)        # + the residual of the FFN
        dz1 = ds2 = None
        del A["s1"]
        dctx, G[p + "attention.output.dense.weight"], G[p + "attention.output.dense.bias"] = ops.linear_bwd(ds1, A["ctx"], P[p + "attention.output.dense.weight"])
        del A["ctx"]
        wqkv = A["wqkv"]
        if not last:
            qkv = A["qkv"]
            dqkv = ops.new(T, 3 * H)
            ops.attention_bwd(dctx, qkv[:, :H], qkv[:, H:2 * H], qkv[:, 2 * H:], A["probs"], off, n, seq, dims.heads, H,
                              dqkv[:, :H], dqkv[:, H:2 * H], dqkv[:, 2 * H:])
            dctx = qkv = None
            del A["qkv"], A["probs"]
            dzin, dwqkv, dbqkv = ops.linear_bwd(dqkv, A["z"], wqkv)
            dqkv = None
            dz, dz_res = dzin, ds1                                              # (added inside the next LayerNorm backward)
        else:
            kv, qc = A["kv"], A["qc"]
            dqc, dkv = ops.new(n, H), ops.new(T, 2 * H)
            ops.attention_bwd(dctx, qc, kv[:, :H], kv[:, H:], A["probs"], off, n, seq, dims.heads, H, dqc, dkv[:, :H], dkv[:, H:], cls_only=True)
            dzc, dwq, dbq = ops.linear_bwd(dqc, A["zc"], wqkv[:H])
            dzc = ops.add(dzc, ds1)                                             # the residual of the attention block, position 0 only
            dz, dwkv, dbkv = ops.linear_bwd(dkv, A["z"], wqkv[H:])
            ops.scatter_add_rows(dz, plan["cls"], dzc)
            dz_res = None
            dwqkv, dbqkv = torch.cat([dwq, dwkv], 0), torch.cat([dbq, dbkv], 0)
        for i, name in enumerate(("query", "key", "value")):
            G[a + name + ".weight"] = dwqkv[i * H:(i + 1) * H].clone()
            G[a + name + ".bias"] = dbqkv[i * H:(i + 1) * H].clone()
        dwqkv = dbqkv = wqkv = None
        A.clear()                                                               # this layer's activations are not needed again
    demb, G["model.embeddings.LayerNorm.weight"], G["model.embeddings.LayerNorm.bias"] = \
        ops.layernorm_bwd(dz, S["emb"], S["emb_st"], P["model.embeddings.LayerNorm.weight"], dy2=dz_res)
    per_pos = torch.zeros((Lp, H), dtype=torch.float32, device=dev)
    ops.scatter_add_rows(per_pos, plan["tok_pos"], demb)
    dpos = torch.zeros_like(P["model.embeddings.position_embeddings.weight"])
    dpos[:L] = per_pos[:L]                                                     # (the language token's type / position terms cancel)
    G["model.embeddings.position_embeddings.weight"] = dpos
    G["model.embeddings.token_type_embeddings.weight"] = ops.colsum(per_pos[:L].contiguous()).reshape(1, -1)
    d_ids = S["uniq"].numel()
    dtable = torch.zeros((d_ids + lam, H), dtype=torch.float32, device=dev)
    ops.scatter_add_rows(dtable, S["slot"], demb)                               # lever 2 backwards: the uses of an id add up
    if lam:
        dlang = torch.zeros_like(P["lang_embeddings.weight"])
        dlang[lang] = dtable[d_ids]
        G["lang_embeddings.weight"] = dlang
    dy0 = _projector_bwd(ops, P, G, "input_projection.1.", S["pb_in"], dtable[:d_ids].contiguous())
    dx0, G["input_projection.0.weight"], G["input_projection.0.bias"] = ops.linear_bwd(dy0, S["x0"], P["input_projection.0.weight"])
    dfb, dsw, dsb = ops.gather_bwd(S["uniq"], src, dims.original_vocab_size, dx0, P["fallback_embeddings.weight"].shape[0])
    G["fallback_embeddings.weight"] = dfb
    if dims.rescale:
        G["in_scaler.w"], G["in_scaler.b"] = dsw.reshape(1, -1), dsb.reshape(1, -1)
    return G


class HypernetFunction(torch.autograd.Function):
    """(target_surface_forms, source_embeddings, lang_index, *parameters) -> (pred_in, pred_out | empty, bias); gradients for
    the parameters only (the reference trains the hypernetwork against frozen source embeddings)."""

    @staticmethod
    def forward(ctx, dims, ln_eps, names, packed, precision, ids, src, lang, *params):
        ops = Ops(src.device, precision)
        P = {n: p.detach().float().contiguous() for n, p in zip(names, params)}
        with torch.no_grad(), torch.cuda.device(src.device):          # (the primitives launch on the CURRENT device and stream)
            fwd = forward_packed if packed else forward_train
            (pred_in, pred_out, bias), S = fwd(ops, dims, ln_eps, P, ids, src, int(lang))
        ctx.dims, ctx.names, ctx.lang, ctx.ops = dims, names, int(lang), ops
        ctx.P, ctx.S, ctx.src = P, S, src
        ctx.has_out = pred_out is not None
        if pred_out is None:
            pred_out = pred_in.new_zeros((0,))
        return pred_in, pred_out, bias

    @staticmethod
    def backward(ctx, d_in, d_out, d_bias):
        with torch.no_grad(), torch.cuda.device(ctx.src.device):
            bwd = backward_packed if ctx.S.get("packed") else backward_train
            G = bwd(ctx.ops, ctx.dims, ctx.P, ctx.S, ctx.src, ctx.lang, d_in, d_out if ctx.has_out else None, d_bias)
        grads = []
        for name, p in zip(ctx.names, ctx.P.values()):
            g = G.get(name)
            grads.append(None if g is None else g.reshape(p.shape))
        ctx.S = None
        if ctx.ops.precision == "f16":
            # half operands have 5 exponent bits and this path does no loss scaling: a gradient (or activation) beyond 65 504
            # becomes inf in an operand and NaN in the result.  Said loudly (one parameter-sized reduction and a host read per
            # step, f16 only) instead of handing NaNs to the optimizer; bf16 operands have the fp32 range.
            worst = torch.stack([g.detach().abs().max() for g in grads if g is not None and g.numel()]).max()
            if not bool(torch.isfinite(worst)):
                raise _lib.RangeError("zett_amd: a gradient left the range of f16 MFMA operands (non-finite parameter gradients); scale the loss "
                                      "down or use model.train_precision = 'bf16' (fp32 exponent range)")
        return (None, None, None, None, None, None, None, None, *grads)


def _check_indices(dims: HypernetDims, ids: torch.Tensor, src: torch.Tensor, lang: int) -> None:
    """The index errors of the reference's F.embedding calls (modeling_hypernet.py:179-188, 192-211) and of the inference
    path (ZETT_E_INDEX), raised BEFORE a kernel reads — and, in the backward, atomically writes — through a bad id.  One
    min/max reduction over the id matrix and one host read per training step."""
    if ids.dim() != 2:
        raise ValueError("target_surface_forms must be [n_tokens, surface_maxlen]")
    lam = 1 if dims.embed_lang else 0
    if ids.shape[1] + lam > dims.max_positions:
        raise IndexError(f"sequence {ids.shape[1]} (+{lam} language token) exceeds position_embeddings ({dims.max_positions} rows)")
    v0, top = dims.original_vocab_size, dims.original_vocab_size + dims.n_extra
    if src.dim() != 2 or src.shape[1] != dims.n_in_embd:
        raise ValueError(f"source_embeddings must be [V, {dims.n_in_embd}], got {tuple(src.shape)}")
    if src.shape[0] < v0:
        raise IndexError(f"source_embeddings has {src.shape[0]} rows, config.original_vocab_size is {v0}")
    if ids.numel():
        lo, hi = (int(v) for v in torch.stack([ids.min(), ids.max()]).tolist())
        if lo < 0 or hi >= top:
            raise IndexError(f"target_surface_forms holds an id outside [0, {top}) (original_vocab_size {v0} + {dims.n_extra} fallback rows): "
                             f"min {lo}, max {hi}")
    if dims.embed_lang and not 0 <= lang < dims.n_langs:
        raise IndexError(f"lang_index {lang} outside [0, {dims.n_langs})")


def differentiable_forward(model, target_surface_forms: torch.Tensor, source_embeddings: torch.Tensor, lang_index: int, packed: bool = True,
                           precision: str = "f32"):
    """The forward of `model` (a zett_amd.hypernet.ZettHypernet) with gradients to its parameters.  packed = True (default):
    the schedule of the inference path (pad skipping, input projection per distinct id, position-0-only last layer);
    False: the reference's dense layout, every position computed — same outputs and gradients to fp32 round-off.
    precision: "f32" (exact fp32 MFMA) or "bf16" / "f16" (16-bit MFMA operands in every forward / dgrad / wgrad contraction,
    fp32 accumulation, fp32 parameters, activations and gradients)."""
    # (parameters of the checkpoint contract that the forward never reads stay outside the graph: a gradient hook that waits for
    # every input of the Function — DistributedDataParallel's bucket logic — would wait for them for ever)
    _check_indices(model.dims, target_surface_forms, source_embeddings, int(lang_index))
    names = [n for n in weight_shapes(model.dims) if n not in NEVER_READ]
    params = dict(model.named_parameters())
    tensors = [params[n] for n in names]
    src = source_embeddings if source_embeddings.dtype in _SRC_DTYPES else source_embeddings.float()
    pred_in, pred_out, bias = HypernetFunction.apply(model.dims, model._ln_eps_encoder, tuple(names), bool(packed), str(precision), target_surface_forms, src.contiguous(),
                                                     int(lang_index), *tensors)
    return pred_in, (pred_out if model.dims.separate_out else None), bias
