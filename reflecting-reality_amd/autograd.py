"""Reverse-mode differentiation of the HIP forward path: a tape of backward closures over the libmfhip kernels.

The reference trains through ATen autograd (examples/brushnet/train_brushnet_mirror.py:1459 `accelerator.backward(loss)`).
Here the forward kernels are hand-written, so their vector-Jacobian products are too: while `ops.TAPE` is set, every
layer-level operator (ops.conv2d / linear / groupnorm / layernorm / attention / geglu / add ...) appends a closure that,
given the gradient of its output, launches the backward kernels of csrc/train.hip (and mf_gemm_conv for the data
gradients) and hands gradients on to its inputs.  Nothing here does arithmetic in PyTorch: torch owns the buffers.

Gradients are keyed by the storage of the tensor they belong to, so the NCHW-shaped channels-last *views* that cross the
BrushNet -> UNet boundary (models.to_nchw_view / from_nchw) carry their gradient back without any layout copy.
Parameters live in flat fp32 arenas (weights, gradients, Adam moments): a `Param` is a pair of views.
"""
from __future__ import annotations

from typing import Callable, Dict, List, Optional

import os

import torch

from . import hip

FUSE_GRAD_ADD = os.environ.get("MFHIP_NO_FUSED_GRAD_ADD", "0") != "1"   # developer A/B: norm backwards add the residual's gradient
CAST_COLSUM = os.environ.get("MFHIP_NO_CAST_COLSUM", "0") != "1"  # developer A/B: mf_cast_bf16 + mf_colsum passes instead
GN_GRAD_ACC = os.environ.get("MFHIP_NO_GN_ACC", "0") != "1"      # developer A/B: per-image partials + mf_colsum instead


class Param:
    """A tensor of a model's flat arenas: `data` (fp32 master weight, in the kernels' layout) and `grad` (same shape; None
    when the parameter is frozen)."""
    __slots__ = ("name", "data", "grad", "fresh")

    def __init__(self, name: str, data: torch.Tensor, grad: Optional[torch.Tensor]):
        self.name, self.data, self.grad = name, data, grad
        self.fresh = False       # a conv / linear weight whose gradient was NOT cleared this step: its first weight gradient writes


def _key(t: torch.Tensor):
    return t.untyped_storage().data_ptr()


class Tape:
    def __init__(self, code: int):
        self.code = code                       # MF_F32, MF_F16X3 or MF_BF16X1: the contraction mode of the backward GEMMs
        self.ops: List[Callable[[], None]] = []
        self.grads: Dict[int, torch.Tensor] = {}
        self.stop = set()                      # storages that need no gradient (the batch's inputs)
        self.on_param_grad: Optional[Callable[[Param], None]] = None   # gradient-bucket hook (distributed.GradBuckets)
        self.dgrad_rebuilt: list = []          # trainable weights whose data-gradient layout this backward had to rebuild
        # bf16 gradients whose producer already added their column sums (a bias gradient), by address; the tensor itself is kept
        # as the value, so the caching allocator cannot hand that address to a later gradient while the entry exists, and the
        # consumer removes the entry when it takes the gradient (a stale address must never read as 'done')
        self.colsum_done: Dict[int, torch.Tensor] = {}

    # ---- bookkeeping ---------------------------------------------------------------------------------------
    def no_grad(self, *ts: Optional[torch.Tensor]) -> None:
        for t in ts:
            if t is not None:
                self.stop.add(_key(t))

    def needs(self, t: Optional[torch.Tensor]) -> bool:
        return t is not None and _key(t) not in self.stop

    def record(self, fn: Callable[[], None]) -> None:
        self.ops.append(fn)

    def add(self, t: Optional[torch.Tensor], g: torch.Tensor) -> None:
        """Accumulate g (laid out like t's storage: same numel) into t's gradient.  Never in place: the same g may be
        handed to several consumers (a residual fan-out)."""
        if not self.needs(t):
            return
        if t.numel() != g.numel():
            raise hip.MfhipError(f"gradient of {tuple(t.shape)} has {g.numel()} elements")
        k = _key(t)
        old = self.grads.get(k)
        self.grads[k] = g if old is None else hip.axpby_n([old, g.view(old.shape)], [1.0, 1.0])

    def peek(self, t: Optional[torch.Tensor]) -> Optional[torch.Tensor]:
        """The gradient t has accumulated so far (or None): a backward kernel that can add it into its own output (put) saves
        the separate accumulation pass add() would launch."""
        if not (FUSE_GRAD_ADD and self.needs(t)):
            return None
        g = self.grads.get(_key(t))
        return g if (g is not None and g.is_contiguous() and g.numel() == t.numel()) else None

    def put(self, t: Optional[torch.Tensor], g: torch.Tensor, fused: bool) -> None:
        """fused: g already contains what peek(t) returned — replace t's gradient by it; otherwise accumulate as add() does."""
        if not fused:
            return self.add(t, g)
        if self.needs(t):
            self.grads[_key(t)] = g

    def add_cols(self, view: torch.Tensor, g_rows: torch.Tensor, n: int, segs: int) -> None:
        """view = parent[:, a:b] (a column slice of a 2-D fp32 tensor): add the per-segment column sums of g_rows
        ([segs * rows_per_seg][n]) into the matching columns of the parent's dense gradient."""
        tgt = self.cols_target(view)
        if tgt is not None:
            hip.colsum(g_rows, n, segs=segs, out=tgt, ldo=view.stride(0), accumulate=True)

    def cols_target(self, view: torch.Tensor) -> Optional[torch.Tensor]:
        """The slice of the parent's dense gradient that matches `view` (created zeroed on first use); None when not needed."""
        if not self.needs(view):
            return None
        k = _key(view)
        dense = self.grads.get(k)
        numel = view.untyped_storage().nbytes() // 4
        if dense is None:
            dense = torch.zeros(numel, dtype=torch.float32, device=view.device)
            self.grads[k] = dense
        return dense.as_strided(view.shape, view.stride(), view.storage_offset())

    def take(self, t: torch.Tensor) -> Optional[torch.Tensor]:
        g = self.grads.pop(_key(t), None)
        return None if g is None else g.view(-1)

    def param_grad_done(self, p: Optional[Param]) -> None:
        if p is not None and p.grad is not None and self.on_param_grad is not None:
            self.on_param_grad(p)

    def backward(self) -> None:
        for fn in reversed(self.ops):
            fn()
        self.ops.clear()
        self.colsum_done.clear()
        self.grads.clear()


def checkpoint(fn: Callable[[], torch.Tensor]) -> torch.Tensor:
    """Activation recomputation for one block (torch.utils.checkpoint as the reference's blocks use it under
    `--gradient_checkpointing`: unet_2d_blocks.py:1167-1196, 2597-2622; brushnet.py:674-676; train_brushnet_mirror.py:1153-1155).

    `fn()` runs the block's forward and returns its one output tensor.  With a tape set, the forward runs on a THROW-AWAY tape
    (the same operators, kernels and code paths as a recorded forward: everything that asks `ops.TAPE is not None` sees a tape),
    whose closures — and with them every intermediate activation of the block — are dropped at once; the real tape gets ONE
    closure that holds the block's inputs (through fn) and, in the backward pass, runs fn again on a sub-tape that shares the
    real tape's gradient table, seeds the recomputed output with the output's gradient and plays the sub-tape back.  Same
    kernels on the same values in the same order: gradients are bit-identical to the un-checkpointed step (tests)."""
    from . import ops
    tape = ops.TAPE
    if tape is None:
        return fn()
    scratch = Tape(tape.code)
    scratch.stop = tape.stop
    ops.TAPE = scratch
    try:
        out = fn()
    finally:
        ops.TAPE = tape
    scratch.ops.clear()
    scratch.colsum_done.clear()
    del scratch

    def bwd():
        g = tape.take(out)
        if g is None:
            return
        sub = Tape(tape.code)
        sub.grads, sub.stop, sub.on_param_grad = tape.grads, tape.stop, tape.on_param_grad
        sub.dgrad_rebuilt, sub.colsum_done = tape.dgrad_rebuilt, tape.colsum_done
        prev = ops.TAPE
        ops.TAPE = sub
        try:
            out2 = fn()
        finally:
            ops.TAPE = prev
        if out2.numel() != g.numel():
            raise hip.MfhipError("checkpoint: the recomputed block output differs in size from the recorded one")
        sub.grads[_key(out2)] = g
        for f in reversed(sub.ops):
            f()
        sub.ops.clear()

    tape.record(bwd)
    return out


# =====================================================================================================================
# backward closures
# =====================================================================================================================
def _dgrad_weight(cw, tape) -> torch.Tensor:
    """[Ctot][taps (flipped)][N]: the weight of the data-gradient convolution, laid out by mf_transpose once per
    optimizer step (the master weight changes every step) into buffers the weight keeps (no allocation after the first
    step: training.train_step rebuilds them on a side stream under the forward pass).  `tape`: a Tape or its compute code."""
    code = tape if isinstance(tape, int) else tape.code
    gen = getattr(cw, "_wd_gen", None)
    from . import ops
    tok = ops.CAPTURE_TOKEN          # capturing a training graph: the rebuild of a weight that trains must be IN the graph
    recapture = tok is not None and cw.p_w is not None and cw.p_w.grad is not None and getattr(cw, "_wd_tok", None) is not tok
    if gen != cw.generation() or getattr(cw, "_wd", None) is None or getattr(cw, "_wd_code", None) != code or recapture:
        cw._wd_tok = tok
        taps, n, ct = cw.kh * cw.kw, cw.n, cw.cin_pad
        if code == hip.MF_BF16X1 and cw.fast16() and n % 8 == 0:
            # MF_BF16X1 on pre-rounded copies (ops.ConvWeight.fast16): the data gradient only ever reads the bf16 layout
            wd16 = getattr(cw, "_wd16", None)
            if wd16 is None:
                wd16 = cw._wd16 = torch.empty(ct, taps * n, dtype=torch.bfloat16, device=cw.w.device)
            hip.transpose(cw.w, n, ct, nz=taps, ldx=taps * ct, ldy=taps * n, zsx=ct, zsy=-n, out=wd16, y_offset=(taps - 1) * n)
            cw._wd, cw._wd_split, cw._wd_ld, cw._wd_gen, cw._wd_code = wd16, 0, taps * n, cw.generation(), code
            if not isinstance(tape, int) and cw.p_w is not None and cw.p_w.grad is not None:
                tape.dgrad_rebuilt.append(cw)
            return cw._wd
        cw._wd16 = None
        wd = getattr(cw, "_wd_f32", None)
        if wd is None:
            wd = cw._wd_f32 = torch.empty(ct, taps * n, dtype=torch.float32, device=cw.w.device)
        # per tap t: x_t[n][c] = w[n][t*ct + c] (ld taps*ct)  ->  y[c][(taps-1-t)*n + n'] (ld taps*n): zsy < 0 flips the taps
        hip.transpose(cw.w, n, ct, nz=taps, ldx=taps * ct, ldy=taps * n, zsx=ct, zsy=-n, out=wd, y_offset=(taps - 1) * n)
        cw._wd_split, cw._wd_ld = 0, taps * n
        if ops.PRESPLIT_TRAINING and code in (hip.MF_F16X3, hip.MF_BF16X3):
            # (hi, lo) halves packed once per weight generation: the pre-split GEMM forms (ConvWeight.operand)
            prev = getattr(cw, "_wd", None)
            wd, cw._wd_ld = hip.split_pack(wd, code, out=prev if (prev is not None and prev.dtype != torch.float32) else None)
            cw._wd_split = 1
        cw._wd, cw._wd_gen, cw._wd_code = wd, cw.generation(), code
        if not isinstance(tape, int) and cw.p_w is not None and cw.p_w.grad is not None:
            tape.dgrad_rebuilt.append(cw)              # a weight that trains: next step's prefetch list
    return cw._wd


def record_conv(tape: Tape, x, x1, cw, out, *, batch, h_in, w_in, h_out, w_out, stride, pad_t, pad_l, upsample, temb, res0, res1,
                alpha, act, x16=None) -> None:
    """out = alpha * (conv(cat(x, x1)) + bias + temb[b]) + res0 + res1  (mf_gemm_conv; linear = 1x1 over batch rows)."""
    if act != hip.ACT_NONE:
        raise hip.MfhipError("training: fused activations are not differentiated; apply ops.silu / ops.geglu as their own op")
    c0 = x.shape[-1]
    c1 = x1.shape[-1] if x1 is not None else 0
    n = cw.n

    def bwd():
        g = tape.take(out)
        if g is None:
            return
        g = g.view(-1, n)
        if g.dtype == torch.bfloat16:
            # a gradient its producer already rounded (the bf16 tensors of the MF_BF16X1 mode: q / k / v and FeedForward's hidden
            # tensor): it IS the 16-bit operand; a bias gradient was added by the producer from the same pass
            want_b = cw.p_bias is not None and cw.p_bias.grad is not None
            bias_done = tape.colsum_done.pop(g.data_ptr(), None) is not None
            if (res0 is not None or res1 is not None or temb is not None or alpha != 1.0 or x16 is None or n % 8 or stride != 1
                    or (want_b and not bias_done)):
                raise hip.MfhipError("training: a bf16 output gradient needs a plain fast-path GEMM (no residual / temb / scale, n % 8 == 0)")
            if want_b:
                tape.param_grad_done(cw.p_bias)
            return _conv_grads(tape, cw, x, x1, c0, c1, n, None, g, x16, batch, h_in, w_in, h_out, w_out, stride, pad_t, pad_l, upsample)
        tape.add(res0, g)
        tape.add(res1, g)
        gc = g if alpha == 1.0 else hip.axpby_n([g], [alpha])
        # x16 = the bf16 copies the forward GEMM ran on (MF_BF16X1 on pre-rounded operands): the gradient is rounded to bf16 ONCE
        # here, and both the weight gradient and the data gradient read 16-bit operands; the bias / time-embedding gradients
        # (column sums of the same tensor) come out of that one read (mf_cast_bf16_colsum)
        want_b = cw.p_bias is not None and cw.p_bias.grad is not None
        gc16, sums_done = None, False
        if x16 is not None and n % 8 == 0:
            t_tgt = tape.cols_target(temb) if temb is not None else None
            if (want_b or t_tgt is not None) and CAST_COLSUM and (t_tgt is None or batch <= 32):
                gc16 = hip.cast_bf16_colsum(gc.contiguous(), n, segs=batch if t_tgt is not None else 1, seg_out=t_tgt,
                                            ldo=temb.stride(0) if t_tgt is not None else None,
                                            tot_out=cw.p_bias.grad.view(n) if want_b else None)
                sums_done = True
                if want_b:
                    tape.param_grad_done(cw.p_bias)
            else:
                gc16 = hip.cast_bf16(gc.contiguous())
        if not sums_done:
            if temb is not None:
                tape.add_cols(temb, gc, n, batch)
            if want_b:
                hip.colsum(gc, n, out=cw.p_bias.grad.view(1, n), accumulate=True)
                tape.param_grad_done(cw.p_bias)
        _conv_grads(tape, cw, x, x1, c0, c1, n, gc, gc16, x16, batch, h_in, w_in, h_out, w_out, stride, pad_t, pad_l, upsample)

    tape.record(bwd)


def _conv_grads(tape: Tape, cw, x, x1, c0, c1, n, gc, gc16, x16, batch, h_in, w_in, h_out, w_out, stride, pad_t, pad_l, upsample) -> None:
    """Weight and data gradients of one conv / linear from its output gradient gc (fp32 [M, n]; None when only the rounded copy
    exists) and / or gc16 (its bf16 copy, MF_BF16X1 on pre-rounded operands)."""
    if cw.p_w is not None and cw.p_w.grad is not None:
        acc = not cw.p_w.fresh           # fresh: the arena still holds the previous step's values here — write, do not add
        cw.p_w.fresh = False
        if gc16 is not None:
            hip.conv_wgrad(x16[0], gc16, cw.p_w.grad, code=hip.MF_BF16, c0=c0, x1=x16[1], c1=c1, batch=batch, h_in=h_in, w_in=w_in,
                           h_out=h_out, w_out=w_out, kh=cw.kh, kw=cw.kw, stride=stride, pad_t=pad_t, pad_l=pad_l, upsample=upsample, n=n,
                           accumulate=acc)
        else:
            hip.conv_wgrad(x, gc, cw.p_w.grad, code=tape.code, c0=c0, x1=x1, c1=c1, batch=batch, h_in=h_in, w_in=w_in, h_out=h_out,
                           w_out=w_out, kh=cw.kh, kw=cw.kw, stride=stride, pad_t=pad_t, pad_l=pad_l, upsample=upsample, n=n,
                           accumulate=acc)
        tape.param_grad_done(cw.p_w)
    if not (tape.needs(x) or tape.needs(x1)):
        return
    wd = _dgrad_weight(cw, tape)
    taps = cw.kh * cw.kw
    # MF_BF16X1 on pre-rounded copies: the gradient is rounded to bf16 once, the transposed weight once per step, and the data
    # gradient runs on the bf16 LDS-DMA kernels (same arithmetic per product as the in-register rounding)
    fast = n % 8 == 0 and (taps * n) % 8 == 0 and cw.fast16() and getattr(cw, "_wd16", None) is not None
    if gc is None and not (fast and stride == 1):
        raise hip.MfhipError("training: a bf16 output gradient needs the bf16 data-gradient path (stride 1, channels % 8 == 0)")
    a, gh, gw = gc, h_out, w_out
    if stride == 2:
        a = hip.zero_insert2x(gc.view(batch, h_out, w_out, n))
        gh, gw = 2 * h_out, 2 * w_out
    elif stride != 1:
        raise hip.MfhipError("training: only stride 1 / 2 convolutions are differentiated")
    hu, wu = (2 * h_in, 2 * w_in) if upsample else (h_in, w_in)
    outs = []
    off = 0
    a16 = (gc16 if (gc16 is not None and stride == 1) else hip.cast_bf16(a.contiguous())) if fast else None
    for seg, cs in ((x, c0), (x1, c1)):
        if seg is None:
            continue
        if tape.needs(seg):
            dx = torch.empty(batch, hu, wu, cs, dtype=torch.float32, device=x.device)
            # what the tensor's other consumers left so far rides the GEMM epilogue as its fp32 residual (no accumulation pass)
            acc = tape.peek(seg) if (not upsample and cs % 8 == 0) else None
            ra = acc.view(-1, cs) if acc is not None else None
            if fast:
                hip.gemm_conv(a16, cw._wd16[off:off + cs], dx, dtype=hip.MF_BF16, w_split=0, c0=n, lda0=n, batch=batch, h_in=gh, w_in=gw,
                              h_out=hu, w_out=wu, kh=cw.kh, kw=cw.kw, stride=1, pad_t=cw.kh - 1 - pad_t, pad_l=cw.kw - 1 - pad_l,
                              ldw=taps * n, n=cs, res0=ra)
            else:
                hip.gemm_conv(a, wd[off:off + cs], dx, dtype=tape.code, w_split=cw._wd_split, c0=n, lda0=n, batch=batch, h_in=gh, w_in=gw,
                              h_out=hu, w_out=wu, kh=cw.kh, kw=cw.kw, stride=1, pad_t=cw.kh - 1 - pad_t, pad_l=cw.kw - 1 - pad_l,
                              ldw=cw._wd_ld, n=cs, res0=ra)
            outs.append((seg, hip.sumpool2x2(dx) if upsample else dx, acc is not None))
        off += cs
    for seg, dx, fused in outs:
        tape.put(seg, dx, fused)


def record_groupnorm(tape: Tape, x0, x1, p_gamma: Param, p_beta: Param, out, groups: int, eps: float, silu: bool, stats=None) -> None:
    def bwd():
        g = tape.take(out)
        if g is None:
            return
        want = p_gamma.grad is not None
        a0, a1 = tape.peek(x0), (tape.peek(x1) if x1 is not None else None)
        dx0, dx1, dg, db = hip.groupnorm_bwd(x0, g.view(out.shape), p_gamma.data, p_beta.data, groups=groups, eps=eps, silu=silu,
                                             x1=x1, want_param_grads=want, grad_acc=(p_gamma.grad, p_beta.grad) if (want and GN_GRAD_ACC) else None,
                                             add0=a0, add1=a1, stats=stats)
        if want:
            if dg is not None:                     # small maps: per-image partials (the streaming form adds into the arena itself)
                c = dg.shape[1]
                hip.colsum(dg, c, out=p_gamma.grad.view(1, c), accumulate=True)
                hip.colsum(db, c, out=p_beta.grad.view(1, c), accumulate=True)
            tape.param_grad_done(p_gamma)
            tape.param_grad_done(p_beta)
        tape.put(x0, dx0, a0 is not None)        # fused: dx already holds what the residual path left (peek above)
        if x1 is not None:
            tape.put(x1, dx1, a1 is not None)

    tape.record(bwd)


def record_layernorm(tape: Tape, x, p_gamma: Param, p_beta: Param, out, eps: float) -> None:
    def bwd():
        g = tape.take(out)
        if g is None:
            return
        want = p_gamma.grad is not None
        ax = tape.peek(x)
        dx, dg, db = hip.layernorm_bwd(x, g.view(x.shape), p_gamma.data, eps, want_param_grads=want, add=ax)
        if want:
            c = dg.shape[1]
            hip.colsum(dg, c, out=p_gamma.grad.view(1, c), accumulate=True)
            hip.colsum(db, c, out=p_beta.grad.view(1, c), accumulate=True)
            tape.param_grad_done(p_gamma)
            tape.param_grad_done(p_beta)
        tape.put(x, dx, ax is not None)

    tape.record(bwd)


def record_pointwise(tape: Tape, inputs, out, fn) -> None:
    """out = f(inputs...) elementwise; fn(g) returns one gradient per input (or None)."""
    def bwd():
        g = tape.take(out)
        if g is None:
            return
        for t, gi in zip(inputs, fn(g)):
            if gi is not None:
                tape.add(t, gi)

    tape.record(bwd)


def record_attention(tape: Tape, q, k, v, out, heads: int, skv: int, scale: float, forward_probs) -> None:
    """out[b][i][h*d:] = softmax(q_h k_h^T * scale) v_h  with q [B, Sq, C], k / v [B, Skv, C] (contiguous fp32).
    Backward recomputes P with the forward's kernels (`forward_probs`), then five strided-batched GEMMs and the
    transposes that bring their operands into the kernel's NT form; P is never kept between forward and backward."""
    b, sq, c = q.shape
    d = c // heads
    z = b * heads

    def bwd():
        g = tape.take(out)
        if g is None:
            return
        g = g.view(b, sq, c)
        code = tape.code
        p = forward_probs()                                      # [z][sq][ld], pad columns zero
        ld = p.shape[-1]
        dev = q.device
        # dP = dO v^T
        dp = torch.empty(z, sq, ld, dtype=torch.float32, device=dev)
        hip.gemm_conv(g, v, dp, dtype=code, c0=d, lda0=c, batch=sq, h_in=1, w_in=1, h_out=1, w_out=1, ldw=c, n=skv, ldc=ld, nz=z,
                      zdiv=heads, a_zs=(sq * c, d), w_zs=(skv * c, d), o_zs=(heads * sq * ld, sq * ld), splitk=1)
        ds = hip.softmax_bwd(p, dp, skv, scale)                  # [z][sq][ld], pad columns zero
        del dp
        # dV = P^T dO   (A = P^T [skv][sq], W = dO^T [d][sq])
        pt = hip.transpose(p, sq, skv, nz=z, ldx=ld, ldy=sq, zsx=sq * ld, zsy=skv * sq)            # [z][skv][sq]
        del p
        gt = hip.transpose(g, sq, c, nz=b, ldx=c, ldy=sq, zsx=sq * c, zsy=c * sq)                   # [b][c][sq]
        dv = torch.empty(b, skv, c, dtype=torch.float32, device=dev)
        hip.gemm_conv(pt, gt, dv, dtype=code, c0=sq, lda0=sq, batch=skv, h_in=1, w_in=1, h_out=1, w_out=1, ldw=sq, n=d, ldc=c, nz=z,
                      zdiv=heads, a_zs=(heads * skv * sq, skv * sq), w_zs=(c * sq, d * sq), o_zs=(skv * c, d), splitk=1)
        del pt, gt
        # dQ = dS k   (W = k^T [d][ld], pad columns zero)
        kt = torch.zeros(b, c, ld, dtype=torch.float32, device=dev)
        hip.transpose(k, skv, c, nz=b, ldx=c, ldy=ld, zsx=skv * c, zsy=c * ld, out=kt)
        dq = torch.empty(b, sq, c, dtype=torch.float32, device=dev)
        hip.gemm_conv(ds, kt, dq, dtype=code, c0=ld, lda0=ld, batch=sq, h_in=1, w_in=1, h_out=1, w_out=1, ldw=ld, n=d, ldc=c, nz=z,
                      zdiv=heads, a_zs=(heads * sq * ld, sq * ld), w_zs=(c * ld, d * ld), o_zs=(sq * c, d), splitk=1)
        del kt
        # dK = dS^T q  (A = dS^T [skv][sq], W = q^T [d][sq])
        dst = hip.transpose(ds, sq, skv, nz=z, ldx=ld, ldy=sq, zsx=sq * ld, zsy=skv * sq)           # [z][skv][sq]
        del ds
        qt = hip.transpose(q, sq, c, nz=b, ldx=c, ldy=sq, zsx=sq * c, zsy=c * sq)                   # [b][c][sq]
        dk = torch.empty(b, skv, c, dtype=torch.float32, device=dev)
        hip.gemm_conv(dst, qt, dk, dtype=code, c0=sq, lda0=sq, batch=skv, h_in=1, w_in=1, h_out=1, w_out=1, ldw=sq, n=d, ldc=c, nz=z,
                      zdiv=heads, a_zs=(heads * skv * sq, skv * sq), w_zs=(c * sq, d * sq), o_zs=(skv * c, d), splitk=1)
        tape.add(q, dq)
        tape.add(k, dk)
        tape.add(v, dv)

    tape.record(bwd)


def record_attention_flash_bf16(tape: Tape, q, k, v, out16, lse, heads: int, scale: float, q16, k16, v16) -> None:
    """Backward of the bf16 flash forward of the bf16x1 mode (ops.attention_train): mf_attention_bwd_bf16 on single bf16 planes.
    q / k / v are the fp32 tensors the gradients are keyed on; q16 / k16 / v16 their rounded copies made for the forward; the
    transposed operands are rounded by the transposes themselves."""
    from . import ops
    b, sq, c = q.shape
    skv = k.shape[1]

    def bwd():
        g = tape.take(out16)
        if g is None:
            return
        g = g.view(b, sq, c).contiguous()
        ldq, ldk = (sq + 7) // 8 * 8, (skv + 7) // 8 * 8
        if (c // heads) % 8 == 0 and c <= 2048:
            dd, g16 = hip.rowdot_heads_cast(g, out16, heads)            # D and the rounded dO from one read of dO
        else:
            dd, g16 = hip.rowdot_heads(g, out16, heads), hip.cast_bf16(g)
        qt = ops.transpose_tokens(q, ldq, torch.bfloat16)
        kt = ops.transpose_tokens(k, ldk, torch.bfloat16)
        gt = ops.transpose_tokens(g16, ldq, torch.bfloat16)
        dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
        hip.attention_bwd_bf16(q16, k16, v16, g16, qt, kt, gt, lse, dd, dq, dk, dv, heads=heads, scale=scale)
        tape.add(q, dq)
        tape.add(k, dk)
        tape.add(v, dv)

    tape.record(bwd)


def record_attention_flash(tape: Tape, q, k, v, out, lse, heads: int, scale: float, q_planes, k_planes) -> None:
    """Backward of the flash forward (ops.attention_train on the split-precision path): mf_attention_bwd_f16x3 recomputes P tile by
    tile from q, k and the forward's row statistics `lse`; dK / dV and dQ are two passes of one kernel, no atomics, nothing of
    size Sq x Skv is written.  The (hi, lo) planes of q and k made for the forward are kept; v, dO and the transposed operands
    are split here."""
    from . import ops
    b, sq, c = q.shape
    skv = k.shape[1]

    def bwd():
        g = tape.take(out)
        if g is None:
            return
        g = g.view(b, sq, c).contiguous()
        dd = hip.rowdot_heads(g, out, heads)
        ldq, ldk = (sq + 7) // 8 * 8, (skv + 7) // 8 * 8
        vs, gs = hip.split_halves(v.contiguous()), hip.split_halves(g)
        qt = hip.split_halves(ops.transpose_tokens(q, ldq))
        kt = hip.split_halves(ops.transpose_tokens(k, ldk))
        gt = hip.split_halves(ops.transpose_tokens(g, ldq))
        dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
        hip.attention_bwd_f16x3(q_planes, k_planes, vs, gs, qt, kt, gt, lse, dd, dq, dk, dv, heads=heads, scale=scale)
        tape.add(q, dq)
        tape.add(k, dk)
        tape.add(v, dv)

    tape.record(bwd)
