"""``torch.library`` registration of the hot-path ops whose interface is pure tensors (BASELINE.json north_star: "Python host
on PyTorch-ROCm registering custom ops through a thin C-ABI").  Each ``tamgcn::*`` op is a ``torch.library.custom_op`` whose
implementation launches the C ABI (through ``tam_gcn_amd.ops``), with a fake (meta) implementation so that export /
``torch.compile`` trace through it without a graph break, and -- for the differentiable ones -- ``register_autograd`` whose
backward is itself a registered op:

    tamgcn::ctrgc(x, A, alpha, w1, b1, w2, b2, w3, b3, w4, b4) -> y          reference models/ctrgcn.py:172-177
    tamgcn::ctrgc_backward(dy, x, A, alpha, w1, ..., b4) -> 11 gradients
    tamgcn::cross_entropy(logits, labels) -> (loss, g);  tamgcn::cross_entropy_backward(g, dloss) -> dlogits
    tamgcn::pointwise_conv(x, w, b) -> y;  tamgcn::pointwise_conv_backward(dy, x, w) -> (dx, dw, db)
    tamgcn::head(x, W, b, M) -> logits;  tamgcn::head_backward(dlogits, x, W, M) -> (dx, dW, db)        models/ctrgcn.py:343-348
    tamgcn::head_pooled(x, rowmean, W, b, M): the same with the row means of x already taken by the producer of x
    tamgcn::stream_derive(x, parent, mode) -> stream            feeder/feeder_nucla_gcn.py:119-127
    tamgcn::feeder_transform(raw, offsets, rot, idx, parent, V, time_steps, center_joint, mode) -> clips   :85-130

    tamgcn::tcn_gcn_unit_eval(x, xpart, params, geom) -> (out, xpart)   the whole eval-mode TCN_GCN_unit (:266-284) for small
                                                 batches, BatchNorm folded; registered by tam_gcn_amd/f2.py with the engine that uses it

The TRAINING block-level nodes (unit_gcn / MultiScale_TemporalConv / TCN_GCN_unit / st_gcn) stay ``autograd.Function``s: they update
BatchNorm running statistics in place, keep ~20 intermediate tensors between forward and backward and take their
configuration from the nn.Module; the modules ``CTRGC``, ``CrossEntropyLoss``, the ST-GCN per-position classifier and the
model heads call the registered ops."""
import torch
from torch import Tensor

from . import ops
from .ops import S


def _c(t):
    return t.contiguous()


# ---------------------------------------------------------------------------------------------------------------
# CTRGC (single subset; A and alpha are inputs)
# ---------------------------------------------------------------------------------------------------------------
def _ctrgc_pack(x, A, alpha, w1, b1, w2, b2, w3, b3, w4, b4):
    N, Cin, T, V = x.shape
    Cout, R = w3.shape[0], w1.shape[0]
    W12 = torch.cat((w1.reshape(R, Cin), w2.reshape(R, Cin)))
    B12 = torch.cat((b1, b2))
    W3, W4 = w3.reshape(Cout, Cin).contiguous(), w4.reshape(1, Cout, R).contiguous()
    A3 = A.reshape(1, V, V).contiguous()
    al = alpha.reshape(1).to(torch.float32).contiguous()
    xs = S(x)
    xbar = ops.tmean(xs, Cin)
    pq, _ = ops.conv(S(xbar.view(1, Cin, N, V)), K=Cin, w=W12, bias=B12, M=2 * R)
    return dict(N=N, Cin=Cin, T=T, V=V, Cout=Cout, R=R, W12=W12, W3=W3, W4=W4, A3=A3, al=al, xs=xs, xbar=xbar,
                pq=pq.view(2 * R, N, V), b3=_c(b3), b4=b4.reshape(1, Cout).contiguous())


@torch.library.custom_op('tamgcn::ctrgc', mutates_args=())
def ctrgc(x: Tensor, A: Tensor, alpha: Tensor, w1: Tensor, b1: Tensor, w2: Tensor, b2: Tensor, w3: Tensor, b3: Tensor,
          w4: Tensor, b4: Tensor) -> Tensor:
    p = _ctrgc_pack(_c(x), A, alpha, w1, b1, w2, b2, w3, b3, w4, b4)
    y, _, _ = ops.ctrgc_fwd(p['xs'], p['pq'], p['W3'], p['b3'], p['W4'], p['b4'], p['A3'], p['al'], p['Cin'], p['Cout'], 1, p['R'],
                            stats=False)
    return y


@ctrgc.register_fake
def _(x, A, alpha, w1, b1, w2, b2, w3, b3, w4, b4):
    return x.new_empty(x.shape[0], w3.shape[0], x.shape[2], x.shape[3])


@torch.library.custom_op('tamgcn::ctrgc_backward', mutates_args=())
def ctrgc_backward(dy: Tensor, x: Tensor, A: Tensor, alpha: Tensor, w1: Tensor, b1: Tensor, w2: Tensor, b2: Tensor, w3: Tensor,
                   b3: Tensor, w4: Tensor, b4: Tensor) -> list[Tensor]:
    x = _c(x)
    p = _ctrgc_pack(x, A, alpha, w1, b1, w2, b2, w3, b3, w4, b4)       # x-bar and p, q are recomputed: a mean and a tiny GEMM
    N, Cin, T, V, Cout, R = p['N'], p['Cin'], p['T'], p['V'], p['Cout'], p['R']
    xs = p['xs']
    dx3, db3, dA, dW4, db4, dal, dpq = ops.ctrgc_bwd(xs, p['pq'], p['W3'], p['b3'], p['W4'], p['b4'], p['A3'], p['al'], Cin, Cout, 1, R,
                                                     S(_c(dy)))
    dpq4 = S(dpq.view(1, 2 * R, N, V))
    dW12 = ops.wgrad(dpq4, S(p['xbar'].view(1, Cin, N, V)), M=2 * R, K=Cin)
    dB12 = dpq.sum((1, 2))
    dW3 = ops.wgrad(S(dx3), xs, M=Cout, K=Cin)
    dxbar, _ = ops.conv(dpq4, K=2 * R, w=p['W12'], bias=None, M=Cin, wmode=1)
    dx, _ = ops.conv(S(dx3), K=Cout, w=p['W3'], bias=None, M=Cin, wmode=1, bcast=dxbar.view(Cin, N, V), bcast_scale=1.0 / T)
    return [dx, dA.reshape(A.shape), dal.reshape(alpha.shape).to(alpha.dtype), dW12[:R].reshape(w1.shape).clone(), dB12[:R].clone(),
            dW12[R:].reshape(w2.shape).clone(), dB12[R:].clone(), dW3.reshape(w3.shape), db3, dW4.reshape(w4.shape), db4.reshape(b4.shape)]


@ctrgc_backward.register_fake
def _(dy, x, A, alpha, w1, b1, w2, b2, w3, b3, w4, b4):
    return [torch.empty_like(t) for t in (x, A, alpha, w1, b1, w2, b2, w3, b3, w4, b4)]


def _ctrgc_setup(ctx, inputs, output):
    ctx.save_for_backward(*inputs)


def _ctrgc_bwd(ctx, dy):
    return tuple(torch.ops.tamgcn.ctrgc_backward(dy, *ctx.saved_tensors))


ctrgc.register_autograd(_ctrgc_bwd, setup_context=_ctrgc_setup)


# ---------------------------------------------------------------------------------------------------------------
# cross-entropy (mean reduction)
# ---------------------------------------------------------------------------------------------------------------
@torch.library.custom_op('tamgcn::cross_entropy', mutates_args=())
def cross_entropy(logits: Tensor, labels: Tensor) -> tuple[Tensor, Tensor]:
    return ops.ce_fwd(_c(logits), _c(labels))


@cross_entropy.register_fake
def _(logits, labels):
    return logits.new_empty(()), torch.empty_like(logits)


@torch.library.custom_op('tamgcn::cross_entropy_backward', mutates_args=())
def cross_entropy_backward(g: Tensor, dloss: Tensor) -> Tensor:
    return ops.ce_bwd(_c(g), dloss.to(torch.float32).contiguous())


@cross_entropy_backward.register_fake
def _(g, dloss):
    return torch.empty_like(g)


def _ce_setup(ctx, inputs, output):
    ctx.save_for_backward(output[1])


def _ce_bwd(ctx, dloss, dg):
    (g,) = ctx.saved_tensors
    return torch.ops.tamgcn.cross_entropy_backward(g, dloss), None


cross_entropy.register_autograd(_ce_bwd, setup_context=_ce_setup)


# ---------------------------------------------------------------------------------------------------------------
# 1x1 convolution with bias over (N, C, T, V)
# ---------------------------------------------------------------------------------------------------------------
@torch.library.custom_op('tamgcn::pointwise_conv', mutates_args=())
def pointwise_conv(x: Tensor, w: Tensor, b: Tensor) -> Tensor:
    x = _c(x)
    y, _ = ops.conv(S(x), K=x.shape[1], w=w.reshape(w.shape[0], -1).contiguous(), bias=_c(b), M=w.shape[0])
    return y


@pointwise_conv.register_fake
def _(x, w, b):
    return x.new_empty(x.shape[0], w.shape[0], x.shape[2], x.shape[3])


@torch.library.custom_op('tamgcn::pointwise_conv_backward', mutates_args=())
def pointwise_conv_backward(dy: Tensor, x: Tensor, w: Tensor) -> tuple[Tensor, Tensor, Tensor]:
    dy, x = _c(dy), _c(x)
    M, K = w.shape[0], x.shape[1]
    dw = ops.wgrad(S(dy), S(x), M=M, K=K).reshape(w.shape)
    db = dy.sum((0, 2, 3))
    dx, _ = ops.conv(S(dy), K=M, w=w.reshape(M, K).contiguous(), bias=None, M=K, wmode=1)
    return dx, dw, db


@pointwise_conv_backward.register_fake
def _(dy, x, w):
    return torch.empty_like(x), torch.empty_like(w), x.new_empty(w.shape[0])


def _pc_setup(ctx, inputs, output):
    ctx.save_for_backward(inputs[0], inputs[1])


def _pc_bwd(ctx, dy):
    x, w = ctx.saved_tensors
    return torch.ops.tamgcn.pointwise_conv_backward(dy, x, w)


pointwise_conv.register_autograd(_pc_bwd, setup_context=_pc_setup)


# ---------------------------------------------------------------------------------------------------------------
# head: mean over (m, t, v) then the classifier
# ---------------------------------------------------------------------------------------------------------------
@torch.library.custom_op('tamgcn::head', mutates_args=())
def head(x: Tensor, W: Tensor, b: Tensor, M: int) -> Tensor:
    pooled = ops.head_pool_fwd(_c(x), M)
    return ops.head_fc_fwd(pooled, _c(W), _c(b))


@head.register_fake
def _(x, W, b, M):
    return x.new_empty(x.shape[0] // M, W.shape[0])


@torch.library.custom_op('tamgcn::head_backward', mutates_args=())
def head_backward(dlogits: Tensor, x: Tensor, W: Tensor, M: int) -> tuple[Tensor, Tensor, Tensor]:
    x = _c(x)
    pooled = ops.head_pool_fwd(x, M)                                     # recomputed: one pass over the last block's output
    dW, db, dpooled = ops.head_fc_bwd(_c(dlogits), pooled, _c(W))
    return ops.head_pool_bwd(dpooled, M, x.shape[2], x.shape[3]), dW, db


@head_backward.register_fake
def _(dlogits, x, W, M):
    return torch.empty_like(x), torch.empty_like(W), x.new_empty(W.shape[0])


def _head_setup(ctx, inputs, output):
    ctx.save_for_backward(inputs[0], inputs[1])
    ctx.M = inputs[3]


def _head_bwd(ctx, dl):
    x, W = ctx.saved_tensors
    dx, dW, db = torch.ops.tamgcn.head_backward(dl, x, W, ctx.M)
    return dx, dW, db, None


head.register_autograd(_head_bwd, setup_context=_head_setup)


# the same head when the last block already produced the (N*M, C) row means of x in its final pass: x is an argument only
# for its gradient (the broadcast of d pooled), it is not read
def _pooled_of(rowmean, M):
    rowmean = _c(rowmean)
    if M == 1:                                           # one body per clip: the row means are the pooled features
        return rowmean
    NM, Cc = rowmean.shape
    return ops.head_pool_fwd(rowmean.view(NM, Cc, 1, 1), M)


@torch.library.custom_op('tamgcn::head_pooled', mutates_args=())
def head_pooled(x: Tensor, rowmean: Tensor, W: Tensor, b: Tensor, M: int) -> Tensor:
    return ops.head_fc_fwd(_pooled_of(rowmean, M), _c(W), _c(b))


@head_pooled.register_fake
def _(x, rowmean, W, b, M):
    return x.new_empty(x.shape[0] // M, W.shape[0])


@torch.library.custom_op('tamgcn::head_pooled_backward', mutates_args=())
def head_pooled_backward(dlogits: Tensor, rowmean: Tensor, W: Tensor, M: int, T: int, V: int) -> tuple[Tensor, Tensor, Tensor]:
    dW, db, dpooled = ops.head_fc_bwd(_c(dlogits), _pooled_of(rowmean, M), _c(W))
    return ops.head_pool_bwd(dpooled, M, T, V), dW, db


@head_pooled_backward.register_fake
def _(dlogits, rowmean, W, M, T, V):
    return rowmean.new_empty(rowmean.shape[0], rowmean.shape[1], T, V), torch.empty_like(W), rowmean.new_empty(W.shape[0])


def _headp_setup(ctx, inputs, output):
    ctx.save_for_backward(inputs[1], inputs[2])
    ctx.M, ctx.T, ctx.V = inputs[4], inputs[0].shape[2], inputs[0].shape[3]


def _headp_bwd(ctx, dl):
    rm, W = ctx.saved_tensors
    dx, dW, db = torch.ops.tamgcn.head_pooled_backward(dl, rm, W, ctx.M, ctx.T, ctx.V)
    return dx, None, dW, db, None


head_pooled.register_autograd(_headp_bwd, setup_context=_headp_setup)


# ---------------------------------------------------------------------------------------------------------------
# input side (no autograd)
# ---------------------------------------------------------------------------------------------------------------
@torch.library.custom_op('tamgcn::stream_derive', mutates_args=())
def stream_derive(x: Tensor, parent: Tensor, mode: int) -> Tensor:
    return ops.stream_derive(_c(x), parent, mode) if mode else x.clone()


@stream_derive.register_fake
def _(x, parent, mode):
    return torch.empty_like(x)


@torch.library.custom_op('tamgcn::feeder_transform', mutates_args=())
def feeder_transform(raw: Tensor, offsets: Tensor, rot: Tensor, idx: Tensor, parent: Tensor, V: int, time_steps: int,
                     center_joint: int, mode: int) -> Tensor:
    return ops.feeder_transform(raw, offsets, rot, idx, parent, V, time_steps, center_joint, mode)


@feeder_transform.register_fake
def _(raw, offsets, rot, idx, parent, V, time_steps, center_joint, mode):
    return raw.new_empty(offsets.shape[0] - 1, 3, time_steps, V, 1, dtype=torch.float32)
