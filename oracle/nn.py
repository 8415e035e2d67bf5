"""CPU restatement of the TensorFlow-1.x ops the reference's P3D path is built from.

TEST INFRASTRUCTURE ONLY.  Nothing under sap3d_tensorflow_amd/ may import this
package; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
use it, and only as the checker.

PARITY UNPINNED: the reference (Python 2 + TensorFlow 1.x) cannot run in this
image and ships no golden vectors.  The arithmetic lives
in a third-party module that is absent from /root/reference: TensorFlow 1.x,
version unpinned (API use brackets it to 1.5..1.15).  The op semantics below are
restated from TF's documented definitions and anchored on
the reference's own call sites, cited per function.  tests/ cross-check every op
here against an independent torch-CPU composition (forward and autograd).

Everything is plain numpy on NDHWC arrays: a convolution is the sum over kernel
taps of a shifted [M,Cin] x [Cin,Cout] product (Appendix A.2), nothing else.
Storage dtype follows the input (float32 by default, float64 for a stricter
reference).  A tiny tape records one backward closure per op so the network
files can be written forward-only, like the reference's graph-building code.
"""
import math

import numpy as np


# ----------------------------------------------------------------------------
# tape
# ----------------------------------------------------------------------------
class Var:
    """A value on the tape.  `grad` is filled by Tape.backward().  `tag` names the normalisation layer a value came out of
    (its TF scope): relu() looks a pinned decision mask up by it (Tape.pins)."""
    __slots__ = ("data", "grad", "name", "tag")

    def __init__(self, data, name=None):
        self.data = data
        self.grad = None
        self.name = name
        self.tag = None

    @property
    def shape(self):
        return self.data.shape

    def acc(self, g):
        # never in place: several Vars may share one incoming gradient array
        self.grad = g if self.grad is None else self.grad + g


class Tape:
    """pins: decisions taken from ANOTHER evaluation of the same graph (tests/: the HIP forward's), so that the gradient of
    this evaluation is the gradient of the same piecewise-linear branch.  {'relu': {norm scope: bool array}, 'pool':
    [input arrays]} -- a relu whose argument carries a pinned tag takes that mask (forward and backward) instead of
    x > 0; a max-pool picks its arg-max on the pinned input that matches its own (same shape, closest).  pin_log counts
    what was used and how many decisions differed from this evaluation's own."""
    def __init__(self):
        self.ops = []          # backward closures, forward order
        self.updates = []      # (array_to_update, new_value): BN moving stats (UPDATE_OPS)
        self.taps = {}         # name -> Var, activation taps for parity tests
        self.pins = None
        self.pin_log = {"relu": 0, "relu_flips": 0, "pool": 0, "pool_flips": 0, "elements": 0}

    def record(self, fn):
        self.ops.append(fn)

    def tap(self, name, v):
        self.taps[name] = v
        return v

    def backward(self, loss):
        loss.grad = np.ones_like(loss.data)
        for fn in reversed(self.ops):
            fn()

    def apply_updates(self):
        """train.py:170-172: the UPDATE_OPS run with the train op."""
        for arr, new in self.updates:
            arr[...] = new


# ----------------------------------------------------------------------------
# SAME padding arithmetic (Appendix A.1)
# ----------------------------------------------------------------------------
def same_pads(size, k, s):
    out = -(-size // s)
    pad_total = max((out - 1) * s + k - size, 0)
    return out, pad_total // 2, pad_total - pad_total // 2


def _tap_ranges(I, O, k, s, pb):
    """For kernel tap a in [0,k): output range [o_lo, o_hi] whose input index
    i = o*s + a - pb lies inside [0, I).  Returns list of (a, o_lo, o_hi, i_lo)."""
    r = []
    for a in range(k):
        o_lo = max(0, -((a - pb) // s))            # ceil((pb-a)/s)
        o_hi = min(O - 1, (I - 1 + pb - a) // s)
        if o_hi >= o_lo:
            r.append((a, o_lo, o_hi, o_lo * s + a - pb))
    return r


def _conv_taps(xshape, kshape, strides):
    _, D, H, W, _ = xshape
    kd, kh, kw = kshape[:3]
    sd, sh, sw = strides
    Do, pd, _ = same_pads(D, kd, sd)
    Ho, ph, _ = same_pads(H, kh, sh)
    Wo, pw, _ = same_pads(W, kw, sw)
    taps = []
    for (a, d0, d1, id0) in _tap_ranges(D, Do, kd, sd, pd):
        for (b, h0, h1, ih0) in _tap_ranges(H, Ho, kh, sh, ph):
            for (c, w0, w1, iw0) in _tap_ranges(W, Wo, kw, sw, pw):
                osl = (slice(None), slice(d0, d1 + 1), slice(h0, h1 + 1), slice(w0, w1 + 1))
                isl = (slice(None),
                       slice(id0, id0 + (d1 - d0) * sd + 1, sd),
                       slice(ih0, ih0 + (h1 - h0) * sh + 1, sh),
                       slice(iw0, iw0 + (w1 - w0) * sw + 1, sw))
                taps.append(((a, b, c), osl, isl))
    return (Do, Ho, Wo), taps


def conv3d_forward(x, w, strides):
    """tf.nn.conv3d(x, w, [1,sd,sh,sw,1], 'SAME') -- Appendix A.2; call sites
    p3d.py:19,24,86,112,125,172."""
    N = x.shape[0]
    Ci, Co = w.shape[3], w.shape[4]
    (Do, Ho, Wo), taps = _conv_taps(x.shape, w.shape, strides)
    y = np.zeros((N, Do, Ho, Wo, Co), dtype=x.dtype)
    for (a, b, c), osl, isl in taps:
        xs = x[isl]
        y[osl] += (xs.reshape(-1, Ci) @ w[a, b, c]).reshape(xs.shape[:4] + (Co,))
    return y


def conv3d_backward_input(dy, w, strides, xshape):
    """Gradient of conv3d_forward w.r.t. x (TF Conv3DBackpropInputV2)."""
    Ci, Co = w.shape[3], w.shape[4]
    _, taps = _conv_taps(xshape, w.shape, strides)
    dx = np.zeros(xshape, dtype=dy.dtype)
    for (a, b, c), osl, isl in taps:
        g = dy[osl]
        dx[isl] += (g.reshape(-1, Co) @ w[a, b, c].T).reshape(g.shape[:4] + (Ci,))
    return dx


def conv3d_backward_filter(x, dy, wshape, strides):
    """Gradient of conv3d_forward w.r.t. w (TF Conv3DBackpropFilterV2)."""
    Ci, Co = wshape[3], wshape[4]
    _, taps = _conv_taps(x.shape, wshape, strides)
    dw = np.zeros(wshape, dtype=dy.dtype)
    for (a, b, c), osl, isl in taps:
        dw[a, b, c] = x[isl].reshape(-1, Ci).T @ dy[osl].reshape(-1, Co)
    return dw


# ----------------------------------------------------------------------------
# ops on the tape
# ----------------------------------------------------------------------------
def conv3d(tape, x, w, strides=(1, 1, 1), bias=None):
    """tf.nn.conv3d (+ tf.nn.bias_add) / tf.layers.conv3d, SAME padding."""
    y = conv3d_forward(x.data, w.data, strides)
    if bias is not None:
        y += bias.data
    out = Var(y)

    def bwd():
        g = out.grad
        if g is None:
            return
        x.acc(conv3d_backward_input(g, w.data, strides, x.data.shape))
        w.acc(conv3d_backward_filter(x.data, g, w.data.shape, strides))
        if bias is not None:
            bias.acc(g.reshape(-1, g.shape[-1]).sum(0))
    tape.record(bwd)
    return out


def conv3d_transpose(tape, x, kernel, strides, bias=None):
    """tf.layers.conv3d_transpose(x, filters, k, s, 'same') -- Appendix A.3;
    call sites p3d.py:200,205,210,217.  kernel is [kd,kh,kw,Cout,Cin]; the op is
    the input-gradient of a SAME forward conv of stride s whose input has
    extent in*s."""
    N, D, H, W, Cin = x.data.shape
    Cout = kernel.data.shape[3]
    oshape = (N, D * strides[0], H * strides[1], W * strides[2], Cout)
    y = conv3d_backward_input(x.data, kernel.data, strides, oshape)
    if bias is not None:
        y += bias.data
    out = Var(y)

    def bwd():
        g = out.grad
        if g is None:
            return
        x.acc(conv3d_forward(g, kernel.data, strides))
        kernel.acc(conv3d_backward_filter(g, x.data, kernel.data.shape, strides))
        if bias is not None:
            bias.acc(g.reshape(-1, Cout).sum(0))
    tape.record(bwd)
    return out


BN_MOMENTUM = 0.99   # tf.layers.batch_normalization default
BN_EPS = 1e-3        # tf.layers.batch_normalization default


def batch_normalization(tape, x, gamma, beta, moving_mean, moving_var, training):
    """tf.layers.batch_normalization on a rank-5 tensor (non-fused path) --
    Appendix A.4; call sites p3d.py:58,61,67,70,76,79,88,114,127,173,201,206,211.
    moving_mean / moving_var are plain arrays (not trained)."""
    C = x.data.shape[-1]
    xf = x.data.reshape(-1, C)
    M = xf.shape[0]
    if training:
        mean = xf.mean(0, dtype=np.float64)
        var = ((xf.astype(np.float64) - mean) ** 2).mean(0)      # biased, two-pass
        tape.updates.append((moving_mean,
                             (moving_mean * BN_MOMENTUM + mean * (1 - BN_MOMENTUM)).astype(moving_mean.dtype)))
        tape.updates.append((moving_var,
                             (moving_var * BN_MOMENTUM + var * (1 - BN_MOMENTUM)).astype(moving_var.dtype)))
    else:
        mean = moving_mean.astype(np.float64)
        var = moving_var.astype(np.float64)
    inv = 1.0 / np.sqrt(var + BN_EPS)
    dt = x.data.dtype
    xhat = ((xf - mean.astype(dt)) * inv.astype(dt))
    out = Var((xhat * gamma.data + beta.data).reshape(x.data.shape))

    def bwd():
        g = out.grad
        if g is None:
            return
        gf = g.reshape(-1, C)
        dbeta = gf.sum(0, dtype=np.float64)
        dgamma = (gf * xhat).sum(0, dtype=np.float64)
        gamma.acc(dgamma.astype(dt))
        beta.acc(dbeta.astype(dt))
        gi = (gamma.data.astype(np.float64) * inv)
        if training:
            dx = (gf - (dbeta / M).astype(dt) - xhat * (dgamma / M).astype(dt)) * gi.astype(dt)
        else:
            dx = gf * gi.astype(dt)
        x.acc(dx.reshape(x.data.shape))
    tape.record(bwd)
    return out


def group_norm(tape, x, gamma, beta, G=32, eps=1e-5):
    """GroupNorm -- gn/p3d_gn.py:24-46 (== utils/network.py:65-87), Appendix A.5."""
    N, D, H, W, C = x.data.shape
    G = min(G, C)
    cg = C // G
    dt = x.data.dtype
    xg = x.data.reshape(N, D * H * W, G, cg)
    mean = xg.mean(axis=(1, 3), dtype=np.float64, keepdims=True)
    var = ((xg.astype(np.float64) - mean) ** 2).mean(axis=(1, 3), keepdims=True)
    inv = 1.0 / np.sqrt(var + eps)
    xhat = ((xg - mean.astype(dt)) * inv.astype(dt)).reshape(N, D, H, W, C)
    out = Var(xhat * gamma.data + beta.data)
    cnt = D * H * W * cg

    def bwd():
        g = out.grad
        if g is None:
            return
        gamma.acc((g * xhat).reshape(-1, C).sum(0, dtype=np.float64).astype(dt))
        beta.acc(g.reshape(-1, C).sum(0, dtype=np.float64).astype(dt))
        gh = (g * gamma.data).reshape(N, D * H * W, G, cg)
        xh = xhat.reshape(N, D * H * W, G, cg)
        s1 = gh.sum(axis=(1, 3), dtype=np.float64, keepdims=True) / cnt
        s2 = (gh * xh).sum(axis=(1, 3), dtype=np.float64, keepdims=True) / cnt
        dx = (gh - s1.astype(dt) - xh * s2.astype(dt)) * inv.astype(dt)
        x.acc(dx.reshape(x.data.shape))
    tape.record(bwd)
    return out


def relu(tape, x):
    """tf.nn.relu; gradient is dy where y > 0 (zero at 0).  With a pinned mask for x's tag (Tape.pins) the mask decides."""
    pin = None
    if tape.pins is not None and x.tag is not None:
        pin = tape.pins.get("relu", {}).get(x.tag)
    if pin is not None:
        mask = np.asarray(pin).reshape(x.data.shape).astype(bool)
        own = x.data > 0
        tape.pin_log["relu"] += 1
        tape.pin_log["relu_flips"] += int((own != mask).sum())
        tape.pin_log["elements"] += mask.size
        out = Var(np.where(mask, x.data, 0).astype(x.data.dtype))
    else:
        mask = None
        out = Var(np.maximum(x.data, 0))

    def bwd():
        if out.grad is not None:
            x.acc(out.grad * (out.data > 0 if mask is None else mask))
    tape.record(bwd)
    return out


def add(tape, a, b):
    out = Var(a.data + b.data)
    out.tag = a.tag if a.tag is not None else b.tag      # relu(bn(y) + r): the sum is gated under the BatchNorm's name

    def bwd():
        if out.grad is not None:
            a.acc(out.grad)
            b.acc(out.grad)
    tape.record(bwd)
    return out


def mul(tape, a, b):
    """Broadcasting product (CBAM scale, utils/network.py:249,274)."""
    out = Var(a.data * b.data)

    def _unb(g, shape):
        lead = g.ndim - len(shape)              # numpy broadcasting aligns trailing axes
        if lead:
            g = g.sum(axis=tuple(range(lead)))
        axes = tuple(i for i, (gs, s) in enumerate(zip(g.shape, shape)) if s == 1 and gs != 1)
        return g.sum(axis=axes, keepdims=True) if axes else g

    def bwd():
        if out.grad is not None:
            a.acc(_unb(out.grad * b.data, a.data.shape))
            b.acc(_unb(out.grad * a.data, b.data.shape))
    tape.record(bwd)
    return out


def sigmoid(tape, x):
    """tf.sigmoid (p3d.py:219)."""
    out = Var(1.0 / (1.0 + np.exp(-x.data)))

    def bwd():
        if out.grad is not None:
            x.acc(out.grad * out.data * (1 - out.data))
    tape.record(bwd)
    return out


def concat(tape, xs):
    """tf.concat(axis=-1) (p3d.py:203,208)."""
    out = Var(np.concatenate([v.data for v in xs], axis=-1))

    def bwd():
        if out.grad is None:
            return
        o = 0
        for v in xs:
            c = v.data.shape[-1]
            v.acc(np.ascontiguousarray(out.grad[..., o:o + c]))
            o += c
    tape.record(bwd)
    return out


def max_pool3d(tape, x, ksize, strides):
    """tf.nn.max_pool3d(x, [1,kd,kh,kw,1], [1,sd,sh,sw,1], 'SAME') -- Appendix
    A.1/A.6; call sites p3d.py:176,177,183,189,195.  Padded cells are ignored.
    Backward routes dy to the first maximum in (kd,kh,kw) scan order; ties only
    occur between post-ReLU zeros, whose ReLU gradient is zero anyway."""
    xd = x.data
    N, D, H, W, C = xd.shape
    (Do, Ho, Wo), taps = _conv_taps(xd.shape, tuple(ksize) + (C, C), strides)

    def arg_max(src):
        best = np.full((N, Do, Ho, Wo, C), -np.inf, dtype=src.dtype)
        which = np.full((N, Do, Ho, Wo, C), -1, dtype=np.int16)
        for t, (_, osl, isl) in enumerate(taps):
            v = src[isl]
            better = v > best[osl]
            best[osl] = np.where(better, v, best[osl])
            which[osl] = np.where(better, t, which[osl])
        return best, which

    y, idx = arg_max(xd)
    if tape.pins is not None and tape.pins.get("pool"):
        # the pinned evaluation's input of THIS pool: same shape, closest values (it is the same tensor up to rounding)
        cands = [p for p in tape.pins["pool"] if tuple(p.shape) == tuple(xd.shape)]
        scale = max(float(np.abs(xd).max()), 1e-30)
        dist = [float(np.abs(p.astype(np.float64) - xd).max()) / scale for p in cands]
        if not cands or min(dist) > tape.pins.get("pool_tol", 1e-3):
            raise ValueError("no pinned max-pool input matches this pool's input %s (distances %s)" % (xd.shape, dist))
        _, pinned = arg_max(cands[int(np.argmin(dist))])
        tape.pin_log["pool"] += 1
        tape.pin_log["pool_flips"] += int((pinned != idx).sum())
        idx = pinned
        y = np.zeros((N, Do, Ho, Wo, C), dtype=xd.dtype)
        for t, (_, osl, isl) in enumerate(taps):
            y[osl] = np.where(idx[osl] == t, xd[isl], y[osl])
    out = Var(y)

    def bwd():
        if out.grad is None:
            return
        dx = np.zeros_like(xd)
        for t, (_, osl, isl) in enumerate(taps):
            dx[isl] += out.grad[osl] * (idx[osl] == t)
        x.acc(dx)
    tape.record(bwd)
    return out


def dropout(tape, x, rate, training, keep_mask=None):
    """tf.layers.dropout (p3d.py:214): inverted dropout; identity when not
    training or rate == 0.  TF's RNG stream cannot be matched, so parity runs use
    rate 0 or pass the keep mask explicitly."""
    if not training or rate == 0:
        return x
    if keep_mask is None:
        raise ValueError("dropout with rate>0 needs an explicit keep_mask in the oracle")
    scale = 1.0 / (1.0 - rate)
    out = Var(x.data * keep_mask * x.data.dtype.type(scale))

    def bwd():
        if out.grad is not None:
            x.acc(out.grad * keep_mask * x.data.dtype.type(scale))
    tape.record(bwd)
    return out


def smooth_l1_loss(tape, pred, target, inside_w=1.0, outside_w=1.0, sigma=1.0):
    """utils/network.py:49-62, called at train.py:159 with weights 1,1 sigma=1.
    NOTE reduce_mean wraps a scalar reduce_sum: the loss is a SUM."""
    s2 = sigma ** 2
    d = inside_w * (pred.data - target)
    ad = np.abs(d)
    sign = (ad < 1.0 / s2).astype(pred.data.dtype)      # stop_gradient
    per = (d * d) * (s2 / 2.0) * sign + (ad - 0.5 / s2) * (1 - sign)
    out = Var(np.asarray((outside_w * per).sum(dtype=np.float64), dtype=pred.data.dtype))

    def bwd():
        g = out.grad
        dd = d * s2 * sign + np.sign(d) * (1 - sign)
        pred.acc((g * outside_w * inside_w * dd).astype(pred.data.dtype))
    tape.record(bwd)
    return out


def reshape(tape, x, shape):
    out = Var(x.data.reshape(shape))

    def bwd():
        if out.grad is not None:
            x.acc(out.grad.reshape(x.data.shape))
    tape.record(bwd)
    return out


def reduce_mean(tape, x, axes):
    cnt = int(np.prod([x.data.shape[a] for a in axes]))
    out = Var(x.data.mean(axis=tuple(axes), keepdims=True, dtype=np.float64).astype(x.data.dtype))

    def bwd():
        if out.grad is not None:
            x.acc(np.broadcast_to(out.grad / cnt, x.data.shape).astype(x.data.dtype))
    tape.record(bwd)
    return out


def reduce_max(tape, x, axes):
    """tf.reduce_max; gradient split equally between tied maxima (TF's
    _MinOrMaxGrad divides by the number of ties)."""
    m = x.data.max(axis=tuple(axes), keepdims=True)
    out = Var(m)

    def bwd():
        if out.grad is not None:
            ind = (x.data == m).astype(x.data.dtype)
            x.acc(ind / ind.sum(axis=tuple(axes), keepdims=True) * out.grad)
    tape.record(bwd)
    return out


def matmul(tape, a, b, transpose_b=False):
    """tf.matmul on [batch, m, k] x [batch, k, n] (or [batch, n, k] with transpose_b); utils/network.py:183,185."""
    bd = np.swapaxes(b.data, -1, -2) if transpose_b else b.data
    out = Var(np.matmul(a.data, bd))

    def bwd():
        if out.grad is None:
            return
        a.acc(np.matmul(out.grad, np.swapaxes(bd, -1, -2)))
        gb = np.matmul(np.swapaxes(a.data, -1, -2), out.grad)
        b.acc(np.swapaxes(gb, -1, -2) if transpose_b else gb)
    tape.record(bwd)
    return out


def softmax(tape, x):
    """tf.nn.softmax(x, axis=-1) (utils/network.py:184)."""
    e = np.exp(x.data - x.data.max(axis=-1, keepdims=True))
    y = e / e.sum(axis=-1, keepdims=True)
    out = Var(y)

    def bwd():
        if out.grad is not None:
            x.acc(y * (out.grad - (out.grad * y).sum(axis=-1, keepdims=True)))
    tape.record(bwd)
    return out


def dense(tape, x, kernel, bias):
    """tf.layers.dense on the last axis (utils/network.py:218-245)."""
    out = Var(x.data @ kernel.data + bias.data)

    def bwd():
        g = out.grad
        if g is None:
            return
        x.acc(g @ kernel.data.T)
        kernel.acc(x.data.reshape(-1, x.data.shape[-1]).T @ g.reshape(-1, g.shape[-1]))
        bias.acc(g.reshape(-1, g.shape[-1]).sum(0))
    tape.record(bwd)
    return out


# ----------------------------------------------------------------------------
# optimiser and initialisers
# ----------------------------------------------------------------------------
def adam_step(p, g, m, v, t, lr=1e-4, b1=0.9, b2=0.999, eps=1e-8):
    """tf.train.AdamOptimizer (train.py:168), Appendix A.6: epsilon-hat form.
    Updates p, m, v in place; t is the 1-based step count."""
    m *= b1
    m += (1 - b1) * g
    v *= b2
    v += (1 - b2) * g * g
    lr_t = lr * math.sqrt(1 - b2 ** t) / (1 - b1 ** t)
    p -= (lr_t * m / (np.sqrt(v) + eps)).astype(p.dtype)


def xavier_uniform(rng, shape, dtype=np.float32):
    """tf.contrib.layers.xavier_initializer() / glorot_uniform -- Appendix A.7.
    fan_in / fan_out follow TF's _compute_fans: receptive field x shape[-2] /
    shape[-1]; a rank-1 shape [C] has fan_in = fan_out = C (p3d.py:22,27)."""
    if len(shape) == 1:
        fi = fo = shape[0]
    else:
        rf = int(np.prod(shape[:-2]))
        fi, fo = rf * shape[-2], rf * shape[-1]
    lim = math.sqrt(6.0 / (fi + fo))
    return rng.uniform(-lim, lim, size=shape).astype(dtype)


def variance_scaling_normal(rng, shape, dtype=np.float32):
    """tf.contrib.layers.variance_scaling_initializer() default (factor 2,
    FAN_IN, truncated normal with stddev sqrt(1.3*2/fan_in)) -- Appendix A.7."""
    rf = int(np.prod(shape[:-2])) if len(shape) > 2 else 1
    fan_in = rf * shape[-2]
    std = math.sqrt(1.3 * 2.0 / fan_in)
    x = rng.standard_normal(size=shape)
    bad = np.abs(x) > 2
    while bad.any():
        x[bad] = rng.standard_normal(size=int(bad.sum()))
        bad = np.abs(x) > 2
    return (x * std).astype(dtype)
