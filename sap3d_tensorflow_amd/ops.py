"""The TensorFlow ops the reference's P3D path is made of, as eager numpy-in / numpy-out calls
into the HIP kernels (argument order and meaning follow tf.nn.* / tf.layers.*)."""
import ctypes as C

import numpy as np

from ._lib import check, fptr, lib


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _shape5(s):
    return (C.c_int64 * 5)(*[int(v) for v in s])


def _i3(s):
    return (C.c_int * 3)(*[int(v) for v in s])


def _same_out(size, s):
    return -(-size // s)


def _strides3(strides):
    strides = list(strides)
    if len(strides) == 5:
        if strides[0] != 1 or strides[4] != 1:
            raise ValueError("strides must be [1, sd, sh, sw, 1]")
        strides = strides[1:4]
    return strides


def conv3d(x, filter, strides, padding="SAME", bias=None, device=0):
    """tf.nn.conv3d(x, filter, strides, 'SAME') (+ tf.nn.bias_add)."""
    if padding.upper() != "SAME":
        raise ValueError("only SAME padding is on the P3D path")
    x, w = _f32(x), _f32(filter)
    s = _strides3(strides)
    if w.shape[3] != x.shape[4]:
        raise ValueError("filter in-channels %d != input channels %d" % (w.shape[3], x.shape[4]))
    y = np.empty((x.shape[0],) + tuple(_same_out(x.shape[1 + i], s[i]) for i in range(3)) + (w.shape[4],), np.float32)
    b = _f32(bias) if bias is not None else None
    check(lib().p3d_op_conv3d(device, fptr(x), _shape5(x.shape), fptr(w), _shape5(w.shape), _i3(s), fptr(b), fptr(y)))
    return y


def conv3d_backprop_input(input_sizes, filter, out_backprop, strides, device=0):
    """tf.nn.conv3d_backprop_input_v2."""
    w, dy = _f32(filter), _f32(out_backprop)
    s = _strides3(strides)
    dx = np.empty(tuple(input_sizes), np.float32)
    check(lib().p3d_op_conv3d_backprop_input(device, fptr(dy), fptr(w), _shape5(w.shape), _i3(s), _shape5(input_sizes), fptr(dx)))
    return dx


def conv3d_backprop_filter(input, filter_sizes, out_backprop, strides, with_bias=False, device=0):
    """tf.nn.conv3d_backprop_filter_v2 (optionally also the bias_add gradient)."""
    x, dy = _f32(input), _f32(out_backprop)
    s = _strides3(strides)
    dw = np.empty(tuple(filter_sizes), np.float32)
    db = np.empty((filter_sizes[4],), np.float32) if with_bias else None
    check(lib().p3d_op_conv3d_backprop_filter(device, fptr(x), _shape5(x.shape), fptr(dy), _shape5(filter_sizes), _i3(s),
                                             fptr(dw), fptr(db)))
    return (dw, db) if with_bias else dw


def conv3d_transpose(x, kernel, strides, bias=None, device=0):
    """tf.layers.conv3d_transpose(x, filters, k, strides, 'same'); kernel is [kd,kh,kw,Cout,Cin]."""
    x, k = _f32(x), _f32(kernel)
    s = _strides3(strides)
    y = np.empty((x.shape[0], x.shape[1] * s[0], x.shape[2] * s[1], x.shape[3] * s[2], k.shape[3]), np.float32)
    b = _f32(bias) if bias is not None else None
    check(lib().p3d_op_conv3d_transpose(device, fptr(x), _shape5(x.shape), fptr(k), _shape5(k.shape), _i3(s), fptr(b), fptr(y)))
    return y


def bias_add_grad(dy, device=0):
    """BiasAddGrad (gradient of the bias of tf.layers.conv3d / conv3d_transpose): sum of dy over every axis but the last."""
    g = _f32(dy)
    c = g.shape[-1]
    out = np.empty((c,), np.float32)
    check(lib().p3d_op_bias_add_grad(device, fptr(g), g.size // c if c else 0, c, fptr(out)))
    return out


def attention_core(g, f, h, d_o=None, device=0):
    """utils/network.py:183-185 on flattened operands: o = softmax(g f^T) h for g [B, Ng, ch/8], f [B, Nf, ch/8], h [B, Nf, ch]
    (ch in 32, 64, 128, 256), the score matrix never stored.  With d_o (gradient of o): returns (o, dg, df, dh)."""
    g, f, h = _f32(g), _f32(f), _f32(h)
    B, ng, ci = g.shape
    nf, ch = h.shape[1], h.shape[2]
    if f.shape != (B, nf, ci) or h.shape[0] != B or ci * 8 != ch:
        raise ValueError("attention_core: g [B,Ng,ch/8], f [B,Nf,ch/8], h [B,Nf,ch]")
    o = np.empty((B, ng, ch), np.float32)
    if d_o is None:
        check(lib().p3d_op_attention_core(device, B, ng, nf, ch, fptr(g), fptr(f), fptr(h), fptr(o), None, None, None, None))
        return o
    d = _f32(d_o)
    if d.shape != o.shape:
        raise ValueError("attention_core: d_o has the shape of o")
    dg, df, dh = np.empty_like(g), np.empty_like(f), np.empty_like(h)
    check(lib().p3d_op_attention_core(device, B, ng, nf, ch, fptr(g), fptr(f), fptr(h), fptr(o), fptr(d), fptr(dg), fptr(df), fptr(dh)))
    return o, dg, df, dh


def max_pool3d(x, ksize, strides, padding="SAME", device=0):
    """tf.nn.max_pool3d(x, [1,kd,kh,kw,1], [1,sd,sh,sw,1], 'SAME')."""
    x = _f32(x)
    k, s = _strides3(ksize), _strides3(strides)
    y = np.empty((x.shape[0],) + tuple(_same_out(x.shape[1 + i], s[i]) for i in range(3)) + (x.shape[4],), np.float32)
    check(lib().p3d_op_max_pool3d(device, fptr(x), _shape5(x.shape), _i3(k), _i3(s), fptr(y)))
    return y


def max_pool3d_grad(x, ksize, strides, grad, device=0):
    x, g = _f32(x), _f32(grad)
    k, s = _strides3(ksize), _strides3(strides)
    dx = np.empty(x.shape, np.float32)
    check(lib().p3d_op_max_pool3d_grad(device, fptr(x), _shape5(x.shape), _i3(k), _i3(s), fptr(g), fptr(dx)))
    return dx
