"""CPU restatement of the reference's BatchNorm P3D networks (p3d.py).

TEST INFRASTRUCTURE ONLY (see oracle/nn.py header).  PARITY UNPINNED.

Function and variable names follow /root/reference/p3d.py so the two can be read
side by side; every function cites the lines it restates.  The graph functions
take an extra leading `g` (a Graph: parameter store + tape) because there is no
TF default graph here, and a `cfg` that scales widths/depths for fast tests
(cfg=None is the reference architecture: P3D-199, 64/128/256 planes, 3/8/36
blocks).
"""
from collections import OrderedDict

import numpy as np

from . import nn

BLOCK_EXPANSION = 4      # p3d.py:8


class NetConfig:
    """Architecture knobs.  The reference hard-codes base=64, blocks=(3,8,36)."""

    def __init__(self, base=64, blocks=(3, 8, 36)):
        self.base = base
        self.blocks = tuple(blocks)
        self.depth_3d = sum(blocks)        # p3d.py:31 depth_3d=47: every block is 3-D


REFERENCE_CFG = NetConfig()


class Graph:
    """Parameter store (TF variable names) + tape.

    create=True draws missing parameters from `rng` with the reference's
    initialisers (Appendix A.7); create=False requires them to be present."""

    def __init__(self, params=None, rng=None, dtype=np.float32, create=True):
        self.params = params if params is not None else OrderedDict()
        self.trainable = OrderedDict()     # name -> Var (this graph's trainables, creation order)
        self.rng = rng if rng is not None else np.random.default_rng(1)
        self.dtype = dtype
        self.create = create
        self.tape = nn.Tape()
        self._uniq = {}
        self.prefix = ''                   # enclosing tf.variable_scope, e.g. 'P3D/' (gn/p3d_gn.py:490)

    # -- TF-style unique auto names: batch_normalization, batch_normalization_1, ...
    def unique(self, base):
        k = self._uniq.get(base, 0)
        self._uniq[base] = k + 1
        return base if k == 0 else "%s_%d" % (base, k)

    def _array(self, name, shape, init):
        name = self.prefix + name
        if name not in self.params:
            if not self.create:
                raise KeyError("missing parameter " + name)
            self.params[name] = init(shape).astype(self.dtype)
        a = self.params[name]
        assert tuple(a.shape) == tuple(shape), (name, a.shape, shape)
        return a

    def variable(self, name, shape, init):
        full = self.prefix + name
        if full in self.trainable:
            return self.trainable[full]
        v = nn.Var(self._array(name, shape, init), full)
        self.trainable[full] = v
        return v

    def state(self, name, shape, value):
        return self._array(name, shape, lambda s: np.full(s, value))

    def xavier(self, shape):
        return nn.xavier_uniform(self.rng, shape, self.dtype)

    def zeros(self, shape):
        return np.zeros(shape, self.dtype)

    def ones(self, shape):
        return np.ones(shape, self.dtype)


# ---------------------------------------------------------------------------
# helpers restating p3d.py:10-27
# ---------------------------------------------------------------------------
def get_conv_weight(g, name, kshape, wd=0.001):
    """p3d.py:10-16.  The weight-decay term goes to a collection that is never
    added to the loss (train.py:161-162), so it is not modelled."""
    return g.variable(name, tuple(kshape), g.xavier)


def convS(g, name, l_input, in_channels, out_channels):
    """p3d.py:18-22: 1x3x3 SAME conv + bias (bias is Xavier-initialised too)."""
    return nn.conv3d(g.tape, l_input, get_conv_weight(g, name, [1, 3, 3, in_channels, out_channels]),
                     (1, 1, 1), get_conv_weight(g, name + '_bias', [out_channels], 0))


def convT(g, name, l_input, in_channels, out_channels):
    """p3d.py:23-27: 3x1x1 SAME conv + bias."""
    return nn.conv3d(g.tape, l_input, get_conv_weight(g, name, [3, 1, 1, in_channels, out_channels]),
                     (1, 1, 1), get_conv_weight(g, name + '_bias', [out_channels], 0))


def batch_normalization(g, x, training, name=None):
    """tf.layers.batch_normalization(x, training=..., name=...) with TF's
    variable naming (<scope>/gamma, beta, moving_mean, moving_variance)."""
    scope = name if name is not None else g.unique('batch_normalization')
    C = x.data.shape[-1]
    gamma = g.variable(scope + '/gamma', (C,), g.ones)
    beta = g.variable(scope + '/beta', (C,), g.zeros)
    mm = g.state(scope + '/moving_mean', (C,), 0.0)
    mv = g.state(scope + '/moving_variance', (C,), 1.0)
    out = nn.batch_normalization(g.tape, x, gamma, beta, mm, mv, training)
    out.tag = scope              # relu() finds a pinned decision mask under this name (nn.Tape.pins)
    return out


def layers_conv3d(g, x, filters, kernel, strides, name=None):
    """tf.layers.conv3d(x, filters, kernel, strides, 'same'): glorot-uniform
    kernel, zero bias (Appendix A.7)."""
    scope = name if name is not None else g.unique('conv3d')
    k = (kernel,) * 3 if isinstance(kernel, int) else tuple(kernel)
    s = (strides,) * 3 if isinstance(strides, int) else tuple(strides)
    w = g.variable(scope + '/kernel', k + (x.data.shape[-1], filters), g.xavier)
    b = g.variable(scope + '/bias', (filters,), g.zeros)
    return nn.conv3d(g.tape, x, w, s, b)


def layers_conv3d_transpose(g, x, filters, kernel, strides, name=None):
    """tf.layers.conv3d_transpose(x, filters, kernel, strides, 'same'): kernel
    variable is [kd,kh,kw,Cout,Cin] (Appendix A.3)."""
    scope = name if name is not None else g.unique('conv3d_transpose')
    k = (kernel,) * 3 if isinstance(kernel, int) else tuple(kernel)
    s = (strides,) * 3 if isinstance(strides, int) else tuple(strides)
    w = g.variable(scope + '/kernel', k + (filters, x.data.shape[-1]), g.xavier)
    b = g.variable(scope + '/bias', (filters,), g.zeros)
    return nn.conv3d_transpose(g.tape, x, w, s, b)


def max_pool3d(g, x, ksize, strides):
    return nn.max_pool3d(g.tape, x, tuple(ksize[1:4]), tuple(strides[1:4]))


# ---------------------------------------------------------------------------
# p3d.py:30-136
# ---------------------------------------------------------------------------
class Bottleneck():
    def __init__(self, g, l_input, inplanes, planes, stride=1, downsample='', training=True, n_s=0, depth_3d=47):
        """p3d.py:31-54 (stride bookkeeping; the 2-D branch n_s>=depth_3d is
        unreachable with 3+8+36 = 47 blocks and is not restated)."""
        self.g = g
        self.X_input = l_input
        self.downsample = downsample
        self.planes = planes
        self.inplanes = inplanes
        self.depth_3d = depth_3d
        self.ST_struc = ('A', 'B', 'C')
        self.len_ST = len(self.ST_struc)
        self.id = n_s
        self.n_s = n_s
        self.ST = list(self.ST_struc)[self.id % self.len_ST]
        self.stride_p = [1, 1, 1, 1, 1]
        self.training = training
        if self.downsample != '':
            self.stride_p = [1, 1, 2, 2, 1]
        assert n_s < self.depth_3d, "2-D bottlenecks are dead code in the reference"
        if n_s == 0:
            self.stride_p = [1, 1, 1, 1, 1]

    def _bn_relu(self, x):
        g = self.g
        return nn.relu(g.tape, batch_normalization(g, x, self.training))

    def ST_A(self, name, x):
        """p3d.py:56-63: S -> BN -> ReLU -> T -> BN -> ReLU."""
        x = self._bn_relu(convS(self.g, name + '_S', x, self.planes, self.planes))
        x = self._bn_relu(convT(self.g, name + '_T', x, self.planes, self.planes))
        return x

    def ST_B(self, name, x):
        """p3d.py:65-72: relu(BN(S(x))) + relu(BN(T(x))); S-branch BN created first."""
        tmp_x = self._bn_relu(convS(self.g, name + '_S', x, self.planes, self.planes))
        x = self._bn_relu(convT(self.g, name + '_T', x, self.planes, self.planes))
        return nn.add(self.g.tape, x, tmp_x)

    def ST_C(self, name, x):
        """p3d.py:74-81: s = relu(BN(S(x))); s + relu(BN(T(s)))."""
        x = self._bn_relu(convS(self.g, name + '_S', x, self.planes, self.planes))
        tmp_x = self._bn_relu(convT(self.g, name + '_T', x, self.planes, self.planes))
        return nn.add(self.g.tape, x, tmp_x)

    def infer(self):
        """p3d.py:83-136."""
        g = self.g
        t = g.tape
        residual = self.X_input
        out = nn.conv3d(t, self.X_input,
                        get_conv_weight(g, 'conv3_{}_1'.format(self.id), [1, 1, 1, self.inplanes, self.planes]),
                        tuple(self.stride_p[1:4]))
        out = batch_normalization(g, out, self.training)
        out = nn.relu(t, out)
        t.tap('block{}/conv1_bn_relu'.format(self.id), out)
        if self.ST == 'A':
            out = self.ST_A('STA_{}_2'.format(self.id), out)
        elif self.ST == 'B':
            out = self.ST_B('STB_{}_2'.format(self.id), out)
        elif self.ST == 'C':
            out = self.ST_C('STC_{}_2'.format(self.id), out)
        t.tap('block{}/st'.format(self.id), out)
        out = nn.conv3d(t, out,
                        get_conv_weight(g, 'conv3_{}_3'.format(self.id),
                                        [1, 1, 1, self.planes, self.planes * BLOCK_EXPANSION]),
                        (1, 1, 1))
        out = batch_normalization(g, out, self.training)
        if len(self.downsample) == 2:
            residual = nn.conv3d(t, residual,
                                 get_conv_weight(g, 'dw3d_{}'.format(self.id),
                                                 [1, 1, 1, self.inplanes, self.planes * BLOCK_EXPANSION]),
                                 tuple(self.downsample[1][1:4]))
            residual = batch_normalization(g, residual, self.training)
        out = nn.add(t, out, residual)
        out = nn.relu(t, out)
        t.tap('block{}/out'.format(self.id), out)
        return out


class make_block():
    def __init__(self, g, _X, planes, num, inplanes, cnt, training=True, depth_3d=47, stride=1):
        """p3d.py:140-158.  NOTE the call sites (p3d.py:179,185,191) never pass
        `training`, so backbone BN always uses batch statistics."""
        self.g = g
        self.input = _X
        self.planes = planes
        self.inplanes = inplanes
        self.num = num
        self.cnt = cnt
        self.depth_3d = depth_3d
        self.stride = stride
        self.training = training
        self.downsample = ''
        if self.cnt == 0:
            stride_p = [1, 1, 1, 1, 1]
        else:
            stride_p = [1, 1, 2, 2, 1]
        if stride != 1 or inplanes != planes * BLOCK_EXPANSION:
            self.downsample = ['3d', stride_p]

    def infer(self):
        """p3d.py:159-166."""
        x = Bottleneck(self.g, self.input, self.inplanes, self.planes, self.stride, self.downsample,
                       training=self.training, n_s=self.cnt, depth_3d=self.depth_3d).infer()
        self.cnt += 1
        self.inplanes = BLOCK_EXPANSION * self.planes
        for i in range(1, self.num):
            x = Bottleneck(self.g, x, self.inplanes, self.planes, training=self.training,
                           n_s=self.cnt, depth_3d=self.depth_3d).infer()
            self.cnt += 1
        return x


def _encoder(g, _X, training, cfg):
    """p3d.py:170-195 (shared verbatim by every head).  Returns the tensors the
    heads consume."""
    t = g.tape
    b = cfg.base
    cnt = 0
    conv1_custom = nn.conv3d(t, _X, get_conv_weight(g, 'firstconv1', [1, 7, 7, 3, b]), (1, 2, 2))
    t.tap('conv1_custom', conv1_custom)
    conv1_custom_bn = batch_normalization(g, conv1_custom, training)
    conv1_custom_bn_relu = nn.relu(t, conv1_custom_bn)
    t.tap('conv1_custom_bn_relu', conv1_custom_bn_relu)
    pool1 = max_pool3d(g, conv1_custom_bn_relu, [1, 2, 3, 3, 1], [1, 2, 2, 2, 1])
    t.tap('pool1', pool1)
    b1 = make_block(g, pool1, b, cfg.blocks[0], b, cnt, depth_3d=cfg.depth_3d)
    res1 = b1.infer()
    cnt = b1.cnt
    pool2 = max_pool3d(g, res1, [1, 2, 1, 1, 1], [1, 2, 1, 1, 1])
    t.tap('pool2', pool2)
    b2 = make_block(g, pool2, 2 * b, cfg.blocks[1], 4 * b, cnt, depth_3d=cfg.depth_3d, stride=2)
    res2 = b2.infer()
    cnt = b2.cnt
    pool3 = max_pool3d(g, res2, [1, 2, 1, 1, 1], [1, 2, 1, 1, 1])
    t.tap('pool3', pool3)
    b3 = make_block(g, pool3, 4 * b, cfg.blocks[2], 8 * b, cnt, depth_3d=cfg.depth_3d, stride=2)
    res3 = b3.infer()
    pool4 = max_pool3d(g, res3, [1, 2, 1, 1, 1], [1, 2, 1, 1, 1])
    t.tap('pool4', pool4)
    return conv1_custom_bn_relu, pool2, pool3, pool4


def p3d_unet(g, _X, _dropout, batch_size=2, training=True, cfg=None, keep_mask=None):
    """p3d.py:169-221.  pool1_concat / deconv3_concat (p3d.py:176,213) are dead
    values in the reference and are skipped."""
    cfg = cfg or REFERENCE_CFG
    t = g.tape
    b = cfg.base
    _, pool2, pool3, pool4 = _encoder(g, _X, training, cfg)
    deconv1 = layers_conv3d_transpose(g, pool4, 8 * b, [1, 3, 3], [2, 2, 2])
    deconv1_bn = batch_normalization(g, deconv1, training, name='deconv1_bn')
    deconv1_re = nn.relu(t, deconv1_bn)
    deconv1_concat = nn.concat(t, [deconv1_re, pool3])
    deconv2 = layers_conv3d_transpose(g, deconv1_concat, 4 * b, [2, 3, 3], [2, 2, 2])
    deconv2_bn = batch_normalization(g, deconv2, training, name='deconv2_bn')
    deconv2_re = nn.relu(t, deconv2_bn)
    deconv2_concat = nn.concat(t, [deconv2_re, pool2])
    deconv3 = layers_conv3d_transpose(g, deconv2_concat, 2 * b, 3, [2, 2, 2])
    deconv3_bn = batch_normalization(g, deconv3, training, name='deconv3_bn')
    deconv3_re = nn.relu(t, deconv3_bn)
    t.tap('deconv3_re', deconv3_re)
    deconv3_drop = nn.dropout(t, deconv3_re, _dropout, training, keep_mask)
    deconv4_conv1 = layers_conv3d(g, deconv3_drop, b // 2, 1, 1)
    t.tap('deconv4_conv1', deconv4_conv1)
    results = layers_conv3d_transpose(g, deconv4_conv1, 1, 3, [2, 2, 2])
    results = nn.sigmoid(t, results)
    return results


def p3d_concat(g, _X, _dropout, batch_size=2, training=True, cfg=None, keep_mask=None):
    """p3d.py:224-276.  NOTE: no sigmoid on this head (p3d.py:275-276), and the
    deconv_pool2/3 layers sit between the stages so the unnamed backbone BN
    counters are unaffected (they are all named)."""
    cfg = cfg or REFERENCE_CFG
    t = g.tape
    b = cfg.base
    # The reference interleaves the skip deconvs with the encoder stages; all of
    # them carry explicit names, so building them after the encoder creates the
    # same variables.
    _, pool2, pool3, pool4 = _encoder(g, _X, training, cfg)

    def up(x, filters, strides, name, bn_name):
        y = layers_conv3d_transpose(g, x, filters, 3, strides, name=name)
        y = batch_normalization(g, y, training, name=bn_name)
        return nn.relu(t, y)
    deconv_pool2 = up(pool2, 2 * b, [1, 1, 1], 'deconv_pool2', 'deconv_pool2_bn')
    deconv_pool3 = up(pool3, 4 * b, [2, 2, 2], 'deconv_pool3', 'deconv_pool3_bn')
    deconv_pool4 = up(pool4, 8 * b, [4, 4, 4], 'deconv_pool4', 'deconv_pool4_bn')
    concatenator = nn.concat(t, [deconv_pool2, deconv_pool3, deconv_pool4])
    conv_concat = layers_conv3d(g, concatenator, 8 * b, 3, 1, name='conv_concat')
    conv_concat = nn.relu(t, batch_normalization(g, conv_concat, training, name='conv_concat_bn'))
    deconv1_revise = layers_conv3d_transpose(g, conv_concat, 2 * b, 3, 2, name='deconv_revise')
    deconv1_revise = nn.relu(t, batch_normalization(g, deconv1_revise, training, name='deconv1_revise_bn'))
    deconv1_revise = nn.dropout(t, deconv1_revise, _dropout, training, keep_mask)
    results = layers_conv3d_transpose(g, deconv1_revise, 1, 3, 2, name='predict_revise')
    return results


def p3d_unetplusplus_nonsa(g, _X, _dropout, batch_size=2, training=True, cfg=None, keep_mask=None):
    """p3d.py:401-459: the nested (UNet++) head without the attention blocks.
    Layer wrappers are utils/network.py:97-110: tf.layers.conv3d / conv3d_transpose
    (named), an UNNAMED tf.layers.batch_normalization that follows `training`
    (so it continues the backbone's batch_normalization_<n> counter in call
    order), ReLU.  The x_1_0 temporal pool of the stem output (p3d.py:408) is
    created before pool1, which changes no variable."""
    cfg = cfg or REFERENCE_CFG
    t = g.tape
    b = cfg.base
    stem, x_2_0, x_3_0, x_4_0 = _encoder(g, _X, training, cfg)
    x_1_0 = max_pool3d(g, stem, [1, 2, 1, 1, 1], [1, 2, 1, 1, 1])
    t.tap('x_1_0', x_1_0)

    def transpose_conv3d(x, channel, kernel, strides, name):      # utils/network.py:106-110
        y = layers_conv3d_transpose(g, x, channel, kernel, strides, name=name)
        y = nn.relu(t, batch_normalization(g, y, training))
        t.tap(name, y)
        return y

    def conv3d(x, channel, kernel, strides, name):                # utils/network.py:100-104
        y = layers_conv3d(g, x, channel, kernel, strides, name=name)
        y = nn.relu(t, batch_normalization(g, y, training))
        t.tap(name, y)
        return y

    def concat(xs):                                               # utils/network.py:97-98
        return nn.concat(t, xs)
    upx_4_0 = transpose_conv3d(x_4_0, 8 * b, [1, 3, 3], [2, 2, 2], 'upx_4_0')
    x_3_1 = conv3d(concat([x_3_0, upx_4_0]), 8 * b, [2, 3, 3], [1, 1, 1], 'x_3_1')
    upx_3_0 = transpose_conv3d(x_3_0, 4 * b, [2, 3, 3], [2, 2, 2], 'upx_3_0')
    x_2_1 = conv3d(concat([x_2_0, upx_3_0]), 4 * b, [3, 3, 3], [1, 1, 1], 'x_2_1')
    upx_3_1 = transpose_conv3d(x_3_1, 4 * b, [2, 3, 3], [2, 2, 2], 'upx_3_1')
    x_2_2 = conv3d(concat([x_2_1, upx_3_1]), 4 * b, [3, 3, 3], [1, 1, 1], 'x_2_2')
    upx_2_0 = transpose_conv3d(x_2_0, 2 * b, [3, 3, 3], [2, 2, 2], 'upx_2_0')
    x_1_1 = conv3d(concat([x_1_0, upx_2_0]), 2 * b, [3, 3, 3], [1, 1, 1], 'x_1_1')
    upx_2_1 = transpose_conv3d(x_2_1, 2 * b, [3, 3, 3], [2, 2, 2], 'upx_2_1')
    x_1_2 = conv3d(concat([x_1_1, upx_2_1]), 2 * b, [3, 3, 3], [1, 1, 1], 'x_1_2')
    upx_2_2 = transpose_conv3d(x_2_2, 2 * b, [3, 3, 3], [2, 2, 2], 'upx_2_2')
    x_1_3 = conv3d(concat([x_1_2, upx_2_2]), 2 * b, [3, 3, 3], [1, 1, 1], 'x_1_3')
    x_1_3 = nn.dropout(t, x_1_3, _dropout, training, keep_mask)
    x_0_1 = layers_conv3d_transpose(g, x_1_3, 1, 3, 2, name='x_0_1')
    return nn.sigmoid(t, x_0_1)


def attention(g, x, name, training, subsample=False, sub_size=2):
    """utils/network.py:157-192 (SAGAN-style self attention, mode='bn').  f, g, h are unnamed tf.layers.conv3d
    inside variable_scope(name) -> name/conv3d, name/conv3d_1, name/conv3d_2; the output conv is an unnamed
    top-level tf.layers.conv3d (conv3d, conv3d_1, ... in call order), its BatchNorm an unnamed top-level
    batch_normalization that follows `training`; the mixing scalar is the top-level variable 'gamma'+name,
    initialised to 0.  Python-2 integer divisions (sub_size/2, each*2/sub_size) are written as //.
    pool3d = tf.layers.max_pooling3d(v, s, s), 'valid' (utils/network.py:6-7); size 1 is the identity."""
    t = g.tape
    B, D, H, W, ch = x.data.shape
    inter = max(1, ch // 8)
    f = layers_conv3d(g, x, inter, 1, 1, name=name + '/conv3d')
    gq = layers_conv3d(g, x, inter, 1, 1, name=name + '/conv3d_1')
    h = layers_conv3d(g, x, ch, 1, 1, name=name + '/conv3d_2')
    if subsample:
        def pool3d(v, s):
            if s == 1:
                return v
            assert all(e % s == 0 for e in v.data.shape[1:4]), "valid pooling = SAME pooling only for divisible extents"
            return nn.max_pool3d(t, v, (s, s, s), (s, s, s))
        f = pool3d(f, sub_size)
        gq = pool3d(gq, sub_size // 2)
        h = pool3d(h, sub_size)
    flat = lambda v: nn.reshape(t, v, (B, -1, v.data.shape[-1]))            # hw_flatten, utils/network.py:194-195
    s = nn.matmul(t, flat(gq), flat(f), transpose_b=True)
    beta = nn.softmax(t, s)
    o = nn.matmul(t, beta, flat(h))
    o = nn.reshape(t, o, (B,) + tuple(e * 2 // sub_size for e in (D, H, W)) + (ch,))
    o = layers_conv3d(g, o, ch, 1, sub_size // 2)                          # unnamed, top level
    o = nn.relu(t, batch_normalization(g, o, training))
    gamma = g.variable('gamma' + name, (1,), g.zeros)
    return nn.add(t, nn.mul(t, o, gamma), x)


def p3d_unetplusplus_ds(g, _X, _dropout, batch_size=2, training=True, cfg=None, keep_mask=None):
    """p3d.py:340-397: the UNet++ head WITH self attention that can actually be built (p3d_unetplusplus, p3d.py:280,
    adds a quarter-resolution tensor to a full-resolution one at p3d.py:334 and cannot).
    = p3d_unetplusplus_nonsa + attention on x_4_0, x_3_1, x_2_2 (full) and x_1_3 (keys/values pooled by 2)."""
    cfg = cfg or REFERENCE_CFG
    t = g.tape
    b = cfg.base
    stem, x_2_0, x_3_0, x_4_0 = _encoder(g, _X, training, cfg)
    x_1_0 = max_pool3d(g, stem, [1, 2, 1, 1, 1], [1, 2, 1, 1, 1])

    def transpose_conv3d(x, channel, kernel, strides, name):
        y = layers_conv3d_transpose(g, x, channel, kernel, strides, name=name)
        return t.tap(name, nn.relu(t, batch_normalization(g, y, training)))

    def conv3d(x, channel, kernel, strides, name):
        y = layers_conv3d(g, x, channel, kernel, strides, name=name)
        return t.tap(name, nn.relu(t, batch_normalization(g, y, training)))

    def concat(xs):
        return nn.concat(t, xs)

    def sa(x, name, **kw):
        return t.tap(name, attention(g, x, name, training, **kw))
    x_4_0 = sa(x_4_0, 'x_4_0_sa')
    upx_4_0 = transpose_conv3d(x_4_0, 8 * b, [1, 3, 3], [2, 2, 2], 'upx_4_0')
    x_3_1 = conv3d(concat([x_3_0, upx_4_0]), 8 * b, [2, 3, 3], [1, 1, 1], 'x_3_1')
    x_3_1 = sa(x_3_1, 'x_3_1_sa')
    upx_3_0 = transpose_conv3d(x_3_0, 4 * b, [2, 3, 3], [2, 2, 2], 'upx_3_0')
    x_2_1 = conv3d(concat([x_2_0, upx_3_0]), 4 * b, [3, 3, 3], [1, 1, 1], 'x_2_1')
    upx_3_1 = transpose_conv3d(x_3_1, 4 * b, [2, 3, 3], [2, 2, 2], 'upx_3_1')
    x_2_2 = conv3d(concat([x_2_1, upx_3_1]), 4 * b, [3, 3, 3], [1, 1, 1], 'x_2_2')
    x_2_2 = sa(x_2_2, 'x_2_2_sa')
    upx_2_0 = transpose_conv3d(x_2_0, 2 * b, [3, 3, 3], [2, 2, 2], 'upx_2_0')
    x_1_1 = conv3d(concat([x_1_0, upx_2_0]), 2 * b, [3, 3, 3], [1, 1, 1], 'x_1_1')
    upx_2_1 = transpose_conv3d(x_2_1, 2 * b, [3, 3, 3], [2, 2, 2], 'upx_2_1')
    x_1_2 = conv3d(concat([x_1_1, upx_2_1]), 2 * b, [3, 3, 3], [1, 1, 1], 'x_1_2')
    upx_2_2 = transpose_conv3d(x_2_2, 2 * b, [3, 3, 3], [2, 2, 2], 'upx_2_2')
    x_1_3 = conv3d(concat([x_1_2, upx_2_2]), 2 * b, [3, 3, 3], [1, 1, 1], 'x_1_3')
    x_1_3 = sa(x_1_3, 'x_1_3_sa', subsample=True)
    x_1_3 = nn.dropout(t, x_1_3, _dropout, training, keep_mask)
    x_0_1 = layers_conv3d_transpose(g, x_1_3, 1, 3, 2, name='x_0_1')
    return nn.sigmoid(t, x_0_1)


STRUCTURES = {'unet': p3d_unet, 'concat': p3d_concat,      # train.py:149-154
              'unet++nonsa': p3d_unetplusplus_nonsa,       # p3d.py:401 (unet++ minus attention)
              'unet++ds': p3d_unetplusplus_ds}             # p3d.py:340 (unet++ with the attention blocks that build)


# ---------------------------------------------------------------------------
# train.py:143-172 as one function: forward, loss, backward, Adam, BN updates
# ---------------------------------------------------------------------------
def synthetic_clip(seed, shape):
    """Synthetic input with the loader's value law (dataflow.py:204-208):
    (RGB uint8 - [90,102,98]) / 255 ; SURVEY.md section 8(d)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    u8 = rng.integers(0, 256, size=shape).astype(np.float32)
    return ((u8 - np.array([90, 102, 98], np.float32)) / 255.0).astype(np.float32)


def synthetic_target(seed, shape):
    rng = np.random.Generator(np.random.PCG64(seed))
    return rng.random(size=shape, dtype=np.float32)


def forward(params, x, dropout=0.0, training=False, structure='unet', cfg=None, dtype=np.float32):
    """The eval fetch of train.py:225-226 / gen_pred.py:151."""
    g = Graph(params, dtype=dtype, create=False)
    pred = STRUCTURES[structure](g, nn.Var(x.astype(dtype)), dropout, x.shape[0], training, cfg)
    return pred.data, g


def loss_and_grads(params, x, y, dropout=0.0, training=True, structure='unet', cfg=None, dtype=np.float32,
                   keep_mask=None, pins=None):
    """Forward + Smooth-L1-sum loss (train.py:156-159) + backward.  Returns
    (loss, pred, grads{name->array}, graph).  keep_mask: the dropout keep pattern when dropout > 0.
    pins: ReLU / max-pool decisions of another evaluation of this graph (nn.Tape.pins)."""
    g = Graph(params, dtype=dtype, create=False)
    g.tape.pins = pins
    X = nn.Var(x.astype(dtype))
    pred = STRUCTURES[structure](g, X, dropout, x.shape[0], training, cfg, keep_mask)
    pred_reshape = nn.reshape(g.tape, pred, y.shape)
    loss = nn.smooth_l1_loss(g.tape, pred_reshape, y.astype(dtype), 1, 1, sigma=1.0)
    g.tape.backward(loss)
    grads = OrderedDict((n, v.grad) for n, v in g.trainable.items())
    return float(loss.data), pred.data, grads, g


def train_step(params, adam_state, x, y, lr=1e-4, dropout=0.0, structure='unet', cfg=None, dtype=np.float32):
    """One sess.run([train_op, loss]) of train.py:217-218: Adam on every
    trainable + BN moving-average updates.  adam_state = {'t': int, 'm': {}, 'v': {}}."""
    loss, pred, grads, g = loss_and_grads(params, x, y, dropout, True, structure, cfg, dtype)
    adam_state['t'] += 1
    for n, gr in grads.items():
        m = adam_state['m'].setdefault(n, np.zeros_like(params[n]))
        v = adam_state['v'].setdefault(n, np.zeros_like(params[n]))
        nn.adam_step(params[n], gr, m, v, adam_state['t'], lr=lr)
    g.tape.apply_updates()
    return loss, pred


def init_params(seed=1, structure='unet', cfg=None, input_shape=(1, 16, 32, 32, 3), dtype=np.float32):
    """Create every variable of a structure with the reference's initialisers by
    tracing one forward pass on zeros (what tf.global_variables_initializer
    would fill, train.py:178).  Shapes do not depend on the clip extent, so a small
    clip is traced."""
    g = Graph(rng=np.random.default_rng(seed), dtype=dtype, create=True)
    STRUCTURES[structure](g, nn.Var(np.zeros(input_shape, dtype)), 0.0, input_shape[0], False, cfg)
    return g.params
