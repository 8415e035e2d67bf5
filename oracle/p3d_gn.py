"""CPU restatement of the reference's GroupNorm + CBAM P3D network (gn/p3d_gn.py, BASELINE config 4).

TEST INFRASTRUCTURE ONLY (see oracle/nn.py header).  PARITY UNPINNED.

Names follow /root/reference/gn/p3d_gn.py and utils/network.py:198-274.  Differences from the
BatchNorm file (oracle/p3d.py): every batch_normalization is GroupNorm (gn/p3d_gn.py:24-46), the
residual of EVERY bottleneck goes through cbam_block before the add (gn/p3d_gn.py:175-177), and the
head is inference_p3d (gn/p3d_gn.py:214-258): no sigmoid.
"""
import numpy as np

from . import nn
from .p3d import (BLOCK_EXPANSION, Graph, NetConfig, REFERENCE_CFG, convS, convT, get_conv_weight,
                  layers_conv3d, layers_conv3d_transpose, max_pool3d)


def GroupNorm(g, x, G=32, esp=1e-5):
    """gn/p3d_gn.py:24-46: variables live in scope group_norm, group_norm_1, ... (tf.Variable in a
    uniquified variable_scope)."""
    scope = g.unique('group_norm')
    C = x.data.shape[-1]
    gamma = g.variable(scope + '/gamma', (C,), g.ones)
    beta = g.variable(scope + '/beta', (C,), g.zeros)
    out = nn.group_norm(g.tape, x, gamma, beta, G, esp)
    out.tag = scope              # relu() finds a pinned decision mask under this name (nn.Tape.pins)
    return out


def GNReLU(g, x):
    """gn/p3d_gn.py:49-51."""
    return nn.relu(g.tape, GroupNorm(g, x))


def _vs(g):
    return lambda shape: nn.variance_scaling_normal(g.rng, shape, g.dtype)


def channel_attention(g, input_feature, name, ratio=8):
    """utils/network.py:208-249: shared MLP on the (D,H,W)-mean and -max, sigmoid of the sum, scale."""
    t = g.tape
    C = input_feature.data.shape[-1]
    k0 = g.variable(name + '/mlp_0/kernel', (C, C // ratio), _vs(g))
    b0 = g.variable(name + '/mlp_0/bias', (C // ratio,), g.zeros)
    k1 = g.variable(name + '/mlp_1/kernel', (C // ratio, C), _vs(g))
    b1 = g.variable(name + '/mlp_1/bias', (C,), g.zeros)

    def mlp(v):
        return nn.dense(t, nn.relu(t, nn.dense(t, v, k0, b0)), k1, b1)
    avg_pool = mlp(nn.reduce_mean(t, input_feature, (1, 2, 3)))
    max_pool = mlp(nn.reduce_max(t, input_feature, (1, 2, 3)))
    scale = nn.sigmoid(t, nn.add(t, avg_pool, max_pool))
    return nn.mul(t, input_feature, scale)


def spatial_attention(g, input_feature, name):
    """utils/network.py:251-274: channel-mean and channel-max maps -> 7x7x7 conv (2->1, no bias) -> sigmoid."""
    t = g.tape
    avg_pool = nn.reduce_mean(t, input_feature, (4,))
    max_pool = nn.reduce_max(t, input_feature, (4,))
    concat = nn.concat(t, [avg_pool, max_pool])
    k = g.variable(name + '/conv3d/kernel', (7, 7, 7, 2, 1), _vs(g))
    concat = nn.sigmoid(t, nn.conv3d(t, concat, k, (1, 1, 1)))
    return nn.mul(t, input_feature, concat)


def cbam_block(g, input_feature, name, ratio=8):
    """utils/network.py:198-206."""
    attention_feature = channel_attention(g, input_feature, name + '/ch_at', ratio)
    attention_feature = spatial_attention(g, attention_feature, name + '/sp_at')
    return attention_feature


class Bottleneck():
    def __init__(self, g, l_input, inplanes, planes, stride=1, downsample='', n_s=0, depth_3d=47):
        """gn/p3d_gn.py:75-98 (same stride bookkeeping as the BN file)."""
        self.g = g
        self.X_input = l_input
        self.downsample = downsample
        self.planes = planes
        self.inplanes = inplanes
        self.id = n_s
        self.ST = 'ABC'[n_s % 3]
        self.stride_p = [1, 1, 1, 1, 1]
        if self.downsample != '':
            self.stride_p = [1, 1, 2, 2, 1]
        assert n_s < depth_3d
        if n_s == 0:
            self.stride_p = [1, 1, 1, 1, 1]

    def ST_A(self, name, x):
        """gn/p3d_gn.py:100-107."""
        x = GNReLU(self.g, convS(self.g, name + '_S', x, self.planes, self.planes))
        return GNReLU(self.g, convT(self.g, name + '_T', x, self.planes, self.planes))

    def ST_B(self, name, x):
        """gn/p3d_gn.py:109-116."""
        tmp_x = GNReLU(self.g, convS(self.g, name + '_S', x, self.planes, self.planes))
        x = GNReLU(self.g, convT(self.g, name + '_T', x, self.planes, self.planes))
        return nn.add(self.g.tape, x, tmp_x)

    def ST_C(self, name, x):
        """gn/p3d_gn.py:118-125."""
        x = GNReLU(self.g, convS(self.g, name + '_S', x, self.planes, self.planes))
        tmp_x = GNReLU(self.g, convT(self.g, name + '_T', x, self.planes, self.planes))
        return nn.add(self.g.tape, x, tmp_x)

    def infer(self):
        """gn/p3d_gn.py:127-179."""
        g, t = self.g, self.g.tape
        residual = self.X_input
        out = nn.conv3d(t, self.X_input,
                        get_conv_weight(g, 'conv3_{}_1'.format(self.id), [1, 1, 1, self.inplanes, self.planes]),
                        tuple(self.stride_p[1:4]))
        out = GNReLU(g, out)
        t.tap('block{}/conv1_bn_relu'.format(self.id), out)
        out = getattr(self, 'ST_' + self.ST)('ST{}_{}_2'.format(self.ST, self.id), out)
        t.tap('block{}/st'.format(self.id), out)
        out = nn.conv3d(t, out, get_conv_weight(g, 'conv3_{}_3'.format(self.id),
                                                [1, 1, 1, self.planes, self.planes * BLOCK_EXPANSION]), (1, 1, 1))
        out = GroupNorm(g, out)
        if len(self.downsample) == 2:
            residual = nn.conv3d(t, residual,
                                 get_conv_weight(g, 'dw3d_{}'.format(self.id),
                                                 [1, 1, 1, self.inplanes, self.planes * BLOCK_EXPANSION]),
                                 tuple(self.downsample[1][1:4]))
            residual = GroupNorm(g, residual)
        residual = cbam_block(g, residual, 'cbam_{}'.format(self.id))      # gn/p3d_gn.py:175
        t.tap('block{}/cbam'.format(self.id), residual)
        out = nn.relu(t, nn.add(t, out, residual))
        t.tap('block{}/out'.format(self.id), out)
        return out


class make_block():
    def __init__(self, g, _X, planes, num, inplanes, cnt, depth_3d=47, stride=1):
        """gn/p3d_gn.py:183-201."""
        self.g, self.input, self.planes, self.inplanes, self.num, self.cnt = g, _X, planes, inplanes, num, cnt
        self.depth_3d, self.stride = depth_3d, stride
        self.downsample = ''
        stride_p = [1, 1, 1, 1, 1] if self.cnt == 0 else [1, 1, 2, 2, 1]
        if stride != 1 or inplanes != planes * BLOCK_EXPANSION:
            self.downsample = ['3d', stride_p]

    def infer(self):
        """gn/p3d_gn.py:202-209."""
        x = Bottleneck(self.g, self.input, self.inplanes, self.planes, self.stride, self.downsample,
                       n_s=self.cnt, depth_3d=self.depth_3d).infer()
        self.cnt += 1
        self.inplanes = BLOCK_EXPANSION * self.planes
        for i in range(1, self.num):
            x = Bottleneck(self.g, x, self.inplanes, self.planes, n_s=self.cnt, depth_3d=self.depth_3d).infer()
            self.cnt += 1
        return x


def inference_p3d(g, _X, _dropout, batch_size=2, training=True, cfg=None, keep_mask=None, pool4_filters=16):
    """gn/p3d_gn.py:214-258.  GroupNorm has no train/eval difference; `training` only gates dropout.
    pool4_filters (in units of cfg.base) is 16 here and 8 in inference_p3d_concat, the only difference."""
    cfg = cfg or REFERENCE_CFG
    t = g.tape
    b = cfg.base
    cnt = 0
    conv1_custom = nn.conv3d(t, _X, get_conv_weight(g, 'firstconv1', [1, 7, 7, 3, b]), (1, 2, 2))
    conv1_custom_bn_relu = nn.relu(t, GroupNorm(g, conv1_custom))
    t.tap('conv1_custom_bn_relu', conv1_custom_bn_relu)
    pool1 = max_pool3d(g, conv1_custom_bn_relu, [1, 2, 3, 3, 1], [1, 2, 2, 2, 1])
    b1 = make_block(g, pool1, b, cfg.blocks[0], b, cnt, depth_3d=cfg.depth_3d)
    res1 = b1.infer()
    cnt = b1.cnt
    pool2 = max_pool3d(g, res1, [1, 2, 1, 1, 1], [1, 2, 1, 1, 1])
    b2 = make_block(g, pool2, 2 * b, cfg.blocks[1], 4 * b, cnt, depth_3d=cfg.depth_3d, stride=2)
    res2 = b2.infer()
    cnt = b2.cnt
    pool3 = max_pool3d(g, res2, [1, 2, 1, 1, 1], [1, 2, 1, 1, 1])
    deconv_pool3 = layers_conv3d_transpose(g, pool3, 8 * b, 3, [2, 2, 2], name='deconv_pool3')
    deconv_pool3_gn = GNReLU(g, deconv_pool3)                     # created BEFORE stage 3 (GN numbering)
    b3 = make_block(g, pool3, 4 * b, cfg.blocks[2], 8 * b, cnt, depth_3d=cfg.depth_3d, stride=2)
    res3 = b3.infer()
    pool4 = max_pool3d(g, res3, [1, 2, 1, 1, 1], [1, 2, 1, 1, 1])
    deconv_pool4 = layers_conv3d_transpose(g, pool4, pool4_filters * b, 3, [4, 4, 4], name='deconv_pool4')
    deconv_pool4_gn = GNReLU(g, deconv_pool4)
    concatenator = nn.concat(t, [deconv_pool3_gn, deconv_pool4_gn, pool2])
    conv_concat = GNReLU(g, layers_conv3d(g, concatenator, 16 * b, 3, 1, name='conv_concat'))
    t.tap('conv_concat', conv_concat)
    deconv1_revise = GNReLU(g, layers_conv3d_transpose(g, conv_concat, 4 * b, 3, 2, name='deconv_revise'))
    deconv1_revise = nn.dropout(t, deconv1_revise, _dropout, training, keep_mask)
    results = layers_conv3d_transpose(g, deconv1_revise, 1, 3, 2, name='predict_revise')
    return results


def inference_p3d_concat(g, _X, _dropout, batch_size=2, training=True, cfg=None, keep_mask=None):
    """gn/p3d_gn.py:279-324 (net='P3D_CONCAT'): inference_p3d with deconv_pool4 at 512 instead of 1024 filters."""
    return inference_p3d(g, _X, _dropout, batch_size, training, cfg, keep_mask, pool4_filters=8)


def conv3d_layers(g, x, filters, kernel, strides, name):
    """gn/p3d_gn.py:14-17 (xavier_initializer() = glorot uniform, the tf.layers default)."""
    return GNReLU(g, layers_conv3d(g, x, filters, kernel, strides, name=name))


def deconv3d_layers(g, x, filters, kernel, strides, name):
    """gn/p3d_gn.py:19-22."""
    return GNReLU(g, layers_conv3d_transpose(g, x, filters, kernel, strides, name=name))


def inference_p3d_decoder_block(g, _X, _dropout, batch_size=2, training=True, cfg=None, keep_mask=None):
    """gn/p3d_gn.py:489-539 (net='P3D_DECODER', gn/train_p3d_gn_dataset.py:177).  The whole graph is built
    inside tf.variable_scope('P3D'), so every variable name carries the 'P3D/' prefix."""
    cfg = cfg or REFERENCE_CFG
    t = g.tape
    b = cfg.base
    g.prefix = 'P3D/'
    cnt = 0
    conv1_custom = nn.conv3d(t, _X, get_conv_weight(g, 'firstconv1', [1, 7, 7, 3, b]), (1, 2, 2))
    conv1_custom_bn_relu = nn.relu(t, GroupNorm(g, conv1_custom))
    pool1 = max_pool3d(g, conv1_custom_bn_relu, [1, 2, 3, 3, 1], [1, 2, 2, 2, 1])
    b1 = make_block(g, pool1, b, cfg.blocks[0], b, cnt, depth_3d=cfg.depth_3d)
    res1 = b1.infer()
    cnt = b1.cnt
    pool2 = max_pool3d(g, res1, [1, 2, 1, 1, 1], [1, 2, 1, 1, 1])
    deconv_pool2 = deconv3d_layers(g, pool2, 2 * b, [3, 3, 3], [1, 1, 1], 'deconv_pool2')
    t.tap('deconv_pool2', deconv_pool2)
    b2 = make_block(g, pool2, 2 * b, cfg.blocks[1], 4 * b, cnt, depth_3d=cfg.depth_3d, stride=2)
    res2 = b2.infer()
    cnt = b2.cnt
    pool3 = max_pool3d(g, res2, [1, 2, 1, 1, 1], [1, 2, 1, 1, 1])
    deconv_pool3 = deconv3d_layers(g, pool3, 4 * b, [2, 3, 3], [2, 2, 2], 'deconv_pool3')
    t.tap('deconv_pool3', deconv_pool3)
    b3 = make_block(g, pool3, 4 * b, cfg.blocks[2], 8 * b, cnt, depth_3d=cfg.depth_3d, stride=2)
    res3 = b3.infer()
    pool4 = max_pool3d(g, res3, [1, 2, 1, 1, 1], [1, 2, 1, 1, 1])
    deconv_pool4 = deconv3d_layers(g, pool4, 8 * b, [1, 3, 3], [4, 4, 4], 'deconv_pool4')
    t.tap('deconv_pool4', deconv_pool4)
    concatenator = nn.concat(t, [deconv_pool2, deconv_pool3, deconv_pool4])
    x = conv3d_layers(g, concatenator, 16 * b, 3, 1, 'conv_concat')
    t.tap('conv_concat', x)
    for name, fn, filters, stride in (('decoder1_conv1', conv3d_layers, 4 * b, 1),
                                      ('decoder1_deconv', deconv3d_layers, 4 * b, 2),
                                      ('decoder1_conv2', conv3d_layers, 2 * b, 1),
                                      ('decoder2_conv1', conv3d_layers, b // 2, 1),
                                      ('decoder2_deconv', deconv3d_layers, b // 2, 2),
                                      ('decoder2_conv2', conv3d_layers, b // 4, 1)):
        x = fn(g, x, filters, 3, stride, name)
        t.tap(name, x)
    final_conv = nn.dropout(t, x, _dropout, training, keep_mask)
    results = layers_conv3d(g, final_conv, 1, 3, 1, name='results')
    return results


HEADS = {'p3d': inference_p3d,                         # gn/train_p3d_gn_dataset.py:169-170  net='P3D'
         'concat': inference_p3d_concat,               # gn/train_p3d_gn_dataset.py:171-172  net='P3D_CONCAT'
         'decoder': inference_p3d_decoder_block}       # gn/train_p3d_gn_dataset.py:177-178  net='P3D_DECODER'


def init_params(seed=1, cfg=None, input_shape=(1, 16, 32, 32, 3), dtype=np.float32, head='p3d'):
    g = Graph(rng=np.random.default_rng(seed), dtype=dtype, create=True)
    HEADS[head](g, nn.Var(np.zeros(input_shape, dtype)), 0.0, input_shape[0], False, cfg)
    return g.params


def forward(params, x, dropout=0.0, training=False, cfg=None, dtype=np.float32, head='p3d'):
    g = Graph(params, dtype=dtype, create=False)
    pred = HEADS[head](g, nn.Var(x.astype(dtype)), dropout, x.shape[0], training, cfg)
    return pred.data, g


def loss_and_grads(params, x, y, dropout=0.0, training=True, cfg=None, dtype=np.float32, head='p3d', keep_mask=None, pins=None):
    """gn/train_p3d_gn_dataset.py:186: the same Smooth-L1 sum on the raw prediction."""
    from collections import OrderedDict
    g = Graph(params, dtype=dtype, create=False)
    g.tape.pins = pins
    pred = HEADS[head](g, nn.Var(x.astype(dtype)), dropout, x.shape[0], training, cfg, keep_mask)
    loss = nn.smooth_l1_loss(g.tape, nn.reshape(g.tape, pred, y.shape), y.astype(dtype), 1, 1, sigma=1.0)
    g.tape.backward(loss)
    return float(loss.data), pred.data, OrderedDict((n, v.grad) for n, v in g.trainable.items()), g
