"""Independent torch-CPU composition of the same TF-1.x graph (fp32 or fp64).

TEST INFRASTRUCTURE ONLY, like the rest of oracle/ (see oracle/nn.py): it cross-checks the numpy restatement
(tests/test_oracle_vs_torch.py; the two share no code) and, being oneDNN-backed, is the stronger of the two CPU
baselines bench.py reports -- run in a child process by `python -m oracle.cpu_time`.
PARITY UNPINNED upstream (the reference cannot run here).  Semantics per SURVEY.md Appendix A; tensors are NCDHW
inside, NDHWC at the interface."""
import math

import torch
import torch.nn.functional as F


def _same(size, k, s):
    out = -(-size // s)
    pt = max((out - 1) * s + k - size, 0)
    return out, pt // 2, pt - pt // 2


def conv3d_same(x, w, strides, bias=None):
    # x NCDHW, w TF layout [kd,kh,kw,Ci,Co]
    kd, kh, kw = w.shape[:3]
    pads = []
    for size, k, s in zip(x.shape[2:], (kd, kh, kw), strides):
        _, pb, pa = _same(size, k, s)
        pads.append((pb, pa))
    xp = F.pad(x, (pads[2][0], pads[2][1], pads[1][0], pads[1][1], pads[0][0], pads[0][1]))
    return F.conv3d(xp, w.permute(4, 3, 0, 1, 2), bias, stride=tuple(strides))


def conv3d_transpose_same(x, kernel, strides, bias=None):
    # kernel TF layout [kd,kh,kw,Cout,Cin]; torch wants [Cin, Cout, kd,kh,kw]
    y = F.conv_transpose3d(x, kernel.permute(4, 3, 0, 1, 2), None, stride=tuple(strides), padding=0)
    sl = [slice(None), slice(None)]
    for i, (size, k, s) in enumerate(zip(x.shape[2:], kernel.shape[:3], strides)):
        pt = max(k - s, 0)
        start = pt // 2
        want = size * s
        have = y.shape[2 + i]
        if have - start < want:      # k < s: zero-extend at the end
            padn = want - (have - start)
            pad = [0, 0] * (2 - i) + [0, padn]
            y = F.pad(y, pad)
        sl.append(slice(start, start + want))
    y = y[tuple(sl)]
    if bias is not None:
        y = y + bias.view(1, -1, 1, 1, 1)
    return y


def max_pool_same(x, ksize, strides):
    pads = []
    for size, k, s in zip(x.shape[2:], ksize, strides):
        _, pb, pa = _same(size, k, s)
        pads.append((pb, pa))
    xp = F.pad(x, (pads[2][0], pads[2][1], pads[1][0], pads[1][1], pads[0][0], pads[0][1]), value=float('-inf'))
    return F.max_pool3d(xp, tuple(ksize), tuple(strides))


class TorchP3D:
    """p3d_unet / p3d_concat from a {tf_name: numpy} dict; parameters become leaf
    tensors with requires_grad so autograd supplies every gradient."""

    def __init__(self, params, dtype=torch.float32, base=64, blocks=(3, 8, 36)):
        self.p = {}
        self.dtype = dtype
        for k, v in params.items():
            t = torch.tensor(v, dtype=dtype)
            if not (k.endswith('moving_mean') or k.endswith('moving_variance')):
                t.requires_grad_(True)
            self.p[k] = t
        self.base, self.blocks = base, blocks
        self._bn = 0
        self._uniq = {}
        self.new_moving = {}

    def uniq(self, base):
        k = self._uniq.get(base, 0)
        self._uniq[base] = k + 1
        return base if k == 0 else '%s_%d' % (base, k)

    def bn(self, x, training, name=None):
        scope = name or self.uniq('batch_normalization')
        g, b = self.p[scope + '/gamma'], self.p[scope + '/beta']
        mm, mv = self.p[scope + '/moving_mean'], self.p[scope + '/moving_variance']
        if training:
            mean = x.mean(dim=(0, 2, 3, 4))
            var = x.var(dim=(0, 2, 3, 4), unbiased=False)
            self.new_moving[scope + '/moving_mean'] = (mm * 0.99 + mean.detach() * 0.01)
            self.new_moving[scope + '/moving_variance'] = (mv * 0.99 + var.detach() * 0.01)
        else:
            mean, var = mm, mv
        sh = (1, -1, 1, 1, 1)
        return (x - mean.view(sh)) / torch.sqrt(var.view(sh) + 1e-3) * g.view(sh) + b.view(sh)

    def block(self, x, i, inplanes, planes, first, stride2):
        p = self.p
        st = 'ABC'[i % 3]
        s = (1, 2, 2) if (stride2 and first) else (1, 1, 1)
        out = torch.relu(self.bn(conv3d_same(x, p['conv3_%d_1' % i], s), True))
        nm = 'ST%s_%d_2' % (st, i)
        S = lambda t: conv3d_same(t, p[nm + '_S'], (1, 1, 1), p[nm + '_S_bias'])
        T = lambda t: conv3d_same(t, p[nm + '_T'], (1, 1, 1), p[nm + '_T_bias'])
        if st == 'A':
            out = torch.relu(self.bn(S(out), True))
            out = torch.relu(self.bn(T(out), True))
        elif st == 'B':
            a = torch.relu(self.bn(S(out), True))
            b = torch.relu(self.bn(T(out), True))
            out = a + b
        else:
            a = torch.relu(self.bn(S(out), True))
            b = torch.relu(self.bn(T(a), True))
            out = a + b
        out = self.bn(conv3d_same(out, p['conv3_%d_3' % i], (1, 1, 1)), True)
        res = x
        if first:
            res = self.bn(conv3d_same(x, p['dw3d_%d' % i], s), True)
        return torch.relu(out + res)

    def encoder(self, x, training):
        p, b = self.p, self.base
        x = torch.relu(self.bn(conv3d_same(x, p['firstconv1'], (1, 2, 2)), training))
        self.stem = x
        x = max_pool_same(x, (2, 3, 3), (2, 2, 2))
        i = 0
        skips = []
        inpl = b
        for stage, (n, planes) in enumerate(zip(self.blocks, (b, 2 * b, 4 * b))):
            for j in range(n):
                x = self.block(x, i, inpl, planes, j == 0, stage > 0)
                inpl = planes * 4
                i += 1
            x = max_pool_same(x, (2, 1, 1), (2, 1, 1))
            skips.append(x)
        return skips      # pool2, pool3, pool4

    def unet(self, x_ndhwc, training):
        p = self.p
        x = x_ndhwc.permute(0, 4, 1, 2, 3)
        pool2, pool3, pool4 = self.encoder(x, training)
        d = conv3d_transpose_same(pool4, p['conv3d_transpose/kernel'], (2, 2, 2), p['conv3d_transpose/bias'])
        d = torch.relu(self.bn(d, training, 'deconv1_bn'))
        d = torch.cat([d, pool3], 1)
        d = conv3d_transpose_same(d, p['conv3d_transpose_1/kernel'], (2, 2, 2), p['conv3d_transpose_1/bias'])
        d = torch.relu(self.bn(d, training, 'deconv2_bn'))
        d = torch.cat([d, pool2], 1)
        d = conv3d_transpose_same(d, p['conv3d_transpose_2/kernel'], (2, 2, 2), p['conv3d_transpose_2/bias'])
        d = torch.relu(self.bn(d, training, 'deconv3_bn'))
        d = conv3d_same(d, p['conv3d/kernel'], (1, 1, 1), p['conv3d/bias'])
        d = conv3d_transpose_same(d, p['conv3d_transpose_3/kernel'], (2, 2, 2), p['conv3d_transpose_3/bias'])
        return torch.sigmoid(d).permute(0, 2, 3, 4, 1)

    def concat(self, x_ndhwc, training):
        p = self.p
        x = x_ndhwc.permute(0, 4, 1, 2, 3)
        pool2, pool3, pool4 = self.encoder(x, training)

        def up(t, name, s, bn):
            y = conv3d_transpose_same(t, p[name + '/kernel'], s, p[name + '/bias'])
            return torch.relu(self.bn(y, training, bn))
        c = torch.cat([up(pool2, 'deconv_pool2', (1, 1, 1), 'deconv_pool2_bn'),
                       up(pool3, 'deconv_pool3', (2, 2, 2), 'deconv_pool3_bn'),
                       up(pool4, 'deconv_pool4', (4, 4, 4), 'deconv_pool4_bn')], 1)
        c = conv3d_same(c, p['conv_concat/kernel'], (1, 1, 1), p['conv_concat/bias'])
        c = torch.relu(self.bn(c, training, 'conv_concat_bn'))
        c = up(c, 'deconv_revise', (2, 2, 2), 'deconv1_revise_bn')
        c = conv3d_transpose_same(c, p['predict_revise/kernel'], (2, 2, 2), p['predict_revise/bias'])
        return c.permute(0, 2, 3, 4, 1)


def _unetpp_nonsa(self, x_ndhwc, training):
    """p3d.py:401-459 with utils/network.py:97-110 wrappers (unnamed BN follows `training`)."""
    p = self.p
    x = x_ndhwc.permute(0, 4, 1, 2, 3)
    x_2_0, x_3_0, x_4_0 = self.encoder(x, training)
    x_1_0 = max_pool_same(self.stem, (2, 1, 1), (2, 1, 1))

    def up(t, name, s=(2, 2, 2)):
        y = conv3d_transpose_same(t, p[name + '/kernel'], s, p[name + '/bias'])
        return torch.relu(self.bn(y, training))

    def conv(ts, name):
        y = conv3d_same(torch.cat(ts, 1), p[name + '/kernel'], (1, 1, 1), p[name + '/bias'])
        return torch.relu(self.bn(y, training))
    upx_4_0 = up(x_4_0, 'upx_4_0')
    x_3_1 = conv([x_3_0, upx_4_0], 'x_3_1')
    upx_3_0 = up(x_3_0, 'upx_3_0')
    x_2_1 = conv([x_2_0, upx_3_0], 'x_2_1')
    upx_3_1 = up(x_3_1, 'upx_3_1')
    x_2_2 = conv([x_2_1, upx_3_1], 'x_2_2')
    upx_2_0 = up(x_2_0, 'upx_2_0')
    x_1_1 = conv([x_1_0, upx_2_0], 'x_1_1')
    upx_2_1 = up(x_2_1, 'upx_2_1')
    x_1_2 = conv([x_1_1, upx_2_1], 'x_1_2')
    upx_2_2 = up(x_2_2, 'upx_2_2')
    x_1_3 = conv([x_1_2, upx_2_2], 'x_1_3')
    d = conv3d_transpose_same(x_1_3, p['x_0_1/kernel'], (2, 2, 2), p['x_0_1/bias'])
    return torch.sigmoid(d).permute(0, 2, 3, 4, 1)


TorchP3D.unetpp_nonsa = _unetpp_nonsa


def _attention(self, x, name, training, subsample=False, sub_size=2):
    """utils/network.py:157-192 on NCDHW tensors."""
    p = self.p
    B, ch = x.shape[0], x.shape[1]
    c1 = lambda t, n: conv3d_same(t, p[n + '/kernel'], (1, 1, 1), p[n + '/bias'])
    f, g, h = c1(x, name + '/conv3d'), c1(x, name + '/conv3d_1'), c1(x, name + '/conv3d_2')
    if subsample:
        f = F.max_pool3d(f, sub_size, sub_size)
        if sub_size // 2 > 1:
            g = F.max_pool3d(g, sub_size // 2, sub_size // 2)
        h = F.max_pool3d(h, sub_size, sub_size)
    flat = lambda t: t.permute(0, 2, 3, 4, 1).reshape(B, -1, t.shape[1])        # [B, positions, channels]
    beta = torch.softmax(flat(g) @ flat(f).transpose(1, 2), dim=-1)
    o = beta @ flat(h)
    D, H, W = (e * 2 // sub_size for e in x.shape[2:])
    o = o.reshape(B, D, H, W, ch).permute(0, 4, 1, 2, 3)
    o = c1(o, self.uniq('conv3d'))
    o = torch.relu(self.bn(o, training))
    return o * p['gamma' + name] + x


def _unetpp_ds(self, x_ndhwc, training):
    """p3d.py:340-397."""
    p = self.p
    x = x_ndhwc.permute(0, 4, 1, 2, 3)
    x_2_0, x_3_0, x_4_0 = self.encoder(x, training)
    x_1_0 = max_pool_same(self.stem, (2, 1, 1), (2, 1, 1))

    def up(t, name, s=(2, 2, 2)):
        y = conv3d_transpose_same(t, p[name + '/kernel'], s, p[name + '/bias'])
        return torch.relu(self.bn(y, training))

    def conv(ts, name):
        y = conv3d_same(torch.cat(ts, 1), p[name + '/kernel'], (1, 1, 1), p[name + '/bias'])
        return torch.relu(self.bn(y, training))
    x_4_0 = _attention(self, x_4_0, 'x_4_0_sa', training)
    upx_4_0 = up(x_4_0, 'upx_4_0')
    x_3_1 = _attention(self, conv([x_3_0, upx_4_0], 'x_3_1'), 'x_3_1_sa', training)
    upx_3_0 = up(x_3_0, 'upx_3_0')
    x_2_1 = conv([x_2_0, upx_3_0], 'x_2_1')
    upx_3_1 = up(x_3_1, 'upx_3_1')
    x_2_2 = _attention(self, conv([x_2_1, upx_3_1], 'x_2_2'), 'x_2_2_sa', training)
    upx_2_0 = up(x_2_0, 'upx_2_0')
    x_1_1 = conv([x_1_0, upx_2_0], 'x_1_1')
    upx_2_1 = up(x_2_1, 'upx_2_1')
    x_1_2 = conv([x_1_1, upx_2_1], 'x_1_2')
    upx_2_2 = up(x_2_2, 'upx_2_2')
    x_1_3 = _attention(self, conv([x_1_2, upx_2_2], 'x_1_3'), 'x_1_3_sa', training, subsample=True)
    d = conv3d_transpose_same(x_1_3, p['x_0_1/kernel'], (2, 2, 2), p['x_0_1/bias'])
    return torch.sigmoid(d).permute(0, 2, 3, 4, 1)


TorchP3D.unetpp_ds = _unetpp_ds


def smooth_l1_sum(pred, y):
    d = pred - y
    ad = d.abs()
    return torch.where(ad < 1, 0.5 * d * d, ad - 0.5).sum()


class TorchP3DGN(TorchP3D):
    """gn/p3d_gn.py inference_p3d (GroupNorm + CBAM on every residual) from a {tf_name: numpy} dict."""

    def gn(self, x):
        scope = self.uniq('group_norm')
        C = x.shape[1]
        return F.group_norm(x, min(32, C), self.p[scope + '/gamma'], self.p[scope + '/beta'], eps=1e-5)

    def cbam(self, x, name):
        p = self.p
        k0, b0 = p[name + '/ch_at/mlp_0/kernel'], p[name + '/ch_at/mlp_0/bias']
        k1, b1 = p[name + '/ch_at/mlp_1/kernel'], p[name + '/ch_at/mlp_1/bias']
        mlp = lambda v: torch.relu(v @ k0 + b0) @ k1 + b1
        avg = mlp(x.mean(dim=(2, 3, 4)))
        mx = mlp(x.amax(dim=(2, 3, 4)))
        x = x * torch.sigmoid(avg + mx)[:, :, None, None, None]
        sp = torch.cat([x.mean(dim=1, keepdim=True), x.amax(dim=1, keepdim=True)], 1)
        sp = conv3d_same(sp, p[name + '/sp_at/conv3d/kernel'], (1, 1, 1))
        return x * torch.sigmoid(sp)

    def block(self, x, i, inplanes, planes, first, stride2):
        p = self.p
        st = 'ABC'[i % 3]
        s = (1, 2, 2) if (stride2 and first) else (1, 1, 1)
        out = torch.relu(self.gn(conv3d_same(x, p['conv3_%d_1' % i], s)))
        nm = 'ST%s_%d_2' % (st, i)
        S = lambda t: conv3d_same(t, p[nm + '_S'], (1, 1, 1), p[nm + '_S_bias'])
        T = lambda t: conv3d_same(t, p[nm + '_T'], (1, 1, 1), p[nm + '_T_bias'])
        if st == 'A':
            out = torch.relu(self.gn(T(torch.relu(self.gn(S(out))))))
        elif st == 'B':
            a = torch.relu(self.gn(S(out)))
            out = torch.relu(self.gn(T(out))) + a
        else:
            a = torch.relu(self.gn(S(out)))
            out = a + torch.relu(self.gn(T(a)))
        out = self.gn(conv3d_same(out, p['conv3_%d_3' % i], (1, 1, 1)))
        res = x
        if first:
            res = self.gn(conv3d_same(x, p['dw3d_%d' % i], s))
        res = self.cbam(res, 'cbam_%d' % i)
        return torch.relu(out + res)

    def inference_p3d(self, x_ndhwc):
        p, b = self.p, self.base
        x = x_ndhwc.permute(0, 4, 1, 2, 3)
        x = torch.relu(self.gn(conv3d_same(x, p['firstconv1'], (1, 2, 2))))
        x = max_pool_same(x, (2, 3, 3), (2, 2, 2))
        i, inpl = 0, b
        pools = []
        for stage, (n, planes) in enumerate(zip(self.blocks, (b, 2 * b, 4 * b))):
            if stage == 2:      # deconv_pool3 and its GN are created before stage 3
                d3 = torch.relu(self.gn(conv3d_transpose_same(pools[1], p['deconv_pool3/kernel'], (2, 2, 2), p['deconv_pool3/bias'])))
            for j in range(n):
                x = self.block(x, i, inpl, planes, j == 0, stage > 0)
                inpl = planes * 4
                i += 1
            x = max_pool_same(x, (2, 1, 1), (2, 1, 1))
            pools.append(x)
        d4 = torch.relu(self.gn(conv3d_transpose_same(pools[2], p['deconv_pool4/kernel'], (4, 4, 4), p['deconv_pool4/bias'])))
        c = torch.cat([d3, d4, pools[0]], 1)
        c = torch.relu(self.gn(conv3d_same(c, p['conv_concat/kernel'], (1, 1, 1), p['conv_concat/bias'])))
        c = torch.relu(self.gn(conv3d_transpose_same(c, p['deconv_revise/kernel'], (2, 2, 2), p['deconv_revise/bias'])))
        c = conv3d_transpose_same(c, p['predict_revise/kernel'], (2, 2, 2), p['predict_revise/bias'])
        return c.permute(0, 2, 3, 4, 1)

    def decoder_block(self, x_ndhwc):
        """gn/p3d_gn.py:489-539 inference_p3d_decoder_block; parameter names WITHOUT the 'P3D/' scope prefix."""
        p, b = self.p, self.base
        x = x_ndhwc.permute(0, 4, 1, 2, 3)
        x = torch.relu(self.gn(conv3d_same(x, p['firstconv1'], (1, 2, 2))))
        x = max_pool_same(x, (2, 3, 3), (2, 2, 2))

        def up(t, name, s):
            return torch.relu(self.gn(conv3d_transpose_same(t, p[name + '/kernel'], s, p[name + '/bias'])))

        def conv(t, name):
            return torch.relu(self.gn(conv3d_same(t, p[name + '/kernel'], (1, 1, 1), p[name + '/bias'])))
        i, inpl = 0, b
        skips = []
        for stage, (n, planes) in enumerate(zip(self.blocks, (b, 2 * b, 4 * b))):
            if stage == 1:
                skips.append(up(x, 'deconv_pool2', (1, 1, 1)))
            if stage == 2:
                skips.append(up(x, 'deconv_pool3', (2, 2, 2)))
            for j in range(n):
                x = self.block(x, i, inpl, planes, j == 0, stage > 0)
                inpl = planes * 4
                i += 1
            x = max_pool_same(x, (2, 1, 1), (2, 1, 1))
        skips.append(up(x, 'deconv_pool4', (4, 4, 4)))
        c = conv(torch.cat(skips, 1), 'conv_concat')
        c = conv(c, 'decoder1_conv1')
        c = up(c, 'decoder1_deconv', (2, 2, 2))
        c = conv(c, 'decoder1_conv2')
        c = conv(c, 'decoder2_conv1')
        c = up(c, 'decoder2_deconv', (2, 2, 2))
        c = conv(c, 'decoder2_conv2')
        c = conv3d_same(c, p['results/kernel'], (1, 1, 1), p['results/bias'])
        return c.permute(0, 2, 3, 4, 1)
