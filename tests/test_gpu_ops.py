"""Op-level parity of the HIP kernels (through the C ABI's p3d_op_* entry points) against the
oracle in float64.  Tolerance: 2e-5 of the result's max magnitude (fp32 sums of up to ~7k terms;
the north-star tolerance for the path is 1e-3 relative)."""
import numpy as np
import pytest

from oracle import nn

pytestmark = pytest.mark.gpu

TOL = 2e-5


def close(got, want, tol=TOL):
    scale = max(np.abs(want).max(), 1e-30)
    err = np.abs(got.astype(np.float64) - want).max() / scale
    assert err < tol, err


def rnd(rng, shape):
    return rng.standard_normal(shape).astype(np.float32)


# (xshape, kernel, cout, strides): the shapes of the path at reduced extent + ragged / edge cases
CONV_CASES = [
    ((2, 4, 12, 12, 64), (1, 1, 1), 64, (1, 1, 1)),      # conv3_i_1
    ((2, 4, 12, 12, 64), (1, 1, 1), 256, (1, 1, 1)),     # conv3_i_3
    ((2, 4, 14, 14, 256), (1, 1, 1), 128, (1, 2, 2)),    # strided 1x1x1 (ids 3, 11)
    ((1, 4, 13, 11, 32), (1, 1, 1), 48, (1, 2, 2)),      # odd extents, ragged tiles
    ((2, 4, 12, 12, 64), (1, 3, 3), 64, (1, 1, 1)),      # convS
    ((2, 4, 12, 12, 64), (3, 1, 1), 64, (1, 1, 1)),      # convT
    ((2, 2, 7, 7, 256), (1, 3, 3), 256, (1, 1, 1)),      # layer-3 convS
    ((2, 2, 7, 7, 256), (3, 1, 1), 256, (1, 1, 1)),      # layer-3 convT
    ((1, 3, 9, 10, 20), (3, 3, 3), 12, (1, 1, 1)),       # general 3x3x3, K not multiple of 32
    ((1, 4, 10, 10, 16), (3, 3, 3), 24, (2, 2, 2)),      # strided 3x3x3 (deconv input-gradient shape)
    ((1, 3, 37, 41, 3), (1, 7, 7), 64, (1, 2, 2)),       # stem (Cin=3), odd extents
    ((2, 4, 32, 32, 3), (1, 7, 7), 16, (1, 2, 2)),       # stem, reduced width
    ((1, 1, 5, 5, 8), (1, 1, 1), 4, (1, 1, 1)),          # tiny
    ((1, 4, 12, 12, 8), (3, 3, 3), 4, (1, 1, 1)),        # decoder2_conv2 of the GN decoder head at base 16: 8 -> 4 channels
    ((1, 4, 12, 12, 32), (3, 3, 3), 8, (1, 1, 1)),       # decoder2_conv1
    ((2, 3, 6, 6, 4), (3, 3, 3), 8, (1, 1, 1)),          # 4 input channels: one 16-byte chunk per row
    ((1, 4, 10, 10, 8), (3, 3, 3), 8, (2, 2, 2)),        # input-gradient shape of decoder2_deconv
    ((1, 4, 8, 8, 64), (3, 3, 3), 128, (1, 1, 1)),       # filter gradient [64 x 128] per tap: 64x128 tile
    ((1, 4, 8, 8, 128), (3, 3, 3), 64, (1, 1, 1)),       # [128 x 64]: 128x64 tile
    ((2, 16, 32, 32, 64), (1, 1, 1), 32, (1, 1, 1)),     # one filter-gradient tile over 32768 positions: up to 256 cuts
]


@pytest.mark.parametrize("xs,k,co,s", CONV_CASES)
def test_conv3d_forward_and_grads(xs, k, co, s):
    from sap3d_tensorflow_amd import ops
    rng = np.random.default_rng(abs(hash((xs, k, co, s))) % (2 ** 31))
    x = rnd(rng, xs)
    w = rnd(rng, k + (xs[4], co)) * 0.1
    b = rnd(rng, (co,))
    want = nn.conv3d_forward(x.astype(np.float64), w.astype(np.float64), s) + b
    got = ops.conv3d(x, w, s, bias=b)
    assert got.shape == want.shape
    close(got, want)
    dy = rnd(rng, want.shape)
    want_dw = nn.conv3d_backward_filter(x.astype(np.float64), dy.astype(np.float64), w.shape, s)
    got_dw, got_db = ops.conv3d_backprop_filter(x, w.shape, dy, s, with_bias=True)
    close(got_dw, want_dw)
    close(got_db, dy.astype(np.float64).reshape(-1, co).sum(0))
    if xs[4] % 4 == 0:
        want_dx = nn.conv3d_backward_input(dy.astype(np.float64), w.astype(np.float64), s, xs)
        got_dx = ops.conv3d_backprop_input(xs, w, dy, s)
        close(got_dx, want_dx)


STEM_CASES = [
    (2, 4, 32, 32),        # 16 x 16 outputs: 8 pairs per output row
    (1, 2, 30, 28),        # 15 x 14: rows above AND below the image in the last / first patches, runs that end mid-row
    (1, 3, 37, 44),        # odd height: SAME padding 3 + 3
    (3, 5, 18, 12),        # fewer pairs than one block's waves want: empty waves
    (2, 16, 112, 112),     # the reference clip size, two clips: 256 blocks of 7 chunks, 14 pairs per wave and chunk (the unrolled variant)
    (1, 3, 224, 224),      # 224-pixel clips: 28 pairs per wave and chunk, six staging loads per thread
    (1, 2, 64, 200),       # wide rows on the generic variant
]


@pytest.mark.parametrize("shape", STEM_CASES)
def test_stem_filter_gradient_in_one_pass(shape):
    """firstconv1's filter gradient ([1,7,7,3,64], stride [1,2,2], even output width) on stem_wgrad.hip: all seven kernel rows in
    one pass over the output gradient, operands straight into the MFMA layout (csrc/stem_wgrad.hip); p3d.py:172."""
    from sap3d_tensorflow_amd import ops
    rng = np.random.default_rng(sum(shape))
    xs = shape + (3,)
    x = rnd(rng, xs)
    ws = (1, 7, 7, 3, 64)
    s = (1, 2, 2)
    oshape = (xs[0], xs[1], (xs[2] + 1) // 2, (xs[3] + 1) // 2, 64)
    dy = rnd(rng, oshape)
    want = nn.conv3d_backward_filter(x.astype(np.float64), dy.astype(np.float64), ws, s)
    got = ops.conv3d_backprop_filter(x, ws, dy, s, with_bias=False)
    close(got, want)
    again = ops.conv3d_backprop_filter(x, ws, dy, s, with_bias=False)
    assert np.array_equal(got, again)          # fixed summation order


DECONV_CASES = [
    ((2, 1, 7, 7, 64), (1, 3, 3), 32, (2, 2, 2)),     # deconv1: k1 on the temporal axis
    ((1, 2, 6, 6, 64), (2, 3, 3), 16, (2, 2, 2)),     # deconv2: k2
    ((1, 2, 6, 5, 32), (3, 3, 3), 16, (2, 2, 2)),     # deconv3
    ((1, 1, 3, 3, 16), (3, 3, 3), 8, (4, 4, 4)),      # deconv_pool4 of the concat head: k < s
    ((1, 2, 5, 5, 16), (3, 3, 3), 8, (1, 1, 1)),      # deconv_pool2 of the concat head: stride 1
    ((1, 1, 3, 3, 32), (1, 3, 3), 16, (4, 4, 4)),     # deconv_pool4 of the GN decoder head: k < s on every axis
    ((1, 2, 6, 6, 8), (3, 3, 3), 8, (2, 2, 2)),       # decoder2_deconv at base 16
]


@pytest.mark.parametrize("xs,k,co,s", DECONV_CASES)
def test_conv3d_transpose(xs, k, co, s):
    from sap3d_tensorflow_amd import ops
    rng = np.random.default_rng(abs(hash((xs, k, co, s))) % (2 ** 31))
    x = rnd(rng, xs)
    kern = rnd(rng, k + (co, xs[4])) * 0.1
    b = rnd(rng, (co,))
    t = nn.Tape()
    want = nn.conv3d_transpose(t, nn.Var(x.astype(np.float64)), nn.Var(kern.astype(np.float64)), s,
                               nn.Var(b.astype(np.float64))).data
    got = ops.conv3d_transpose(x, kern, s, bias=b)
    assert got.shape == want.shape
    close(got, want)


POOL_CASES = [
    ((2, 4, 12, 12, 16), (2, 3, 3), (2, 2, 2)),     # pool1
    ((2, 4, 6, 6, 32), (2, 1, 1), (2, 1, 1)),       # temporal pools
    ((1, 3, 7, 9, 8), (2, 3, 3), (2, 2, 2)),        # odd extents: SAME padding on every axis
    # border cells of the gather backward (a fault at 0x1000 while its two-window fast path was written -- the
    # candidate windows of a border cell lie at output index -1 and at index Do, and the first version formed their table
    # and gradient addresses before looking at the validity flag).  k = 4, s = 2 on 8 cells: pad 1 before / 1 after, so cell 0
    # sees window -1 and cell 7 sees window 4 = Do on every axis, on the fast path (k <= 2s); k = 5 takes the general loops.
    ((1, 8, 8, 8, 8), (4, 4, 4), (2, 2, 2)),
    ((1, 8, 8, 8, 4), (5, 5, 5), (2, 2, 2)),
]


@pytest.mark.parametrize("xs,k,s", POOL_CASES)
def test_max_pool3d(xs, k, s):
    from sap3d_tensorflow_amd import ops
    rng = np.random.default_rng(7)
    x = np.maximum(rnd(rng, xs), 0)       # post-ReLU input like the path: many tied zeros
    t = nn.Tape()
    X = nn.Var(x.astype(np.float64))
    Y = nn.max_pool3d(t, X, k, s)
    got = ops.max_pool3d(x, k, s)
    assert np.array_equal(got, Y.data.astype(np.float32))
    dy = rnd(rng, Y.data.shape)
    Y.grad = dy.astype(np.float64)
    t.ops[-1]()
    got_dx = ops.max_pool3d_grad(x, k, s, dy)
    close(got_dx, X.grad)


@pytest.mark.parametrize("rows,c", [(1, 4), (17, 64), (5000, 64), (40000, 128), (3001, 1024), (777, 6), (50, 2048), (0, 8)])
def test_bias_add_grad(rows, c):
    """BiasAddGrad of the conv / transposed-conv bias (p3d.py:147-150): column sums, fixed order -> bit-identical runs.
    Shapes cover the float4 path (C % 4 == 0, C <= 1024), the scalar one (C = 6, 2048), one row, and no rows."""
    from sap3d_tensorflow_amd import ops
    rng = np.random.default_rng(rows * 131 + c)
    dy = rnd(rng, (rows, c))
    got = ops.bias_add_grad(dy)
    want = dy.astype(np.float64).sum(0)
    scale = np.abs(dy.astype(np.float64)).sum(0).max() if rows else 1.0
    assert np.abs(got - want).max() <= 2e-6 * max(scale, 1.0)
    assert np.array_equal(got, ops.bias_add_grad(dy))


@pytest.mark.parametrize("tile", ["64x64", "64x128", "128x64", "128x128"])
@pytest.mark.parametrize("xs,k,co", [((2, 4, 14, 14, 256), (1, 3, 3), 128), ((1, 4, 8, 8, 128), (3, 3, 3), 192), ((3, 2, 9, 7, 136), (1, 1, 1), 132)])
def test_filter_gradient_tile_shapes(tile, xs, k, co):
    """Every tile shape of the filter-gradient kernel (conv_wgrad2.hip) on the same problems -- ragged K / Nc tails included
    (136 x 132) -- against the oracle, and bit-identical run to run (fixed cut order)."""
    from sap3d_tensorflow_amd import lib, ops
    tm, tn = (int(v) for v in tile.split("x"))
    lib().p3d_debug_force_plan(-1, 0, tm, tn)
    try:
        _filter_gradient_case(ops, xs, k, co)
    finally:
        lib().p3d_debug_force_plan(-1, 0, 0, 0)


@pytest.mark.parametrize("tile,xs,co", [(1, (1, 8, 56, 112, 32), 64), (0, (1, 4, 56, 112, 32), 64), (2, (1, 8, 56, 112, 32), 128),
                                         (1, (2, 4, 56, 100, 16), 64)])
def test_conv_last_round_more_than_half_full(tile, xs, co):
    """392 (or 350) tiles = 1.53 rounds: the 136 tiles of the last round are cut into several K-slices that themselves take
    more than one round of blocks (conv_igemm2.hip, tail_slices)."""
    from sap3d_tensorflow_amd import lib, ops
    k, s = (3, 3, 3), (1, 1, 1)
    rng = np.random.default_rng(23)
    x = rnd(rng, xs)
    w = rnd(rng, k + (xs[4], co)) * 0.1
    b = rnd(rng, (co,))
    want = nn.conv3d_forward(x.astype(np.float64), w.astype(np.float64), s) + b
    dy = rnd(rng, want.shape)
    want_dx = nn.conv3d_backward_input(dy.astype(np.float64), w.astype(np.float64), s, xs)
    lib().p3d_debug_force_plan(tile, 0, 0, 0)
    try:
        got = ops.conv3d(x, w, s, bias=b)
        got_dx = ops.conv3d_backprop_input(xs, w, dy, s)
    finally:
        lib().p3d_debug_force_plan(-1, 0, 0, 0)
    close(got, want)
    close(got_dx, want_dx)


@pytest.mark.parametrize("tile", [0, 1, 2])
def test_conv_last_round_is_k_sliced(tile):
    """A launch whose tile count is a little over a multiple of the 256 CUs: the tiles of the last round go out as a K-sliced
    class of the same launch (conv_igemm2.hip, p3d_igemm2_tail_split).  33792 output rows = 528 / 264 / 264 tiles of
    64x64 / 128x64 / 128x128; forward and input gradient against the oracle."""
    from sap3d_tensorflow_amd import lib, ops
    xs, k, co, s = (1, 8, 66, 64, 32), (3, 3, 3), 128, (1, 1, 1)
    rng = np.random.default_rng(17)
    x = rnd(rng, xs)
    w = rnd(rng, k + (xs[4], co)) * 0.1
    b = rnd(rng, (co,))
    want = nn.conv3d_forward(x.astype(np.float64), w.astype(np.float64), s) + b
    dy = rnd(rng, want.shape)
    want_dx = nn.conv3d_backward_input(dy.astype(np.float64), w.astype(np.float64), s, xs)
    lib().p3d_debug_force_plan(tile, 0, 0, 0)
    try:
        got = ops.conv3d(x, w, s, bias=b)
        got_dx = ops.conv3d_backprop_input(xs, w, dy, s)
        again = ops.conv3d(x, w, s, bias=b)
    finally:
        lib().p3d_debug_force_plan(-1, 0, 0, 0)
    close(got, want)
    close(got_dx, want_dx)
    assert np.array_equal(got, again)


def _filter_gradient_case(ops, xs, k, co):
    rng = np.random.default_rng(5)
    x = rnd(rng, xs)
    s = (1, 1, 1)
    dy = rnd(rng, xs[:4] + (co,))
    want_dw = nn.conv3d_backward_filter(x.astype(np.float64), dy.astype(np.float64), k + (xs[4], co), s)
    got_dw, got_db = ops.conv3d_backprop_filter(x, k + (xs[4], co), dy, s, with_bias=True)
    close(got_dw, want_dw)
    close(got_db, dy.astype(np.float64).reshape(-1, co).sum(0))
    again, _ = ops.conv3d_backprop_filter(x, k + (xs[4], co), dy, s, with_bias=True)
    assert np.array_equal(got_dw, again)


ATTN_CASES = [
    (2, 300, 77, 32),        # ragged query and key tiles
    (1, 129, 33, 64),        # one row past a tile on both sides
    (2, 128, 64, 128),       # exact tiles
    (1, 50, 200, 256),       # widest instantiation, more keys than queries
    (3, 5, 3, 32),           # smaller than any tile
]


@pytest.mark.parametrize("B,ng,nf,ch", ATTN_CASES)
def test_attention_core_forward_and_grads(B, ng, nf, ch):
    """softmax(g f^T) h (utils/network.py:183-185) on the kernels that keep the scores on chip, against float64 numpy:
    forward, and dg / df / dh for a random gradient of o.  Scores of a few units, so the softmax is neither flat nor one-hot."""
    from sap3d_tensorflow_amd import ops
    rng = np.random.default_rng(ch * 1000 + ng)
    ci = ch // 8
    g = rnd(rng, (B, ng, ci)) * (2.0 / np.sqrt(ci)) ** 0.5
    f = rnd(rng, (B, nf, ci)) * (2.0 / np.sqrt(ci)) ** 0.5 * 2
    h = rnd(rng, (B, nf, ch))
    d_o = rnd(rng, (B, ng, ch))
    g64, f64, h64, d64 = (a.astype(np.float64) for a in (g, f, h, d_o))
    s = g64 @ f64.transpose(0, 2, 1)
    p = np.exp(s - s.max(-1, keepdims=True))
    p /= p.sum(-1, keepdims=True)
    want = p @ h64
    dp = d64 @ h64.transpose(0, 2, 1)
    ds = p * (dp - (dp * p).sum(-1, keepdims=True))
    want_dg, want_df, want_dh = ds @ f64, ds.transpose(0, 2, 1) @ g64, p.transpose(0, 2, 1) @ d64
    assert 0.02 < p.max(-1).mean() < 0.9
    got = ops.attention_core(g, f, h)
    close(got, want)
    o, dg, df, dh = ops.attention_core(g, f, h, d_o)
    assert np.array_equal(o, got)
    close(dh, want_dh)
    close(dg, want_dg, 1e-4)         # ds = p (dp - <p, dp>) cancels: an fp32 rounding of dp is 1e-7 of |dp|, not of the difference
    close(df, want_df, 1e-4)
    again = ops.attention_core(g, f, h, d_o)
    assert all(np.array_equal(a, b) for a, b in zip((o, dg, df, dh), again))


@pytest.mark.parametrize("xs,ci,co", [((8, 8, 28, 28), 64, 256), ((8, 8, 28, 28), 64, 64), ((2, 16, 32, 32), 64, 128),
                                      ((8, 8, 28, 28), 256, 64), ((2, 8, 32, 33), 64, 256)])
def test_dense_pointwise_conv_at_stage1_size(xs, ci, co):
    """1x1x1 convs over >= 16384 rows (conv1 / conv3 / the projection of stage 1 at 8 clips of 16x112x112: 50176 rows), forward with
    bias and input gradient against the oracle.  From 64 input channels the forward runs on the weights-in-registers streaming kernel
    (conv_pointwise.hip), to 64 channels the input gradient does (the last case: 16896 rows = 528 slabs of 32 on 512 blocks); the others on
    the tiled kernel -- same contract either way."""
    from sap3d_tensorflow_amd import ops
    xs = xs + (ci,)
    rng = np.random.default_rng(3)
    x = rnd(rng, xs)
    w = rnd(rng, (1, 1, 1, ci, co)) * 0.2
    b = rnd(rng, (co,))
    want = nn.conv3d_forward(x.astype(np.float64), w.astype(np.float64), (1, 1, 1)) + b.astype(np.float64)
    got = ops.conv3d(x, w, (1, 1, 1), bias=b)
    close(got, want)
    assert np.array_equal(got, ops.conv3d(x, w, (1, 1, 1), bias=b))
    dy = rnd(rng, want.shape)
    close(ops.conv3d_backprop_input(xs, w, dy, (1, 1, 1)), nn.conv3d_backward_input(dy.astype(np.float64), w.astype(np.float64), (1, 1, 1), xs))


@pytest.mark.parametrize("k,ci,co", [((1, 1, 1), 64, 256), ((1, 3, 3), 64, 64), ((3, 1, 1), 64, 64), ((1, 1, 1), 256, 64)])
def test_filter_gradients_at_stage1_size(k, ci, co):
    """Filter (and bias) gradients over 50176 positions (stage 1 at 8 clips of 16x112x112): cut counts, block order and cut
    folds of the real step's launches, against the oracle."""
    from sap3d_tensorflow_amd import ops
    xs = (8, 8, 28, 28, ci)
    rng = np.random.default_rng(11 + ci + co + k[1])
    x = rnd(rng, xs)
    dy = rnd(rng, xs[:4] + (co,))
    want = nn.conv3d_backward_filter(x.astype(np.float64), dy.astype(np.float64), k + (ci, co), (1, 1, 1))
    got, db = ops.conv3d_backprop_filter(x, k + (ci, co), dy, (1, 1, 1), with_bias=True)
    close(got, want)
    close(db, dy.astype(np.float64).reshape(-1, co).sum(0))
    again, _ = ops.conv3d_backprop_filter(x, k + (ci, co), dy, (1, 1, 1), with_bias=True)
    assert np.array_equal(got, again)


def test_conv3d_transpose_at_deconv3_size():
    """conv3d_transpose [3,3,3] stride 2 over the lattice of deconv3 at 8 clips of 16x112x112 (8 x 8x28x28 -> 16x56x56, 128 output
    channels; 32 input channels instead of 256 so that the float64 oracle finishes in seconds): the eight stride residue classes
    as one grouped launch of several rounds, against the oracle."""
    from sap3d_tensorflow_amd import ops
    xs, k, co, s = (8, 8, 28, 28, 32), (3, 3, 3), 128, (2, 2, 2)
    rng = np.random.default_rng(41)
    x = rnd(rng, xs)
    kern = rnd(rng, k + (co, xs[4])) * 0.1
    b = rnd(rng, (co,))
    t = nn.Tape()
    want = nn.conv3d_transpose(t, nn.Var(x.astype(np.float64)), nn.Var(kern.astype(np.float64)), s, nn.Var(b.astype(np.float64))).data
    got = ops.conv3d_transpose(x, kern, s, bias=b)
    assert got.shape == want.shape
    close(got, want)
    assert np.array_equal(got, ops.conv3d_transpose(x, kern, s, bias=b))


# ---- the head of inference_p3d (gn/p3d_gn.py:234-257) at FULL width: conv_concat is 311 of the net's 382 GFLOP per clip --------
# The longest reduction of these launches is K = 27 x 1792 = 48384 terms (the bottleneck convs: <= 2304), so the tolerance is
# stated per case: an fp32 sum of n terms of magnitude ~s sits ~eps*sqrt(n)*s from float64 whatever its order.
GN_HEAD_CONVS = [
    # (x shape, kernel, Cout, strides, tolerance)
    ((1, 4, 28, 28, 1792), (3, 3, 3), 1024, (1, 1, 1), 2e-5),     # conv_concat on the concatenator of ONE clip
    ((1, 2, 10, 10, 1792), (3, 3, 3), 1024, (1, 1, 1), 2e-5),     # ... on a lattice where most positions touch the SAME padding
]


@pytest.mark.parametrize("xs,k,co,s,tol", GN_HEAD_CONVS)
def test_gn_head_conv_concat_at_full_width(xs, k, co, s, tol):
    """conv_concat (gn/p3d_gn.py:253: 3x3x3, 1792 -> 1024 on 4x28x28 per clip): forward, input gradient, filter gradient and
    bias gradient through the C ABI's op entry points against the float64 oracle."""
    from sap3d_tensorflow_amd import ops
    rng = np.random.default_rng(1792)
    x = rnd(rng, xs)
    w = rnd(rng, k + (xs[4], co)) * 0.02
    b = rnd(rng, (co,))
    x64, w64 = x.astype(np.float64), w.astype(np.float64)
    want = nn.conv3d_forward(x64, w64, s) + b
    got = ops.conv3d(x, w, s, bias=b)
    close(got, want, tol)
    assert np.array_equal(got, ops.conv3d(x, w, s, bias=b))
    dy = rnd(rng, want.shape)
    dy64 = dy.astype(np.float64)
    close(ops.conv3d_backprop_input(xs, w, dy, s), nn.conv3d_backward_input(dy64, w64, s, xs), tol)
    got_dw, got_db = ops.conv3d_backprop_filter(x, w.shape, dy, s, with_bias=True)
    close(got_dw, nn.conv3d_backward_filter(x64, dy64, w.shape, s), tol)
    close(got_db, dy64.reshape(-1, co).sum(0), tol)


GN_HEAD_DECONVS = [
    ((1, 1, 7, 7, 1024), (3, 3, 3), 1024, (4, 4, 4)),     # deconv_pool4 (gn/p3d_gn.py:245): kernel 3, stride 4, 1024 -> 1024
    ((1, 2, 14, 14, 512), (3, 3, 3), 512, (2, 2, 2)),     # deconv_pool3 (:239): 512 -> 512
    ((1, 4, 28, 28, 1024), (3, 3, 3), 256, (2, 2, 2)),    # deconv_revise (:255): 1024 -> 256 onto 8x56x56
]


@pytest.mark.parametrize("xs,k,co,s", GN_HEAD_DECONVS)
def test_gn_head_transposed_convs_at_full_width(xs, k, co, s):
    """The transposed convs of the inference_p3d head at the reference widths, forward (tf.layers.conv3d_transpose) and both
    gradients: the input gradient of a transposed conv is the forward conv with the same kernel read [.., Cout, Cin], its filter
    gradient the conv filter gradient with the roles of input and output gradient swapped (net_ops.inc, deconv())."""
    from sap3d_tensorflow_amd import ops
    rng = np.random.default_rng(co + xs[4])
    x = rnd(rng, xs)
    kern = rnd(rng, k + (co, xs[4])) * 0.03
    b = rnd(rng, (co,))
    t = nn.Tape()
    X, K, Bv = nn.Var(x.astype(np.float64)), nn.Var(kern.astype(np.float64)), nn.Var(b.astype(np.float64))
    Y = nn.conv3d_transpose(t, X, K, s, Bv)
    got = ops.conv3d_transpose(x, kern, s, bias=b)
    assert got.shape == Y.data.shape
    close(got, Y.data)
    dy = rnd(rng, Y.data.shape)
    Y.grad = dy.astype(np.float64)
    for fn in reversed(t.ops):
        fn()
    # y = conv_transpose(x, kern)  <=>  x' = conv(y', w = kern) is its adjoint: dx = conv3d(dy, kern, s), dkern = filter gradient of
    # that conv with (input = dy, output gradient = x)
    close(ops.conv3d(dy, kern, s), X.grad)
    close(ops.conv3d_backprop_filter(dy, kern.shape, x, s), K.grad)
    close(ops.bias_add_grad(dy), Bv.grad)


@pytest.mark.parametrize("shape", [(1, 2, 30, 28), (2, 16, 112, 112), (1, 2, 224, 224)])
@pytest.mark.parametrize("batch", [1, 0])
def test_stem_filter_gradient_through_batchnorm_on_the_operand_path(shape, batch):
    """stem_wgrad.hip's FUSED form (what the train step runs: firstconv1's filter gradient with the stem BatchNorm's backward apply
    pass evaluated on the kernel's B operand, p3d.py:172-174) against the two-launch form -- bn_bwd_apply_kernel<0>, then the plain
    one-pass filter gradient -- on the same inputs, AND against the float64 oracle of the two steps; batch statistics (the
    coefficient terms) and moving statistics (c1 = c2 = 0); the generic variant (15 x 14 outputs) and the tiled ones (112, 224)."""
    import ctypes as C
    from sap3d_tensorflow_amd._lib import check, fptr, lib
    rng = np.random.default_rng(sum(shape) + batch)
    xs = shape + (3,)
    x = rnd(rng, xs)
    oshape = (xs[0], xs[1], (xs[2] + 1) // 2, (xs[3] + 1) // 2, 64)
    y = rnd(rng, oshape)
    dz = rnd(rng, oshape)
    gamma = rng.uniform(0.5, 1.5, 64).astype(np.float32)
    beta = rng.uniform(-0.3, 0.3, 64).astype(np.float32)
    y64 = y.astype(np.float64).reshape(-1, 64)
    mean = y64.mean(0) if batch else rng.uniform(-0.1, 0.1, 64)
    var = y64.var(0) if batch else rng.uniform(0.5, 1.5, 64)
    invstd = 1.0 / np.sqrt(var + 1e-3)
    scale = gamma * invstd
    shift = beta - mean * scale
    xhat = (y64 - mean) * invstd
    g = np.where(scale * y64 + shift > 0, dz.astype(np.float64).reshape(-1, 64), 0.0)
    M = y64.shape[0]
    c1, c2 = g.sum(0) / M, (g * xhat).sum(0) / M
    dy = gamma * invstd * ((g - c1 - xhat * c2) if batch else g)
    want = nn.conv3d_backward_filter(x.astype(np.float64), dy.reshape(oshape), (1, 7, 7, 3, 64), (1, 2, 2))
    tab = np.stack([scale, shift, mean, invstd, gamma]).astype(np.float32)
    coef = np.stack([c1, c2], axis=1).astype(np.float32)
    fused = np.empty((1, 7, 7, 3, 64), np.float32)
    two = np.empty_like(fused)
    check(lib().p3d_debug_stem_wgrad_through_bn(0, fptr(x), (C.c_int64 * 5)(*xs), fptr(y), fptr(dz), fptr(np.ascontiguousarray(tab)),
                                               fptr(np.ascontiguousarray(coef)), batch, fptr(fused), fptr(two)))
    # the gate uses fp32 scale*y+shift on the GPU: an element within 1e-7 of zero may flip against float64; one element of ~1.6 M moves the
    # sums by ~1e-6 of their magnitude, well inside the tolerance
    close(two, want, 5e-5)
    close(fused, want, 5e-5)
    close(fused, two.astype(np.float64), 2e-6)
