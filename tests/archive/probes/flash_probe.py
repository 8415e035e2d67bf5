"""Diagnostic: gradient differences between the two executions of the attention core, per tensor (run on the GPU box)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from oracle import p3d
from tests.test_gpu_net import randomise_norm_params, make_session

for shape in [(2, 16, 64, 64), (1, 16, 96, 64)]:
    st, cfg = 'unet++ds', p3d.NetConfig(base=16, blocks=(1, 1, 1))
    p32 = {k: v.astype(np.float32) for k, v in randomise_norm_params(p3d.init_params(1, st, cfg, dtype=np.float64)).items()}
    x = p3d.synthetic_clip(0, shape + (3,))
    y = p3d.synthetic_target(3, shape)
    s = make_session(cfg, shape, p32, st)
    out = {}
    for mode in ("gemm", "flash", "gemm"):
        s.set_attention_mode(mode)
        loss, pred = s.backward(x, y, 0.0)
        grads = {n: s.get_grad(n) for n, _, tr in s.variables() if tr}
        taps = {t: s.activation(t) for t in ['x_4_0_sa', 'x_3_1_sa', 'x_2_2_sa', 'x_1_3_sa']}
        out.setdefault(mode, []).append((loss, grads, taps))
    g0, g1, g2 = out["gemm"][0][1], out["flash"][0][1], out["gemm"][1][1]
    print(shape, "losses", out["gemm"][0][0], out["flash"][0][0])
    for t in out["gemm"][0][2]:
        a, b = out["gemm"][0][2][t], out["flash"][0][2][t]
        print("  tap", t, np.abs(a - b).max() / max(np.abs(a).max(), 1))
    scale = np.median([np.linalg.norm(v) for v in g0.values()])
    rel = {n: float(np.linalg.norm(g0[n] - g1[n]) / max(np.linalg.norm(g0[n]), 1e-2 * scale)) for n in g0}
    rep = max(float(np.linalg.norm(g0[n] - g2[n])) for n in g0)
    print("  gemm vs gemm again (max abs norm diff):", rep)
    for n in list(g0):
        if 'sa' in n or 'gamma' in n[:5] or rel[n] > 1e-3:
            print("   %-40s %.3e  |g|=%.3e" % (n, rel[n], np.linalg.norm(g0[n])))
    print("  median", np.median(list(rel.values())))
    s.close()
