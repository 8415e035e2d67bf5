"""The RCCL gradient all-reduce path of libp3dhip on ONE rank (a one-rank communicator reduces to the
identity): bucketed launches on the comm stream, the joins with the main and side streams, and torch being
loaded in the same process (bench.py imports torch.distributed when N > 1)."""
import os

import numpy as np
import pytest

from oracle import p3d

pytestmark = pytest.mark.gpu


def test_single_rank_allreduce_is_identity():
    import torch  # noqa: F401  (same process as libp3dhip, like bench.py at N > 1)
    from sap3d_tensorflow_amd import P3DSession
    cfg = p3d.NetConfig(base=16, blocks=(2, 2, 3))
    shape = (2, 16, 48, 48)
    params = p3d.init_params(1, 'unet', cfg)
    x = p3d.synthetic_clip(0, shape + (3,))
    y = p3d.synthetic_target(3, shape)

    def run(with_comm):
        s = P3DSession('unet', batch=shape[0], frames=shape[1], height=shape[2], width=shape[3], base=cfg.base,
                       blocks=cfg.blocks)
        s.load(params)
        if with_comm:
            os.environ["P3D_BUCKET_MB"] = "1"          # many small buckets
            s.comm_init(P3DSession.comm_unique_id())
        losses = [s.train_step(x, y, dropout=0.0) for _ in range(3)]
        w = s.get_param('conv3d_transpose_2/kernel')
        s.close()
        return losses, w

    l0, w0 = run(False)
    l1, w1 = run(True)
    assert np.allclose(l0, l1, rtol=2e-4)
    assert np.abs(w0 - w1).max() < 1e-3          # Adam turns atomics-order noise into +-lr noise
