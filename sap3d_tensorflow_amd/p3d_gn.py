"""Eager counterparts of the reference's GroupNorm + CBAM graph functions (/root/reference/gn/p3d_gn.py); see p3d.py
in this package for the conventions."""
from .p3d import _run


def inference_p3d(_X, _dropout, batch_size=2, training=True, seed=0):
    """gn/p3d_gn.py:214-258 (net='P3D') -> raw maps."""
    return _run("gn_p3d", _X, _dropout, batch_size, training, seed)


def inference_p3d_concat(_X, _dropout, batch_size=2, training=True, seed=0):
    """gn/p3d_gn.py:279-324 (net='P3D_CONCAT')."""
    return _run("gn_p3d_concat", _X, _dropout, batch_size, training, seed)


def inference_p3d_decoder_block(_X, _dropout, batch_size=2, training=True, seed=0):
    """gn/p3d_gn.py:489-539 (net='P3D_DECODER'; variables named 'P3D/...')."""
    return _run("gn_p3d_decoder", _X, _dropout, batch_size, training, seed)
