"""Eager counterparts of the reference's graph functions (/root/reference/p3d.py): same names, same arguments,
numpy in / numpy out.  In the reference `p3d_unet(_X, _dropout, batch_size, training)` adds ops to the default
TF graph and a later `sess.run` executes them with the graph's variables; here the call runs the MI355X path at
once on a session cached per (function, input shape) -- the stand-in for the default graph -- whose variables
are reached with `session_for(...)` (`.load(checkpoint)` / `.save()`, TF variable names).  `_dropout` is a rate
(tf.layers.dropout), applied only when `training` is true, like the reference."""
import numpy as np

from .session import P3DSession

_sessions = {}


def session_for(structure, shape, device=0, seed=0):
    """The cached session ("graph + variables") behind `structure` for inputs of `shape` = [B,T,H,W,3]."""
    key = (structure, tuple(shape[:4]), device)
    if key not in _sessions:
        B, T, H, W = shape[:4]
        _sessions[key] = P3DSession(structure, batch=B, frames=T, height=H, width=W, device=device, seed=seed)
    return _sessions[key]


def reset():
    """tf.reset_default_graph(): drop every cached session."""
    for s in _sessions.values():
        s.close()
    _sessions.clear()


def _run(structure, _X, _dropout, batch_size, training, seed):
    x = np.ascontiguousarray(_X, dtype=np.float32)
    if x.ndim != 5 or x.shape[4] != 3:
        raise ValueError("_X must be [batch, frames, height, width, 3], got %s" % (x.shape,))
    if batch_size is not None and batch_size != x.shape[0]:
        raise ValueError("batch_size=%d but _X has %d clips" % (batch_size, x.shape[0]))
    return session_for(structure, x.shape).forward(x, dropout=float(_dropout), training=bool(training), seed=seed)


def p3d_unet(_X, _dropout, batch_size=2, training=True, seed=0):
    """p3d.py:169-221 -> saliency maps [B,T,H,W,1] in (0,1)."""
    return _run("unet", _X, _dropout, batch_size, training, seed)


def p3d_concat(_X, _dropout, batch_size=2, training=True, seed=0):
    """p3d.py:224-276 -> raw maps (no sigmoid)."""
    return _run("concat", _X, _dropout, batch_size, training, seed)


def p3d_unetplusplus_nonsa(_X, _dropout, batch_size=2, training=True, SA=False, seed=0):
    """p3d.py:401-459 (the SA argument is unused there too)."""
    return _run("unet++nonsa", _X, _dropout, batch_size, training, seed)


def p3d_unetplusplus_ds(_X, _dropout, batch_size=2, training=True, SA=False, seed=0):
    """p3d.py:340-397, self attention on x_4_0, x_3_1, x_2_2, x_1_3."""
    return _run("unet++ds", _X, _dropout, batch_size, training, seed)


def p3d_unetplusplus(_X, _dropout, batch_size=2, training=True, SA=False, seed=0):
    """p3d.py:280-338 cannot be built in the reference either: its last attention call (p3d.py:334, sub_size=4) adds
    a quarter-resolution tensor to a full-resolution one (utils/network.py:187-192).  Named, not silently replaced."""
    raise NotImplementedError("p3d_unetplusplus adds tensors of different shapes (p3d.py:334); use p3d_unetplusplus_ds")
