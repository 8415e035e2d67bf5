"""Host-side mirror of the reference's session contract (train.py:175-218, gen_pred.py:48-64,151):
a P3DSession owns one libp3dhip handle = one built graph with its variables resident in HBM."""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import P3dConfig, P3dError, P3dOpTime, check, fptr, lib

STRUCTURES = {"unet": 0, "concat": 1, "gn_p3d": 2,     # train.py:149-154 --structure
              "unet++nonsa": 3,                          # p3d.py:401 p3d_unetplusplus_nonsa
              "gn_p3d_decoder": 4,                       # gn/p3d_gn.py:489 inference_p3d_decoder_block (net='P3D_DECODER')
              "gn_p3d_concat": 5,                        # gn/p3d_gn.py:279 inference_p3d_concat (net='P3D_CONCAT')
              "unet++ds": 6}                             # p3d.py:340 p3d_unetplusplus_ds (unet++ with self attention)


class P3DSession:
    """sess = P3DSession(batch=2)  ~  building the graph + tf.Session() in train.py:143-201."""

    def __init__(self, structure="unet", batch=2, frames=16, height=112, width=112, base=64, blocks=(3, 8, 36),
                 device=0, world_size=1, rank=0, seed=None):
        if structure not in STRUCTURES:
            raise ValueError("unknown structure %r (have %s)" % (structure, sorted(STRUCTURES)))
        self.cfg = P3dConfig()
        lib().p3d_default_config(C.byref(self.cfg))
        self.cfg.structure = STRUCTURES[structure]
        self.cfg.batch, self.cfg.frames, self.cfg.height, self.cfg.width = batch, frames, height, width
        self.cfg.base = base
        for i in range(3):
            self.cfg.blocks[i] = blocks[i]
        self.cfg.device, self.cfg.world_size, self.cfg.rank = device, world_size, rank
        self._h = C.c_void_p()
        check(lib().p3d_create(C.byref(self.cfg), C.byref(self._h)))
        _lib.register_session(self)
        self.x_shape = (batch, frames, height, width, 3)
        self.y_shape = (batch, frames, height, width)
        self.pred_shape = (batch, frames, height, width, 1)
        self._info = None
        if seed is not None:
            self.init_params(seed)

    def close(self):
        if self._h:
            lib().p3d_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- variables (tf.global_variables / Saver) ----------------------------------------------
    def variables(self):
        """[(name, shape, trainable)] in creation order."""
        if self._info is None:
            out = []
            n = lib().p3d_num_params(self._h)
            for i in range(n):
                name = C.c_char_p()
                nd = C.c_int()
                shape = (C.c_int64 * 5)()
                tr = C.c_int()
                check(lib().p3d_param_info(self._h, i, C.byref(name), C.byref(nd), shape, C.byref(tr)))
                out.append((name.value.decode(), tuple(shape[:nd.value]), bool(tr.value)))
            self._info = out
        return self._info

    def init_params(self, seed=0):
        """tf.global_variables_initializer (train.py:178,201)."""
        check(lib().p3d_init_params(self._h, seed))

    def set_param(self, name, value):
        a = np.ascontiguousarray(value, dtype=np.float32)
        check(lib().p3d_set_param(self._h, name.encode(), fptr(a), a.size))

    def get_param(self, name):
        shape = dict((n, s) for n, s, _ in self.variables())[name]
        a = np.empty(shape, np.float32)
        check(lib().p3d_get_param(self._h, name.encode(), fptr(a), a.size))
        return a

    def get_grad(self, name):
        shape = dict((n, s) for n, s, _ in self.variables())[name]
        a = np.empty(shape, np.float32)
        check(lib().p3d_get_grad(self._h, name.encode(), fptr(a), a.size))
        return a

    def load(self, params):
        """saver.restore: {tf variable name: array}."""
        names = set(n for n, _, _ in self.variables())
        missing = names - set(params)
        if missing:
            raise KeyError("checkpoint lacks %d variables, e.g. %s" % (len(missing), sorted(missing)[:3]))
        for n in names:
            self.set_param(n, params[n])

    def restore(self, path):
        """saver.restore (train.py:204-210, gen_pred.py:57-64): `path` is a TF-1.x checkpoint prefix (`.../p3d_1000.ckpt`),
        a directory holding a `checkpoint` state file (the newest bundle is taken), or an .npz keyed by variable names.
        Variables the checkpoint lacks raise; extra ones (e.g. Adam slots of another trainer) are ignored."""
        import os
        from . import tf_checkpoint as tfc
        if os.path.isdir(path):
            latest = tfc.latest_checkpoint(path)
            if latest is None:
                raise FileNotFoundError("no `checkpoint` state file in %s" % path)
            path = latest
        if path.endswith(".npz"):
            self.load(dict(np.load(path)))
        else:
            self.load(tfc.read_checkpoint(path, names=set(n for n, _, _ in self.variables())))
        return path

    def save_checkpoint(self, directory, step, keep=10):
        """saver.save(sess, '<dir>/p3d_<step>.ckpt') with max_to_keep (train.py:180-185,266-267): writes a TF V2 bundle and
        updates the directory's `checkpoint` state file.  Returns the prefix."""
        import os
        from . import tf_checkpoint as tfc
        prefix = os.path.join(directory, "p3d_%d.ckpt" % step)
        tfc.write_checkpoint(prefix, self.save())
        tfc.update_checkpoint_state(directory, prefix, keep)
        return prefix

    def save(self):
        """saver.save: {tf variable name: array} of trainables + moving statistics (train.py:180-185)."""
        return dict((n, self.get_param(n)) for n, _, _ in self.variables())

    # ---- sess.run equivalents --------------------------------------------------------------------
    def _x(self, x):
        a = np.ascontiguousarray(x, dtype=np.float32)
        if a.shape != self.x_shape:
            raise ValueError("x has shape %s, graph was built for %s" % (a.shape, self.x_shape))
        return a

    def _y(self, y):
        a = np.ascontiguousarray(y, dtype=np.float32)
        if a.shape != self.y_shape:
            raise ValueError("y has shape %s, graph was built for %s" % (a.shape, self.y_shape))
        return a

    def forward(self, x, dropout=0.0, training=False, seed=0):
        """sess.run(pred, {x, dropout, training})  (train.py:225-226, gen_pred.py:151)."""
        x = self._x(x)
        pred = np.empty(self.pred_shape, np.float32)
        check(lib().p3d_forward(self._h, fptr(x), int(bool(training)), float(dropout), seed, fptr(pred)))
        return pred

    def block_shapes(self, block_id):
        """(input shape, output shape) of bottleneck `block_id` inside this graph."""
        ishape, oshape = (C.c_int64 * 5)(), (C.c_int64 * 5)()
        check(lib().p3d_block_info(self._h, int(block_id), ishape, oshape))
        return tuple(ishape), tuple(oshape)

    def block_forward(self, block_id, x):
        """Bottleneck `block_id` of this graph in isolation (p3d.py:83-136) on input x [B,D,H,W,inplanes]."""
        ishape, oshape = self.block_shapes(block_id)
        a = np.ascontiguousarray(x, dtype=np.float32)
        if a.shape != ishape:
            raise ValueError("block %d takes %s, got %s" % (block_id, ishape, a.shape))
        out = np.empty(oshape, np.float32)
        check(lib().p3d_block_forward(self._h, int(block_id), fptr(a), a.size, fptr(out), out.size))
        return out

    def block_backward(self, block_id, x, dout):
        """Bottleneck `block_id` in isolation, forward then backward: -> gradient of its input; the block's own variables'
        gradients are then readable with get_grad."""
        ishape, oshape = self.block_shapes(block_id)
        a = np.ascontiguousarray(x, dtype=np.float32)
        d = np.ascontiguousarray(dout, dtype=np.float32)
        if a.shape != ishape or d.shape != oshape:
            raise ValueError("block %d takes %s and returns %s" % (block_id, ishape, oshape))
        din = np.empty(ishape, np.float32)
        check(lib().p3d_block_backward(self._h, int(block_id), fptr(a), a.size, fptr(d), d.size, fptr(din)))
        return din

    def schedule(self, dropout=0.5, seed=0):
        """One train step on the resident inputs, returned as the list of stream operations it issued (p3d_debug_schedule; tests)."""
        import ctypes as C
        cap = 1 << 22
        buf = C.create_string_buffer(cap)
        need = C.c_int64(0)
        check(lib().p3d_debug_schedule(self._h, float(dropout), int(seed), buf, cap, C.byref(need)))
        if need.value > cap:
            raise P3dError("schedule text of %d bytes does not fit" % need.value)
        return buf.value.decode().splitlines()

    def decisions(self):
        """The ReLU gates and max-pool inputs of the last forward pass, as the backward pass of this session uses them
        (p3d_debug_decision_*; tests): {'relu': {BatchNorm scope: bool array [N,D,H,W,C]}, 'pool': [float32 arrays]} -- the `pins`
        an oracle evaluation takes to differentiate the same piecewise-linear branch."""
        import ctypes as C
        out = {"relu": {}, "pool": [], "relu_sites": 0}
        n = lib().p3d_debug_decision_count(self._h)
        for i in range(n):
            kind, n1, n2 = C.c_char_p(), C.c_char_p(), C.c_char_p()
            shape = (C.c_int64 * 5)()
            check(lib().p3d_debug_decision_info(self._h, i, C.byref(kind), C.byref(n1), C.byref(n2), shape))
            shp = tuple(int(v) for v in shape)
            a = np.empty(shp, np.float32)
            if kind.value == b"pool":
                check(lib().p3d_debug_decision_get(self._h, i, fptr(a), None, a.size))
                out["pool"].append(a)
                continue
            b = np.empty(shp, np.float32)
            check(lib().p3d_debug_decision_get(self._h, i, fptr(a), fptr(b), a.size))
            out["relu_sites"] += 1
            out["relu"][n1.value.decode()] = a != 0
            if n2.value:
                out["relu"][n2.value.decode()] = b != 0
        return out

    def set_pointwise_fp16(self, enable=True):
        """BASELINE configs[4]: 1x1x1 convs on the fp16 matrix cores (fp32 accumulate, fp32 storage); fp16-level parity."""
        check(lib().p3d_set_pointwise_fp16(self._h, int(bool(enable))))

    def set_bn_fusion(self, enable=True):
        """BatchNorm + ReLU between the convs of a bottleneck on the convs' operand paths (default) or as passes of their own."""
        check(lib().p3d_set_bn_fusion(self._h, int(enable)))      # 0 off, 1 forward, 2 forward + backward

    def set_attention_mode(self, mode="auto"):
        """How attention() (utils/network.py:183-185) runs: "gemm" stores the score matrix, "flash" recomputes score tiles on
        chip, "auto" picks per block by the size of the score matrix."""
        check(lib().p3d_set_attention_mode(self._h, {"auto": 0, "gemm": 1, "flash": 2}[mode]))

    def predict_windows(self, x):
        """B windows of gen_pred.py:100-168 at once: row k equals forward(x[k:k+1], training=False) of a batch-1
        session, i.e. every batch-statistics BN normalises each clip by its own statistics."""
        x = self._x(x)
        pred = np.empty(self.pred_shape, np.float32)
        check(lib().p3d_predict_windows(self._h, fptr(x), fptr(pred)))
        return pred

    def train_step(self, x, y, dropout=0.5, seed=0):
        """sess.run([train_op, loss], {x, y, dropout, training: True})  (train.py:217-218) -> loss."""
        x, y = self._x(x), self._y(y)
        loss = C.c_float()
        check(lib().p3d_train_step(self._h, fptr(x), fptr(y), float(dropout), seed, C.byref(loss)))
        return loss.value

    def backward(self, x, y, dropout=0.0, seed=0):
        """Forward + loss + gradients without the update -> (loss, pred)."""
        x, y = self._x(x), self._y(y)
        loss = C.c_float()
        pred = np.empty(self.pred_shape, np.float32)
        check(lib().p3d_backward(self._h, fptr(x), fptr(y), float(dropout), seed, C.byref(loss), fptr(pred)))
        return loss.value, pred

    def set_adam(self, lr=1e-4, beta1=0.9, beta2=0.999, eps=1e-8):
        check(lib().p3d_set_adam(self._h, lr, beta1, beta2, eps))

    def activation(self, name):
        shape = (C.c_int64 * 5)()
        check(lib().p3d_activation_info(self._h, name.encode(), shape))
        a = np.empty(tuple(shape), np.float32)
        check(lib().p3d_get_activation(self._h, name.encode(), fptr(a), a.size))
        return a

    # ---- device-resident stepping (bench) ------------------------------------------------------
    def upload(self, x, y):
        check(lib().p3d_upload_inputs(self._h, fptr(self._x(x)), fptr(self._y(y)) if y is not None else None))

    def bucket_audit(self, bucket_floats, dropout=0.0, seed=0, cap=4096):
        """Test hook for the bucketed gradient hand-over of data-parallel training (include/p3d_hip.h,
        p3d_debug_bucket_audit): ([(lo, hi, after_op)], n_train, stale)."""
        lo = (C.c_int64 * cap)(); hi = (C.c_int64 * cap)(); op = (C.c_int32 * cap)()
        n_train = C.c_int64(); stale = C.c_int64()
        n = lib().p3d_debug_bucket_audit(self._h, float(dropout), seed, int(bucket_floats), lo, hi, op, cap, C.byref(n_train),
                                         C.byref(stale))
        if n < 0:
            check(n)
        return [(lo[i], hi[i], op[i]) for i in range(min(n, cap))], n_train.value, stale.value

    def train_step_device(self, dropout=0.0, seed=0):
        check(lib().p3d_train_step_device(self._h, float(dropout), seed))

    def forward_device(self, training=False, dropout=0.0, seed=0):
        check(lib().p3d_forward_device(self._h, int(bool(training)), float(dropout), seed))

    def last_loss(self):
        loss = C.c_float()
        check(lib().p3d_last_loss(self._h, C.byref(loss)))
        return loss.value

    def synchronize(self):
        check(lib().p3d_synchronize(self._h))

    def profile_step(self, dropout=0.0, seed=0):
        cap = 16384
        buf = (P3dOpTime * cap)()
        n = lib().p3d_profile_step(self._h, float(dropout), seed, buf, cap)
        if n < 0:
            raise P3dError(lib().p3d_last_error().decode())
        return [dict(name=r.name.decode(), kernel=r.kernel.decode(), ms=r.ms, flops=r.flops, bytes=r.bytes,
                     phase=r.phase) for r in buf[:min(n, cap)]]

    # ---- data parallel -----------------------------------------------------------------------------
    @staticmethod
    def device_count():
        """GPUs visible to this process (-1: the runtime could not say)."""
        return lib().p3d_device_count()

    @staticmethod
    def comm_unique_id():
        buf = C.create_string_buffer(_lib.P3D_COMM_ID_BYTES)
        check(lib().p3d_comm_unique_id(buf))
        return buf.raw

    def comm_init(self, id_bytes):
        buf = C.create_string_buffer(bytes(id_bytes), _lib.P3D_COMM_ID_BYTES)
        check(lib().p3d_comm_init(self._h, buf))

    def comm_info(self):
        """(ranks, rank, device) as RCCL reports them for this handle's communicator; ranks == 0: no communicator."""
        n, r, d = C.c_int(0), C.c_int(-1), C.c_int(-1)
        check(lib().p3d_comm_info(self._h, C.byref(n), C.byref(r), C.byref(d)))
        return n.value, r.value, d.value
