"""CPU baseline timing in a process of its own (bench.py `cpu_baseline`; TEST INFRASTRUCTURE, see oracle/nn.py).

`python -m oracle.cpu_time torch|numpy [clips]` runs ONE train step (forward, Smooth-L1 sum, backward, Adam) of
p3d_unet at the reference architecture on `clips` synthetic 16x112x112 clips and prints a JSON line.  A child process
because the torch leg must not share a process with libp3dhip (load-order rule) and so that thread
pools start clean."""
import json
import os
import sys
import time


def usable_cores():
    n = len(os.sched_getaffinity(0))
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                if q > 0:
                    n = min(n, max(1, q // int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())))
        except Exception:
            pass
    return n


def main():
    kind = sys.argv[1]
    batch = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    cores = usable_cores()
    import numpy as np
    from oracle import p3d
    params = p3d.init_params(1, "unet", None)
    x = p3d.synthetic_clip(0, (batch, 16, 112, 112, 3))
    y = p3d.synthetic_target(3, (batch, 16, 112, 112))
    if kind == "numpy":
        try:
            from threadpoolctl import threadpool_limits
            threadpool_limits(limits=cores)
        except Exception:
            pass
        state = {"t": 0, "m": {}, "v": {}}
        t0 = time.time()
        p3d.train_step(params, state, x, y)
        dt = time.time() - t0
        what = "numpy/OpenBLAS restatement (oracle/p3d.py)"
    else:
        import torch
        from oracle import torch_port
        torch.set_num_threads(cores)
        m = torch_port.TorchP3D(params, torch.float32)
        xt, yt = torch.tensor(x), torch.tensor(y)
        t0 = time.time()
        pred = m.unet(xt, True)
        loss = torch_port.smooth_l1_sum(pred.reshape(yt.shape), yt)
        loss.backward()
        with torch.no_grad():                      # Adam, first step (m = v = 0): tf.train.AdamOptimizer's epsilon-hat form
            lr_t = 1e-4 * (1 - 0.999) ** 0.5 / (1 - 0.9)
            for p in m.p.values():
                if p.grad is not None:
                    mm = 0.1 * p.grad
                    vv = 0.001 * p.grad * p.grad
                    p -= lr_t * mm / (vv.sqrt() + 1e-8)
        dt = time.time() - t0
        what = "torch-CPU (oneDNN) composition (oracle/torch_port.py)"
    print(json.dumps(dict(value=round(batch / dt, 4), unit="clips/s", cores=cores, kind="port", seconds=round(dt, 2),
                          sample="one train step (fwd+loss+bwd+Adam) of %d clip(s) 16x112x112, %s, %.1f s" % (batch, what, dt))))


if __name__ == "__main__":
    main()
