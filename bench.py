#!/usr/bin/env python3
"""bench.py -- MC-samples/sec of the 784-1200-1200-10 NormalLinear MLP at batch 512
(BASELINE.json metric; configs[1]: bf16 operands / fp32 accumulate, 8 MC samples per forward).

One step = one stochastic forward of S = 8 MC samples over one synthetic batch of 512
(all samples of a layer in one fused sampled-GEMM launch), the Gaussian KL once, the
predictive mean over the samples, and -- for N > 1 -- ONE all-reduce over RCCL of the packed
[KL sums || sum of predictions] buffer, issued asynchronously so that it runs under the next step.  Weak scaling: every rank runs its own 8 samples
(sample ids rank*8 .. rank*8+7 of the same posterior), value = N * 8 * steps / time.

Prints ONE JSON line (rank 0).  Extra keys: `roofline` (dominant kernel, measured live with
events on the launch stream), `cpu_baseline` (torch-CPU port of the reference, N = 1 only),
`f32` (same step in the fp32 parity mode), `train` (N = 1: the reference's training-loop
body, examples/MNIST/train.py:53-65, on the same model -- forward, KL, cross-entropy, HIP
backward, Adam).  `--mode train` makes the training step the headline value
(N > 1: MC samples sharded as in the forward, gradients all-reduced in overlapped buckets).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

DIMS = (784, 1200, 1200, 10)
BATCH = 512
SAMPLES = 8
P_SCALARS = sum(i * o + o for i, o in zip(DIMS[:-1], DIMS[1:]))          # 2 395 210
FLOP_PER_SAMPLE = 2 * BATCH * sum(i * o for i, o in zip(DIMS[:-1], DIMS[1:]))  # 2 450 227 200
PEAK = {"f32": 157.3, "bf16": 2500.0}       # dense MFMA TFLOP/s, MI355X_MICROARCH.md
PEAK_HBM_GBS = 8000.0
# Fabric-side bytes per launch of the dominant kernel from separate rocprofv3 --pmc passes
# (profiles/r01_pmc_layer2_bf16.txt): (2 x FETCH_SIZE + WRITE_SIZE) x 1024, gfx950 FETCH_SIZE correction
# applied.  Not measured live (PMC passes serialise kernels); re-collect with tools/pmc.sh.
TRAFFIC_PMC = {"bf16": 82.7e6, "f32": None}


def posteriors(seed=0):
    """Random-init weights of the architecture (reference init distributions, dense.py:34-42)."""
    gen = torch.Generator().manual_seed(seed)
    out = []
    for i, o in zip(DIMS[:-1], DIMS[1:]):
        bound = 1.0 / i ** 0.5
        out.append(((torch.rand(o, i, generator=gen) * 2 - 1) * bound,
                    torch.randn(o, i, generator=gen) * 0.15 - 2.0,
                    (torch.rand(o, generator=gen) * 2 - 1) * bound,
                    torch.randn(o, generator=gen) * 0.15 - 2.0))
    return out


def build_net(dev, post):
    from bayesianneuralnetworks_amd.nn import NormalLinear, BayesianNetworkModule

    class MLP(BayesianNetworkModule):
        def __init__(self):
            super().__init__(DIMS[0], DIMS[-1], SAMPLES)
            mods = []
            for j, (mw, rw, mb, rb) in enumerate(post):
                L = NormalLinear(mw.shape[1], mw.shape[0])
                with torch.no_grad():
                    L.weight.mean.copy_(mw)
                    L.weight.scale.copy_(rw)
                    L.bias.mean.copy_(mb)
                    L.bias.scale.copy_(rb)
                mods.append(L)
                if j < len(post) - 1:
                    mods.append(torch.nn.ReLU())
            self.layers = torch.nn.Sequential(*mods)

        def _forward(self, x):
            return self.layers(x)

    net = MLP().to(dev)
    net.mc_batched = True
    from bayesianneuralnetworks_amd.nn import fuse_activations
    # ReLU folded into the GEMM epilogue; in bf16 mode hidden activations stay bf16 (the consumer
    # rounds them to bf16 anyway: identical results, half the activation stream)
    fuse_activations(net, bf16_activations=True)
    return net


class Step:
    """One forward of S samples + KL + predictive mean, optionally captured in a HIP graph.

    Launches per step: the multi-tensor KL pair, 3 fused sampled-GEMM kernels (ReLU folded into the
    first two), and the MC reduction, which also bumps the device epoch -- all on one stream
    (measured: forking the 13 us of KL onto a second queue costs more in cross-queue dependency
    latency than it hides).  KL sums and the sum of predictions land directly in the packed buffer
    that the one collective all-reduces."""

    def __init__(self, net, x, rank, world, use_graph):
        from bayesianneuralnetworks_amd import ops, _lib, distributed as bd
        from bayesianneuralnetworks_amd._rng import default_generator
        self.net, self.x, self.rank, self.world = net, x, rank, world
        self.ops, self.lib, self._lib = ops, _lib.load(), _lib
        self.gen = default_generator
        self.graph = None
        dev = x.device
        self.linears = [m for m in net.layers if hasattr(m, "weight")]
        # KL: every rank reduces a 1/world slice of every posterior tensor (eps-independent work).
        # The packed layout is the same on every rank: [sum_t for the 6 tensors | scalar slot | pred];
        # a rank whose slice of a small tensor is empty contributes 0 there.
        self.kl_mu, self.kl_rho, self.kl_idx = [], [], []
        t = 0
        for L in self.linears:
            for p in (L.weight, L.bias):
                lo, hi = bd.shard_range(p.mean.numel(), rank, world)
                if hi > lo:
                    self.kl_mu.append(p.mean.detach().reshape(-1)[lo:hi])
                    self.kl_rho.append(p.scale.detach().reshape(-1)[lo:hi])
                    self.kl_idx.append(t)
                t += 1
        self.T = t
        self.Tl = len(self.kl_idx)
        self.packed = torch.zeros(self.T + 1 + BATCH * DIMS[-1], device=dev)
        self.kl_tmp = torch.zeros(self.Tl + 1, device=dev)
        self.kl_pos = torch.tensor(self.kl_idx, device=dev, dtype=torch.long)
        self.side = torch.cuda.Stream(dev)
        self.comm = torch.zeros_like(self.packed)
        self.pending = None
        if use_graph:
            self._capture()

    def _kl(self):
        if self.world == 1:
            self.ops.kl_normal(self.kl_mu, self.kl_rho, [(0.0, 0.1)] * self.Tl, 1.0, out=self.packed[:self.T + 1])
        else:
            self.ops.kl_normal(self.kl_mu, self.kl_rho, [(0.0, 0.1)] * self.Tl, 1.0, out=self.kl_tmp)
            self._kl_scatter()

    def _kl_scatter(self):
        self.packed[:self.T + 1].zero_()
        self.packed.index_copy_(0, self.kl_pos, self.kl_tmp[:self.Tl])

    def _kl_begin(self, carry):
        """KL's first pass now -- or (carry) inside the classifier head's launch, which leaves 7/8 of the CUs idle --
        and its second pass inside the MC reduction's launch at the end of the step: two launches fewer per forward,
        same values."""
        out = self.packed[:self.T + 1] if self.world == 1 else self.kl_tmp
        return self.ops.kl_normal_begin(self.kl_mu, self.kl_rho, [(0.0, 0.1)] * self.Tl, 1.0, out=out, carry=carry)

    def _body(self):
        """KL placement (BNN_BENCH_KL): 'side' forks it at the start of the step, 'after1' forks it
        behind the first GEMM (so the GEMM is the graph's root node on the main queue), 'serial'
        keeps everything on one stream; 'tail' also keeps one stream and runs KL's second pass inside the MC
        reduction's launch (ops.kl_normal_begin / mc_mean(kl=...)); 'carry' (default) additionally lets the classifier
        head's launch carry KL's first pass."""
        dev = self.x.device
        mode = os.environ.get("BNN_BENCH_KL", "carry")
        with torch.no_grad():
            cur = torch.cuda.current_stream(dev)
            if mode == "side":
                self.side.wait_stream(cur)
                with torch.cuda.stream(self.side):
                    self._kl()
            elif mode == "serial":
                self._kl()
            kl_h = self._kl_begin(mode == "carry") if mode in ("tail", "carry") else None
            hook = None
            if mode == "after1":
                def hook(_m, _i, _o):
                    self.side.wait_stream(cur)
                    with torch.cuda.stream(self.side):
                        self._kl()
                h = self.linears[0].register_forward_hook(hook)
            try:
                ys = self.net.forward_stacked(self.x, SAMPLES, sample0=self.rank * SAMPLES)   # (S, B, 10)
            finally:
                if hook is not None:
                    h.remove()
            # fresh noise on every replay: the reduction also bumps the device epoch (last kernel of the step)
            self.ops.mc_mean(ys, out=self.packed[self.T + 1:], scale=1.0 / (SAMPLES * self.world),
                             advance=self.gen.epoch_dev(dev), kl=kl_h)
            if kl_h is not None and self.world > 1:
                self._kl_scatter()
            if mode in ("side", "after1"):
                cur.wait_stream(self.side)
        return self.packed

    def _capture(self):
        dev = self.x.device
        s = torch.cuda.Stream(dev)
        s.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(s):
            for _ in range(2):
                self._body()
        torch.cuda.current_stream(dev).wait_stream(s)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            self._body()
        self.graph = g

    def run(self):
        if self.graph is not None:
            self.graph.replay()
        else:
            self._body()
        if self.world > 1:
            # RCCL over xGMI, ~20 KB, latency-bound -- so it runs UNDER the next step: the step's result is
            # copied to a communication buffer (stream-ordered, 20 KB) and all-reduced asynchronously; the
            # next replay overwrites `packed`, not the buffer, and the buffer is not reused before its
            # previous reduction has finished (stream-level wait, no host block).
            if self.pending is not None:
                self.pending.wait()
            self.comm.copy_(self.packed)
            self.pending = torch.distributed.all_reduce(self.comm, async_op=True)
            return self.comm
        return self.packed

    def finish(self):
        if self.pending is not None:
            self.pending.wait()
            self.pending = None


class TrainStep:
    """The reference's training-loop body (examples/MNIST/train.py:53-65): zero_grad, S-sample forward,
    KL, mean cross-entropy over the samples, backward, Adam.  Forward, KL, the whole backward of the
    Bayesian layers (re-drawn weights, fused draw-backward), the softmax-cross-entropy on the (S*B, 10)
    logits and the Adam update (one launch for all 12 tensors) are HIP; autograd's bookkeeping is torch.
    N > 1: rank r runs MC samples [r*S, (r+1)*S) of the same batch and the gradients are averaged by
    bucketed all-reduces launched from backward hooks."""

    def __init__(self, net, x, rank, world, use_graph):
        from bayesianneuralnetworks_amd import _lib, ops, optim, distributed as bd
        from bayesianneuralnetworks_amd.nn import KLDivergence, fuse_kl_gradient
        fuse_kl_gradient(True)              # loss.backward() loop: KL gradient rides in the weight-gradient launches
        from bayesianneuralnetworks_amd._rng import default_generator
        self.net, self.x, self.rank, self.world = net, x, rank, world
        dev = x.device
        self.target = torch.randint(0, DIMS[-1], (BATCH,), generator=torch.Generator().manual_seed(3)).to(dev).repeat(SAMPLES)
        self.kld = KLDivergence(number_of_batches=100)
        self.graph = None
        use_graph = use_graph and world == 1
        self.red = bd.GradAllReducer(net.parameters()) if world > 1 else None
        self.ops = ops
        self.opt = optim.Adam(net.parameters(), lr=1e-4)
        self.loss = torch.zeros((), device=dev)
        if use_graph:
            lib = _lib.load()
            cell = default_generator.epoch_dev(dev)
            s = torch.cuda.Stream(dev)
            s.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(s):
                for _ in range(3):
                    self._eager()
            torch.cuda.current_stream(dev).wait_stream(s)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            self.opt.zero_grad(set_to_none=True)
            with torch.cuda.graph(g):
                self._body()
                # fresh noise on every replay, bumped AFTER the backward re-created this step's draws: in Adam's last launch
                self.opt.step(advance=cell)
            self.graph = g

    def _body(self):
        ys = self.net.forward_stacked(self.x, SAMPLES, sample0=self.rank * SAMPLES)      # (S, B, 10)
        loss = self.ops.cross_entropy(ys.reshape(SAMPLES * BATCH, -1), self.target) + self.kld(self.net)
        loss.backward()
        self.loss = loss.detach()           # (no copy launch: under a graph this is the capture's own tensor, rewritten by every replay)

    def _eager(self):
        if self.red is not None:
            self.red.zero_grad()
        else:
            self.opt.zero_grad(set_to_none=True)
        self._body()
        if self.red is not None:
            self.red.finish()
        self.opt.step()

    def run(self):
        if self.graph is not None:
            self.graph.replay()
        else:
            self._eager()


def time_steps(step, steps, warmup, world, dev):
    for _ in range(warmup):
        step.run()
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(steps):
        step.run()
    if hasattr(step, "finish"):
        step.finish()                                   # the last step's collective is inside the timed region
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = t.item()
    return dt


def kernel_roofline(net, x, mode, dev, iters=50):
    """Dominant kernel = the fused sampled GEMM of layer 2 (512 x 1200 x 1200, 8 samples in one
    launch).  Average launch duration from events on the launch stream; algorithmic FLOPs."""
    from bayesianneuralnetworks_amd import _mc
    layer = net.layers[2]
    h = torch.randn(SAMPLES * BATCH, DIMS[1], device=dev).relu_()
    if mode == "bf16":
        h = h.bfloat16()          # the hidden activation the step really feeds this layer
    layer.compute = mode
    with torch.no_grad(), _mc.McContext(SAMPLES, BATCH, 0):
        for _ in range(5):
            layer(h)
        torch.cuda.synchronize(dev)
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            layer(h)
        e1.record()
        torch.cuda.synchronize(dev)
    layer.compute = None
    ms = e0.elapsed_time(e1) / iters
    flops = 2.0 * SAMPLES * BATCH * DIMS[1] * DIMS[2]
    ach = flops / (ms * 1e-3) / 1e12
    # parameter bytes the launch must touch at least once: mu, rho of W and b
    pbytes = 8.0 * (DIMS[1] * DIMS[2] + DIMS[2])
    return {"kernel": "k_linear_sym<sampled> layer2 512x1200x1200 x8 samples (one launch)", "bound": "mfma",
            "achieved": round(ach, 2), "peak": PEAK[mode], "unit": "TFLOP/s", "frac": round(ach / PEAK[mode], 4),
            "traffic": TRAFFIC_PMC.get(mode), "avg_launch_us": round(ms * 1e3, 2),
            "algorithmic_flop_per_launch": flops, "algorithmic_param_bytes_per_launch": pbytes,
            "algorithmic_bytes_per_launch": pbytes + SAMPLES * BATCH * (DIMS[1] * (2 if mode == "bf16" else 4) + DIMS[2] * 4),
            "note": "priced against the bf16 MFMA peak; the launch also makes 8 x 1.44 M eps draws (~10 us of VALU issue) and is bound by "
                    "its per-SIMD issue stream / consume-phase latency chain (DESIGN.md 4): with explicit weights it takes as long"}


def sampler_roofline(dev, iters=20):
    """K1 alone on a working set beyond the Infinity Cache (64 Mi scalars = 768 MiB of traffic):
    HBM-bound, 8 B read + 4 B written per scalar."""
    from bayesianneuralnetworks_amd import ops
    from bayesianneuralnetworks_amd._rng import DrawKey
    n = 64 << 20
    mu = torch.zeros(n, device=dev)
    rho = torch.full((n,), -2.0, device=dev)
    key = DrawKey(1, 1, 0, 1, 0)
    for _ in range(3):
        ops._sample_affine_philox_raw(mu, rho, key)
    torch.cuda.synchronize(dev)
    e0 = torch.cuda.Event(enable_timing=True)
    e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        ops._sample_affine_philox_raw(mu, rho, key)
    e1.record()
    torch.cuda.synchronize(dev)
    ms = e0.elapsed_time(e1) / iters
    gbs = 12.0 * n / (ms * 1e-3) / 1e9
    return {"kernel": "k_sample_affine_philox 64Mi scalars", "bound": "hbm", "achieved": round(gbs, 1),
            "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(gbs / PEAK_HBM_GBS, 4), "traffic": None,
            "avg_launch_us": round(ms * 1e3, 1)}


def kl_roofline(dev, iters=20):
    """K3 alone on a working set beyond the Infinity Cache (64 Mi posterior scalars = 512 MiB read):
    HBM-bound, 8 algorithmic bytes per scalar (mu, rho), one double partial per 2048 scalars written."""
    from bayesianneuralnetworks_amd import ops
    n = 64 << 20
    mu = torch.zeros(n, device=dev)
    rho = torch.full((n,), -2.0, device=dev)
    out = torch.empty(2, device=dev)
    with torch.no_grad():
        for _ in range(3):
            ops.kl_normal([mu], [rho], [(0.0, 0.1)], 1.0, out=out)
        torch.cuda.synchronize(dev)
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            ops.kl_normal([mu], [rho], [(0.0, 0.1)], 1.0, out=out)
        e1.record()
        torch.cuda.synchronize(dev)
    ms = e0.elapsed_time(e1) / iters
    gbs = 8.0 * n / (ms * 1e-3) / 1e9
    return {"kernel": "k_kl_partial + k_kl_final, 64Mi scalars", "bound": "hbm", "achieved": round(gbs, 1),
            "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(gbs / PEAK_HBM_GBS, 4), "traffic": None,
            "avg_launch_us": round(ms * 1e3, 1)}


def cpu_baseline(post, x_cpu):
    """The torch-CPU port of the reference (oracle/reference_port.py, pinned bit-for-bit to the
    reference by tests/test_oracle_golden.py) on this machine's host cores."""
    from oracle import reference_port as port
    # threads = the CPU share this process really has (the GPU box gives one GPU 16 cores; asking
    # torch for all 256 logical CPUs of the host oversubscribes them ~16x)
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    cores = max(1, min(avail, 16))
    torch.set_num_threads(cores)
    with torch.no_grad():
        for _ in range(2):
            port.mlp_forward(x_cpu, post, SAMPLES)
            port.kl_divergence_loss(post)
        n = 0
        t0 = time.perf_counter()
        while True:
            ys = port.mlp_forward(x_cpu, post, SAMPLES)
            port.kl_divergence_loss(post)
            torch.stack(ys).mean(0)
            n += 1
            dt = time.perf_counter() - t0
            if dt > 12.0 or n >= 400:       # a bounded ~12 s sample of the same workload
                break
    try:
        model = [l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")][0]
    except Exception:
        model = "unknown"
    return {"value": round(n * SAMPLES / dt, 2), "unit": "MC-samples/s", "cores": cores, "kind": "port",
            "sample": "%d forwards of 8 MC samples, batch 512, + KL + predictive mean; fp32, no_grad; %.1f s; %s"
                      % (n, dt, model)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--mode", default="forward", choices=["forward", "train"])
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # "nccl" = RCCL over xGMI.  BNN_BENCH_BACKEND=gloo only rehearses the N > 1 code path on a
        # box with fewer GPUs than ranks (ranks then share a card; the number is not a result).
        torch.distributed.init_process_group(os.environ.get("BNN_BENCH_BACKEND", "nccl"))
    local = local % max(1, torch.cuda.device_count())
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    import bayesianneuralnetworks_amd as bnn
    from bayesianneuralnetworks_amd import _lib
    _lib.load()                                     # fail loudly if the HIP library is missing
    post = posteriors(0)
    net = build_net(dev, post)
    x_cpu = torch.randn(BATCH, DIMS[0], generator=torch.Generator().manual_seed(1))
    x = x_cpu.to(dev)
    bnn.manual_seed(2)

    results = {}
    for mode in ([args.dtype] + (["f32"] if args.dtype != "f32" else [])):
        bnn.set_compute(mode)
        # bf16 mode: the synthetic batch is resident in HBM as bf16 (the first layer would round its
        # A operand to bf16 anyway -- identical results, half the input stream); fp32 mode: fp32.
        x_in = x.bfloat16() if mode == "bf16" else x
        step = Step(net, x_in, rank, world, not args.no_graph)
        steps = args.steps if mode == args.dtype else max(10, args.steps // 4)
        dt = time_steps(step, steps, args.warmup, world, dev)
        results[mode] = (world * SAMPLES * steps / dt, dt / steps * 1e3, steps)
    bnn.set_compute(args.dtype)

    # training step (own copy of the model: Adam moves the parameters)
    train = None
    if args.mode == "train" or world == 1:
        tnet = build_net(dev, post)
        tsteps = args.steps if args.mode == "train" else max(10, args.steps // 4)
        tstep = TrainStep(tnet, x.bfloat16() if args.dtype == "bf16" else x, rank, world, not args.no_graph)
        dt = time_steps(tstep, tsteps, args.warmup, world, dev)
        train = (world * SAMPLES * tsteps / dt, dt / tsteps * 1e3, tsteps, float(tstep.loss))
        del tstep, tnet

    if rank == 0:
        val, ms, steps = results[args.dtype]
        line = {
            "metric": "MC-samples/sec (node), 784-1200-1200-10 BayesianLinear MLP, batch 512",
            "value": round(val, 1), "unit": "MC-samples/s", "n_gpus": world, "steps": steps,
            "warmup": args.warmup, "ms_per_step": round(ms, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16" if args.dtype == "bf16" else "f32", "data": "synthetic",
            "config": {"workload": "configs[1]: MNIST-shape 784-1200-1200-10 NormalLinear MLP, batch 512, "
                                   "8 MC samples per forward per GPU, KL once per forward, predictive mean; "
                                   "bf16 operands / fp32 accumulate" if args.dtype == "bf16" else
                                   "784-1200-1200-10 NormalLinear MLP, batch 512, 8 MC samples per forward per GPU, "
                                   "fp32 parity mode (wide layers: bf16x3 splits on the bf16 MFMA, fp32-accurate)",
                       "samples_per_step_per_gpu": SAMPLES, "batch": BATCH,
                       "hip_graph": not args.no_graph, "launches_per_step": 4 if (world == 1 and os.environ.get("BNN_BENCH_KL", "carry") == "carry") else None,
                       "collective": "one all-reduce of [6 KL sums, KL scalar, 512x10 prediction sum] fp32" if world > 1 else None},
        }
        if train is not None:
            line["train"] = {"value": round(train[0], 1), "unit": "MC-samples/s", "ms_per_step": round(train[1], 4),
                             "steps": train[2], "loss": round(train[3], 4), "hip_graph": (not args.no_graph) and world == 1,
                             "note": "training-loop body of examples/MNIST/train.py:53-65 (fwd + KL + CE + HIP bwd + Adam)"}
            if args.mode == "train":
                line["forward"] = {"value": line["value"], "ms_per_step": line["ms_per_step"]}
                line["metric"] = "MC-samples/sec (node), TRAINING step, 784-1200-1200-10 BayesianLinear MLP, batch 512"
                line["value"], line["ms_per_step"], line["steps"] = line["train"]["value"], line["train"]["ms_per_step"], train[2]
        if "f32" in results and args.dtype != "f32":
            line["f32"] = {"value": round(results["f32"][0], 1), "ms_per_step": round(results["f32"][1], 4),
                           "note": "same step in the 1e-5 parity mode (fp32 operands; wide layers as bf16x3 splits on the bf16 MFMA)"}
        line["roofline"] = kernel_roofline(net, x, args.dtype, dev)
        line["roofline_sampler"] = sampler_roofline(dev)
        line["roofline_kl"] = kl_roofline(dev)
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(post, x_cpu)
        print(json.dumps(line), flush=True)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
