#!/usr/bin/env python3
"""bench.py -- MC-samples/sec of the 784-1200-1200-10 NormalLinear MLP at batch 512
(BASELINE.json metric; configs[1]: bf16 operands / fp32 accumulate, 8 MC samples per forward).

One step = one stochastic forward of the MC samples over one synthetic batch of 512 (all samples of
a layer in one launch), the Gaussian KL once, the predictive mean over the samples, and -- for
N > 1 -- ONE all-reduce over RCCL of the packed [KL sums || sum of predictions] buffer, issued
asynchronously so that it runs under the next step.

`--gpus N` with no WORLD_SIZE in the environment starts N rank processes of this script itself
(before the parent makes any GPU call) and relays rank 0's line.  N > 1 headline = STRONG scaling
(SURVEY.md 8e / north_star "one sample per GPU"): the S = 8 global MC samples are sharded S / N per
rank (`distributed.shard_samples`), value = 8 * steps / time; the weak-scaling number (8 samples on
every rank, value = N * 8 * steps / time) rides along under "weak".

Timing: `--windows` (default 25) repetitions of the timed window of `--steps` steps (barrier + synchronize around every
window) inside one run; `value` / `ms_per_step` are the MEDIAN window, `timing` holds the fastest and slowest.
A run whose oracle check fails, or that leaves the device error word set, prints its line with a `failed` list and exits
non-zero.

Prints ONE JSON line (rank 0).  Extra keys: `checked` / `checked_f32` (one more replay of the timed object compared
with the CPU oracle on the replay's own draw keys -- a checker, never timed; N > 1: the all-reduced buffer against global
sample ids, every rank replays, rank 0 compares), `device_error_word`, `roofline` (dominant
kernel, measured live with events on the launch stream), `roofline_conv_lenet` / `roofline_conv_cifar`
/ `roofline_wide_f32` (configs[2], [3], [4] layer launches), `roofline_lenet_net` / `roofline_wide_stack` (configs[2] and [4] as
the whole networks they name), `cpu_baseline` (torch-CPU port of the
reference, N = 1 only), `f32` (same step in the fp32 parity mode), `train` (N = 1: the reference's
training-loop body, examples/MNIST/train.py:53-65, on the same model).
"""
import argparse
import json
import os
import socket
import subprocess
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
# fp32 parity mode on three bf16 planes: six bf16 MFMA products stand for one fp32 product, so the ceiling of the method in
# fp32-equivalent FLOP/s is the bf16 peak / 6 (above the native fp32 MFMA peak of 157.3)
PEAK_X3 = PEAK["bf16"] / 6.0
PEAK_HBM_GBS = 8000.0
# Fabric-side bytes per launch of the dominant kernel come from separate rocprofv3 --pmc passes (PMC passes
# serialise kernels, so they are not collected live): tools/pmc.sh writes the summary under profiles/ and
# the number is recorded in this file; a kernel that is not in the file reports traffic = null.
TRAFFIC_FILE = os.path.join(ROOT, "profiles", "pmc_traffic.json")


def _pmc_entry(kernel_tag):
    try:
        with open(TRAFFIC_FILE) as f:
            return json.load(f).get(kernel_tag)
    except (OSError, ValueError):
        return None


def pmc_traffic(kernel_tag):
    ent = _pmc_entry(kernel_tag)
    if not ent:
        return None, None
    return ent.get("traffic_bytes"), {"file": ent.get("source"), "date": ent.get("date")}


MAX_CLOCK_MHZ = 2400.0      # MI355X_MICROARCH.md: the clock the 2.5 PFLOP/s bf16 peak is quoted at


def pmc_mfma_util(kernel_tag, launch_us):
    """MFMA utilisation of a launch from the PMC pass recorded in profiles/pmc_traffic.json: the matrix pipe's busy cycles per
    SIMD (SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs; = 16 cycles x the MFMAs a SIMD issued) over the launch's duration -- measured
    live here -- in cycles of the MAXIMUM clock: a lower bound of the pipe's busy fraction (the chip holds ~1.85-2.0 GHz under
    this loop, tools/dense_stamps.py: at that clock the same figure is ~1.25 x higher)."""
    ent = _pmc_entry(kernel_tag)
    if not ent or "mfma_busy_cycles_per_simd" not in ent or not launch_us:
        return None
    return {"mfma_util": round(ent["mfma_busy_cycles_per_simd"] / (launch_us * MAX_CLOCK_MHZ), 4),
            "mfma_busy_cycles_per_simd": ent["mfma_busy_cycles_per_simd"], "mfma_insts_per_launch": ent.get("mfma_insts"),
            "at_clock_mhz": MAX_CLOCK_MHZ, "source": ent.get("mfma_source"), "date": ent.get("date")}


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


def resident_input(x, mode):
    """The synthetic batch as it sits in HBM when the timed region starts.  bf16 mode: bf16 (the first layer would round
    its A operand to bf16 anyway -- identical results, half the input stream) with 128-B aligned rows (row pitch 832 =
    roundup(784, 64) elements: the dense kernel's LDS-DMA then reads whole cache lines); fp32 mode: fp32, dense."""
    if mode != "bf16":
        return x
    K = x.shape[1]
    buf = torch.zeros(x.shape[0], (K + 63) // 64 * 64, dtype=torch.bfloat16, device=x.device)
    buf[:, :K] = x
    return buf[:, :K]


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
    # ... and layer 2 may run fused with the classifier head behind it when the step asks for the predictive mean
    fuse_activations(net, bf16_activations=True, fuse_head=True)
    return net


class Step:
    """One forward of `samples` MC samples (global ids sample0 ..) + KL + this rank's share of the predictive
    mean, optionally captured in a HIP graph.  All launches on one stream (measured: forking the KL onto a
    second queue costs more in cross-queue dependency latency than it hides).  KL sums and the sum of
    predictions land directly in the packed buffer that the one collective all-reduces."""

    def __init__(self, net, x, rank, world, use_graph, samples=SAMPLES, sample0=None, total_samples=None, private=False,
                 fuse_head=None):
        from bayesianneuralnetworks_amd import ops, _lib, distributed as bd
        from bayesianneuralnetworks_amd._rng import default_generator
        self.net, self.x, self.rank, self.world = net, x, rank, world
        # private: this step's launches read and bump an epoch word (and use a KL workspace) of their own, so that it can be
        # replayed concurrently with another step on another stream (PipelinedSteps)
        self.private = private
        # fuse_head (default: on, BNN_BENCH_FUSE_HEAD=0 switches it off): layer 2 and the classifier head as ONE launch -- the
        # step asks for the predictive mean, not for the samples (net.predictive_mean) -- four launches per step instead of five
        self.fuse_head = (os.environ.get("BNN_BENCH_FUSE_HEAD", "1") != "0") if fuse_head is None else bool(fuse_head)
        self.cell = None
        self.samples = samples
        self.sample0 = rank * samples if sample0 is None else sample0
        self.total = world * samples if total_samples is None else total_samples
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
        self.comm = torch.zeros_like(self.packed)
        self.pending = None
        if private:
            with self.gen.private_epoch_cell(dev) as cell:
                self.cell = cell
                saved = ops._kl_ws.pop((dev.type, dev.index), None)        # a KL workspace of its own
                try:
                    self._capture() if use_graph else self._body()
                finally:
                    self.kl_ws = ops._kl_ws.get((dev.type, dev.index))      # the captured launches hold its address: keep it alive
                    if saved is not None:
                        ops._kl_ws[(dev.type, dev.index)] = saved
                    else:
                        ops._kl_ws.pop((dev.type, dev.index), None)
            torch.cuda.synchronize(dev)
        elif use_graph:
            self._capture()
        else:
            self._body()
            torch.cuda.synchronize(dev)
        # the draw keys the captured launches carry (a later Step on the same net re-keys the layers)
        self.keys = [(L.weight.draw_key, L.bias.draw_key) for L in self.linears]

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
        """KL placement (BNN_BENCH_KL): 'serial' runs the KL pair first on the one stream; 'tail' runs KL's second
        pass inside the MC reduction's launch (ops.kl_normal_begin / mc_mean(kl=...)); 'carry' (default) additionally
        lets the classifier head's launch carry KL's first pass."""
        dev = self.x.device
        mode = os.environ.get("BNN_BENCH_KL", "carry")
        with torch.no_grad():
            if mode == "serial":
                self._kl()
            kl_h = self._kl_begin(mode == "carry") if mode in ("tail", "carry") else None
            # fresh noise on every replay: the reduction also bumps the device epoch (last kernel of the step)
            if self.fuse_head:
                self.net.predictive_mean(self.x, self.samples, sample0=self.sample0, out=self.packed[self.T + 1:],
                                         scale=1.0 / self.total, advance=self.gen.epoch_dev(dev), kl=kl_h)
            else:
                ys = self.net.forward_stacked(self.x, self.samples, sample0=self.sample0)   # (S, B, 10)
                self.ops.mc_mean(ys, out=self.packed[self.T + 1:], scale=1.0 / self.total,
                                 advance=self.gen.epoch_dev(dev), kl=kl_h)        # (the private cell while one is installed)
            if kl_h is not None and self.world > 1:
                self._kl_scatter()
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
        if self.world > 1 and self.private:
            # one of several steps in flight (PipelinedSteps): its buffer is not touched by the other steps, so the all-reduce
            # runs IN PLACE on `packed` -- this step's next replay (depth iterations later, on this stream) waits for it first
            if self.pending is not None:
                self.pending.wait()
            self.graph.replay() if self.graph is not None else self._body()
            self.pending = torch.distributed.all_reduce(self.packed, async_op=True)
            return self.packed
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


class PipelinedSteps:
    """`depth` independent steps in flight, each a captured graph with its own buffers, epoch word and KL workspace, replayed
    round-robin on `depth` streams: the draw of one step (VALU-bound) runs beside the contractions of another (L2 -> LDS
    ingest / MFMA-bound).  A serving-style throughput pipeline: every step is still one complete stochastic forward + KL +
    predictive mean over one batch; its latency is that of the single-stream step (reported next to the throughput)."""

    def __init__(self, net, x, depth, rank=0, world=1, **step_kw):
        dev = x.device
        self.dev = dev
        self.steps = [Step(net, x, rank, world, True, private=True, **step_kw) for _ in range(depth)]
        self.streams = [torch.cuda.Stream(dev) for _ in range(depth)]
        self.i = 0
        self.world = world
        cur = torch.cuda.current_stream(dev)
        for st in self.streams:
            st.wait_stream(cur)
        # untimed pre-roll: the round-robin of `depth` graphs on `depth` queues takes a few hundred replays to settle into its
        # steady interleaving (measured: 153 k MC-samples/s timed over 200 steps after 20 warm-up steps, 160 k over 1000 after 200)
        for _ in range(int(os.environ.get("BNN_BENCH_PREROLL", "256"))):
            self.run()
        self.finish()
        torch.cuda.synchronize(dev)

    def run(self):
        """Replay the next step on its stream.  N > 1: Step.run also issues that step's one all-reduce (asynchronous, on the
        step's own communication buffer; every rank issues the collectives in the same round-robin order)."""
        k = self.i % len(self.steps)
        self.i += 1
        with torch.cuda.stream(self.streams[k]):
            return self.steps[k].run()

    def finish(self):
        for k, st in enumerate(self.steps):
            with torch.cuda.stream(self.streams[k]):
                st.finish()
        cur = torch.cuda.current_stream(self.dev)
        for st in self.streams:
            cur.wait_stream(st)


# ------------------------------------------------------------------------------------------ checker
# bf16 mode: a drawn weight (or hidden activation) within the eps twin's 1e-6 of a bf16 rounding boundary rounds the other way on
# one side (one bf16 ulp on ~0.3 % of the weights).  Measured on MI355X boxes over rounds 2 and 3: 2.2e-3 .. 2.9e-3 of the
# output scale (profiles/r02_bench_path_check.jsonl, gpurun_out/bench_path_check.jsonl).  Tolerance: TWICE that maximum =
# 1.5 * 2^-8 of the output scale (round 2 allowed a whole bf16 ulp, 2^-7: a real regression of 3 x would have passed).
TOL_BF16 = 1.5 * 2.0 ** -8
TOL_F32 = 1e-5


def oracle_collect(step, replay=True):
    """Run the step ONCE more (every rank: for N > 1 the replay ends in the step's all-reduce) and return what the check needs:
    the packed result as a numpy array -- for N > 1 the ALL-REDUCED buffer -- and the device epoch that replay's launches read.
    replay=False: nothing is launched; the result the step's LAST replay left behind (epoch = cell - 1): the way to check a result
    that was produced while other steps were in flight on other streams (PipelinedSteps, N = 1)."""
    dev = step.x.device
    cell = step.cell if getattr(step, "cell", None) is not None else step.gen.epoch_dev(dev)
    torch.cuda.synchronize(dev)
    if replay:
        e_dev = int(cell[0].item())                 # the epoch this replay's launches will read
        out = step.run()
        step.finish()                               # N > 1: the asynchronous all-reduce of this replay
        torch.cuda.synchronize(dev)
        e_after = int(cell[0].item())
    else:
        e_after = int(cell[0].item())
        e_dev = e_after - 1                         # the epoch the last replay read (it bumped the word at its end)
        out = step.packed
    return out.detach().float().cpu().numpy().copy(), e_dev, e_after


def oracle_evaluate(got, keys, e_dev, e_after, post, x_cpu, mode, nsamples, world=1, rows=64, tap=None):
    """CHECKER -- never timed, never on the product path.  `got`: a step's packed result [6 KL slots | scalar slot | B x 10
    predictive mean]; N = 1: slot 6 holds the KL scalar; N > 1 (the all-reduced buffer): slots 0..5 hold each tensor's KL SUM
    over all ranks' shards and the scalar is formed here as the reference does (mean over the tensor, mean over the tensors,
    loss.py:28,38).  Compared with the CPU oracle (oracle/bnn_oracle.c, pinned to the reference by tests/test_oracle_golden.py)
    on the draw keys of that replay with GLOBAL MC sample ids 0 .. nsamples - 1: KL scalar, `rows` rows of the predictive mean
    and (if `tap`, a (S, rows, N2) copy of the layer-2 output a forward hook made) of the layer-2 output.  bf16 mode: the
    oracle is fed what the kernels feed the MFMA -- bf16-rounded inputs, drawn weights and hidden activations, fp32 bias."""
    import numpy as np
    from oracle import oracle as orc
    T = 2 * len(post)
    rnd = orc.bf16_round if mode == "bf16" else (lambda a: np.asarray(a, np.float32))
    h0 = rnd(x_cpu[:rows].float().numpy())
    pred = np.zeros((rows, DIMS[-1]), np.float64)
    h2_ref = []
    for s in range(nsamples):
        h = h0
        for li, ((mw, rw, mb, rb), (kw, kb)) in enumerate(zip(post, keys)):
            # global sample id s (a rank's key starts at its own sample0: the stream is addressed by the id, not by the rank)
            ew = orc.eps_fill(kw.seed, kw.stream, s, kw.epoch_host, e_dev + kw.epoch_dev_delta, tuple(mw.shape), kw.gen)
            eb = orc.eps_fill(kb.seed, kb.stream, s, kb.epoch_host, e_dev + kb.epoch_dev_delta, tuple(mb.shape), kb.gen)
            w = rnd(orc.sample_affine(mw.numpy(), rw.numpy(), ew))
            b = orc.sample_affine(mb.numpy(), rb.numpy(), eb)
            h = orc.linear(h, w, b)
            if li < len(post) - 1:
                h = rnd(np.maximum(h, 0.0))
            if li == 1:
                h2_ref.append(h)
        pred += h
    pred /= nsamples
    tensors = []
    for mw, rw, mb, rb in post:
        tensors += [(mw.numpy(), rw.numpy(), 0.0, 0.1), (mb.numpy(), rb.numpy(), 0.0, 0.1)]
    kl_ref = orc.kl_divergence(tensors)
    if world == 1:
        kl_got = float(got[T])
    else:
        kl_got = float(np.mean([float(got[t]) / tensors[t][0].size for t in range(T)]))
    pm = got[T + 1:].reshape(BATCH, DIMS[-1])[:rows].astype(np.float64)
    rms = float(np.sqrt((pred ** 2).mean()))
    tol = TOL_BF16 if mode == "bf16" else TOL_F32
    res = {"mode": mode, "rows": rows, "samples": nsamples, "world": world, "epoch_dev": e_dev, "epoch_advanced": e_after == e_dev + 1,
           "kl": kl_got, "kl_ref": kl_ref, "kl_rel_err": abs(kl_got - kl_ref) / abs(kl_ref), "kl_tol": 1e-5,
           "pred_max_err": float(np.abs(pm - pred).max()), "pred_rms": rms, "tol": tol,
           "pred_tol_abs": tol * max(1.0, rms)}
    ok = res["kl_rel_err"] <= 1e-5 and res["pred_max_err"] <= res["pred_tol_abs"] and res["epoch_advanced"]
    if tap is not None:
        t = tap.detach().float().cpu().numpy().astype(np.float64)
        ref = np.stack(h2_ref).astype(np.float64)
        r2 = float(np.sqrt((ref ** 2).mean()))
        res.update({"h2_max_err": float(np.abs(t - ref).max()), "h2_rms": r2, "h2_tol_abs": tol * max(1.0, r2)})
        # a bf16-STORED activation may also round the other way on one side: one bf16 ulp of its value
        ulp = np.abs(ref) * 2.0 ** -7 if mode == "bf16" else 0.0
        ok = ok and bool((np.abs(t - ref) <= res["h2_tol_abs"] + ulp).all())
    res["ok"] = bool(ok)
    return res


def oracle_check(step, post, x_cpu, mode, rows=64, tap=None, replay=True, evaluate=True):
    """oracle_collect + oracle_evaluate.  N > 1: every rank replays (the collective), `evaluate` says who compares (rank 0);
    the global sample count is the step's `total`."""
    got, e_dev, e_after = oracle_collect(step, replay)
    if not evaluate:
        return None
    return oracle_evaluate(got, step.keys, e_dev, e_after, post, x_cpu, mode, step.total, step.world, rows, tap)


class TrainStep:
    """The reference's training-loop body (examples/MNIST/train.py:53-65): zero_grad, S-sample forward,
    KL, mean cross-entropy over the samples, backward, Adam.  Forward, KL, the whole backward of the
    Bayesian layers (re-drawn weights, fused draw-backward), the softmax-cross-entropy on the (S*B, 10)
    logits and the Adam update (one launch for all 12 tensors) are HIP; autograd's bookkeeping is torch.
    N > 1: rank r runs MC samples [r*S, (r+1)*S) of the same batch and the gradients are averaged by
    bucketed all-reduces launched from backward hooks."""

    def __init__(self, net, x, rank, world, use_graph):
        from bayesianneuralnetworks_amd import ops, optim, distributed as bd
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
    """`warmup` untimed steps, then EXACTLY `steps` steps between barrier + synchronize on both sides; max over ranks."""
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


def time_windows(step, steps, warmup, world, dev, windows):
    """`windows` repetitions of the timed window of time_steps (the warm-up before the first only) -> the list of their
    durations.  A 20-step window is ~1 ms: one window is one sample of a noisy quantity, so the line reports the MEDIAN
    window (and the fastest and slowest next to it); every window is `steps` complete steps between barriers."""
    out = [time_steps(step, steps, warmup, world, dev)]
    for _ in range(max(1, windows) - 1):
        out.append(time_steps(step, steps, 0, world, dev))
    return out


def _median(v):
    v = sorted(v)
    n = len(v)
    return v[n // 2] if n % 2 else 0.5 * (v[n // 2 - 1] + v[n // 2])


def _time_launches(fn, dev, iters, warm=5):
    """Average duration of fn() over `iters` back-to-back calls, by events on the launch stream (torch's current
    stream IS the stream the C-ABI launches on: _lib.stream_ptr)."""
    for _ in range(warm):
        fn()
    torch.cuda.synchronize(dev)
    e0 = torch.cuda.Event(enable_timing=True)
    e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize(dev)
    return e0.elapsed_time(e1) / iters


def _graph_time(fn, dev, reps=10, iters=20):
    """us per call of fn(), replayed from a HIP graph of `reps` calls (eager Python launches are host-bound at these sizes)."""
    side = torch.cuda.Stream(dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(side):
        fn()
        fn()
    torch.cuda.current_stream(dev).wait_stream(side)
    torch.cuda.synchronize(dev)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            fn()
    return _time_launches(g.replay, dev, iters, warm=2) * 1e3 / reps


def kernel_roofline(net, x, mode, dev):
    """Dominant MFMA kernel = the contraction of layer 2 (512 x 1200 x 1200, 8 samples in one launch).  bf16 mode: the
    dense GEMM on the drawn weights (k_dense_bf16; its weights come from the step's ONE draw launch, priced under
    `roofline_draw`); fp32 mode: the fused sampled kernel.  Average launch duration from events on the launch stream
    around graph replays of 10 launches; algorithmic FLOPs."""
    from bayesianneuralnetworks_amd import _mc, ops
    from bayesianneuralnetworks_amd._rng import DrawKey, generator_for
    layer = net.layers[2]
    h = torch.randn(SAMPLES * BATCH, DIMS[1], device=dev).relu_()
    flops = 2.0 * SAMPLES * BATCH * DIMS[1] * DIMS[2]
    pbytes = 8.0 * (DIMS[1] * DIMS[2] + DIMS[2])
    tag = "layer2_" + mode
    traffic, src = pmc_traffic(tag)
    out = {"bound": "mfma", "peak": PEAK[mode], "unit": "TFLOP/s", "traffic": traffic, "traffic_source": src,
           "algorithmic_flop_per_launch": flops}
    if mode == "bf16":
        # the hidden activation the step really feeds this layer: bf16, 128-B aligned rows (what layer 1's epilogue writes)
        hb = torch.zeros(SAMPLES, BATCH, (DIMS[1] + 63) // 64 * 64, dtype=torch.bfloat16, device=dev)
        hb[:, :, :DIMS[1]] = h.view(SAMPLES, BATCH, -1)
        hv = hb[:, :, :DIMS[1]]
        g = generator_for("bf16")                   # the stream the bf16 mode keys its draws with (BNN_GEN_PHILOX7_U16)
        kw, kb = DrawKey(1, 1, 0, SAMPLES, 0, gen=g), DrawKey(1, 2, 0, SAMPLES, 0, gen=g)
        spec = (layer.weight.mean.detach(), layer.weight.scale.detach(), layer.bias.mean.detach(), layer.bias.scale.detach(), kw, kb)
        pre = ops.draw_layers([spec], SAMPLES)[0]
        ld = hb.shape[2]
        us = _graph_time(lambda: ops._dense_raw(hv, BATCH * ld, BATCH, pre, DIMS[1], True, torch.bfloat16, ldx=ld, pad_rows=True), dev)
        us_draw = _graph_time(lambda: ops.draw_layers([spec], SAMPLES), dev)
        ach = flops / us / 1e6
        out.update({"kernel": "k_dense_bf16<4,5,2,2,4> (128x160 tile, 4-stage ring): layer 2, 512x1200x1200 x8 samples on drawn weights (%s)" % tag,
                    "achieved": round(ach, 2), "frac": round(ach / PEAK[mode], 4), "avg_launch_us": round(us, 2),
                    "algorithmic_bytes_per_launch": SAMPLES * (BATCH * DIMS[1] * 2 + DIMS[2] * DIMS[1] * 2 + BATCH * DIMS[2] * 2),
                    "mfma": pmc_mfma_util(tag, us),
                    "layer_end_to_end": {"draw_us": round(us_draw, 2), "total_us": round(us + us_draw, 2),
                                         "tflops": round(flops / (us + us_draw) / 1e6, 1),
                                         "frac": round(flops / (us + us_draw) / 1e6 / PEAK[mode], 4),
                                         "note": "layer 2 alone = its own draw launch (8 x 1.44 M weights) + the contraction; in the step the "
                                                 "draw of all three layers is ONE launch (roofline_draw)"}})
        return out
    # fp32 parity mode: the same dense kernel on three bf16 planes per operand, six plane-pair k-steps per k-block
    # (bnn_dense_forward_x3); its weights come from the step's one draw launch (three planes), its input planes from layer
    # 1's epilogue.  FLOPs counted once (the fp32 contraction's), peak = the fp32 MFMA rate the mode replaces.
    kw, kb = DrawKey(1, 1, 0, SAMPLES, 0), DrawKey(1, 2, 0, SAMPLES, 0)
    spec = (layer.weight.mean.detach(), layer.weight.scale.detach(), layer.bias.mean.detach(), layer.bias.scale.detach(), kw, kb)
    pre = ops.draw_layers([spec], SAMPLES, x3=True)[0]
    xp = ops.split_x3(h).view(3, SAMPLES, BATCH, -1)
    us = _graph_time(lambda: ops._dense_raw_x3(xp, False, BATCH, pre, DIMS[1], True, False), dev)
    us_draw = _graph_time(lambda: ops.draw_layers([spec], SAMPLES, x3=True), dev)
    ach = flops / us / 1e6
    out.update({"kernel": "k_dense_bf16<4,5,2,2,4> on three-plane operands (bf16x3): layer 2, 512x1200x1200 x8 samples (%s)" % tag,
                "achieved": round(ach, 2), "peak": round(PEAK_X3, 1), "frac": round(ach / PEAK_X3, 4), "avg_launch_us": round(us, 2),
                "peak_note": "fp32-equivalent FLOP/s; peak = bf16 dense MFMA peak / 6 (six bf16 products per fp32 product); the native "
                             "fp32 MFMA peak is %.1f" % PEAK[mode],
                "algorithmic_bytes_per_launch": SAMPLES * (BATCH * DIMS[1] * 6 + DIMS[2] * DIMS[1] * 6 + BATCH * DIMS[2] * 4),
                "mfma_flop_per_launch": 6 * flops,
                "layer_end_to_end": {"draw_us": round(us_draw, 2), "total_us": round(us + us_draw, 2)}})
    return out


def draw_roofline(net, dev):
    """The step's draw launch: every posterior tensor of the MLP x 8 MC samples in ONE k_draw_multi (bf16 mode).
    Algorithmic bytes: 8 per posterior scalar read (mu, rho) + 2 per drawn weight written.  The kernel is VALU-bound,
    not HBM-bound: one Philox4x32-10 block + two Box-Muller pairs are ~74 VALU instructions per 4 weights, of which the
    20 v_mad_u64_u32 and the 8 transcendentals issue at quarter rate: ~21 us of VALU issue on 1024 SIMDs."""
    from bayesianneuralnetworks_amd import ops
    from bayesianneuralnetworks_amd._rng import DrawKey, generator_for
    g = generator_for("bf16")
    specs = []
    for i, L in enumerate(m for m in net.layers if hasattr(m, "weight")):
        specs.append((L.weight.mean.detach(), L.weight.scale.detach(), L.bias.mean.detach(), L.bias.scale.detach(),
                      DrawKey(1, 2 * i + 1, 0, SAMPLES, 0, gen=g), DrawKey(1, 2 * i + 2, 0, SAMPLES, 0, gen=g)))
    us = _graph_time(lambda: ops.draw_layers(specs, SAMPLES), dev)
    nbytes = 8.0 * P_SCALARS + 2.0 * SAMPLES * P_SCALARS
    gbs = nbytes / us / 1e3
    traffic, src = pmc_traffic("draw_multi")
    return {"kernel": "k_draw_multi: 6 posterior tensors (2.395 M scalars) x 8 MC samples, one launch", "bound": "hbm",
            "achieved": round(gbs, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(gbs / PEAK_HBM_GBS, 4),
            "traffic": traffic, "traffic_source": src, "avg_launch_us": round(us, 2), "algorithmic_bytes_per_launch": nbytes,
            "draws_per_launch": SAMPLES * P_SCALARS, "gdraws_per_s": round(SAMPLES * P_SCALARS / us / 1e3, 2)}


def conv_roofline(which, mode, dev, iters=20):
    """configs[2] / configs[3]: ONE NormalConv2d call over 8 MC samples (per-sample inputs), implicit GEMM.
    Algorithmic FLOP = 2 * S * B * OH * OW * O * C * KH * KW."""
    import bayesianneuralnetworks_amd as bnn
    from bayesianneuralnetworks_amd.nn import NormalConv2d
    from bayesianneuralnetworks_amd import _mc
    B, C, O, HW, k, s, p = {"lenet": (1024, 64, 64, 6, 3, 2, 1), "cifar": (256, 128, 128, 4, 3, 1, 1)}[which]
    torch.manual_seed(11)
    layer = NormalConv2d(C, O, k, stride=s, padding=p).to(dev)
    x = torch.randn(SAMPLES * B, C, HW, HW, device=dev)
    OH = (HW + 2 * p - k) // s + 1
    flops = 2.0 * SAMPLES * B * OH * OH * O * C * k * k
    prev = bnn.get_compute()
    bnn.set_compute(mode)
    with torch.no_grad(), _mc.McContext(SAMPLES, B, 0):
        ms = _graph_time(lambda: layer(x), dev) * 1e-3
    bnn.set_compute(prev)
    ach = flops / (ms * 1e-3) / 1e12
    tag = "conv_%s_%s" % (which, mode)
    traffic, src = pmc_traffic(tag)
    abytes = 4.0 * x.numel() + 4.0 * SAMPLES * B * O * OH * OH + 8.0 * (O * C * k * k + O)
    # which roof: the layer's arithmetic intensity against the machine balance (2.5 PFLOP/s / 8 TB/s = 312 FLOP per byte)
    intensity = flops / abytes
    gbs = abytes / (ms * 1e-3) / 1e9
    out = {"kernel": "NormalConv2d %d->%d k%d s%d p%d on %dx%d, batch %d x 8 samples, one layer call = draw launch + k_conv_bf16 (%s)"
                     % (C, O, k, s, p, HW, HW, B, tag),
           "flop_per_algorithmic_byte": round(intensity, 1), "traffic": traffic, "traffic_source": src,
           "avg_launch_us": round(ms * 1e3, 2), "algorithmic_flop_per_launch": flops, "algorithmic_bytes_per_launch": abytes,
           "tflops": round(ach, 2), "gbytes_per_s": round(gbs, 1)}
    if intensity < PEAK[mode] * 1e12 / (PEAK_HBM_GBS * 1e9):
        out.update({"bound": "hbm", "achieved": round(gbs, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(gbs / PEAK_HBM_GBS, 4)})
    else:
        out.update({"bound": "mfma", "achieved": round(ach, 2), "peak": PEAK[mode], "unit": "TFLOP/s", "frac": round(ach / PEAK[mode], 4)})
    return out


def wide_roofline(dev, iters=5):
    """configs[4] layer: NormalLinear(4096, 4096) at batch 4096 in the fp32 parity mode, ONE MC sample per call
    (the stack is 8 such layers; 1.0995 TFLOP per sample in all).  fp32-accurate contraction on bf16x3 splits."""
    from bayesianneuralnetworks_amd.nn import NormalLinear
    from bayesianneuralnetworks_amd import _mc
    M = N = K = 4096
    torch.manual_seed(12)
    layer = NormalLinear(K, N).to(dev)
    layer.compute = "f32"
    x = torch.randn(M, K, device=dev)
    with torch.no_grad(), _mc.McContext(1, M, 0):
        ms = _time_launches(lambda: layer(x), dev, iters, warm=2)
    flops = 2.0 * M * N * K
    ach = flops / (ms * 1e-3) / 1e12
    traffic, src = pmc_traffic("wide_f32")
    return {"kernel": "NormalLinear 4096x4096, batch 4096, fp32 mode, 1 MC sample per call (draw as three bf16 planes + split of the "
                      "input + k_dense_bf16<4,8,4,1,3> on three-plane operands)",
            "bound": "mfma", "achieved": round(ach, 2), "peak": round(PEAK_X3, 1), "unit": "TFLOP/s",
            "frac": round(ach / PEAK_X3, 4),
            "peak_note": "fp32-equivalent FLOP/s of the whole layer call; peak = bf16 dense MFMA peak / 6; native fp32 MFMA peak %.1f" % PEAK["f32"],
            "traffic": traffic, "traffic_source": src,
            "avg_launch_us": round(ms * 1e3, 1), "algorithmic_flop_per_launch": flops,
            "algorithmic_bytes_per_launch": 8.0 * (N * K + N) + 4.0 * M * (K + N)}


def lenet_net_roofline(mode, dev):
    """configs[2] as the NETWORK the reference ships (examples/MNIST/model.py:20-33, the same architecture the FashionMNIST
    config names with Normal* layers): three stock Conv2d (+ BatchNorm, ELU) -- run ONCE per forward, they see no weight
    noise -- then NormalConv2d(64, 64, 3, s2, p1), ELU, NormalLinear(576, 10), Softmax over 8 MC samples in one batched pass
    (mc_batched), eval mode, batch 1024 (examples/FashionMNIST/train.py:16).  One captured graph per forward.  Random-init
    weights of that architecture, synthetic 28 x 28 inputs.  FLOPs: the Bayesian layers x 8 samples + the prefix once."""
    import bayesianneuralnetworks_amd as bnn
    from bayesianneuralnetworks_amd.nn import BayesianNetworkModule, NormalConv2d, NormalLinear
    from torch.nn import Conv2d, BatchNorm2d, ELU, Softmax, Flatten, Sequential
    B = 1024

    class BCNN(BayesianNetworkModule):
        def __init__(self):
            super().__init__(1, 10, SAMPLES)
            self.layers = Sequential(Conv2d(1, 32, 5, padding=2, stride=2), BatchNorm2d(32), ELU(), Conv2d(32, 32, 3, padding=1, stride=1), ELU(),
                                     Conv2d(32, 64, 3, padding=0, stride=2), ELU(), NormalConv2d(64, 64, 3, padding=1, stride=2), ELU(), Flatten(),
                                     NormalLinear(576, 10), Softmax(dim=-1))

        def _forward(self, x):
            return self.layers(x)

    torch.manual_seed(13)
    net = BCNN().to(dev).eval()
    net.mc_batched = True
    x = torch.randn(B, 1, 28, 28, device=dev)
    prev = bnn.get_compute()
    bnn.set_compute(mode)
    # the stock prefix is MIOpen's: let it search its solvers once (torch.backends.cudnn.benchmark) -- without the search a fresh
    # box may fall back to a per-image im2col + GEMM convolution (6144 launches per forward) and the leg measures that instead
    prev_bm = torch.backends.cudnn.benchmark
    torch.backends.cudnn.benchmark = True
    try:
        with torch.no_grad():
            us = _graph_time(lambda: net.forward_stacked(x, SAMPLES), dev, reps=4, iters=10)
            us_prefix = _graph_time(lambda: net.layers[:7](x), dev, reps=4, iters=10)
    finally:
        bnn.set_compute(prev)
        torch.backends.cudnn.benchmark = prev_bm
    f_bayes = SAMPLES * (2.0 * B * 9 * 64 * 576 + 2.0 * B * 576 * 10)
    f_prefix = 2.0 * B * (196 * 32 * 25 + 196 * 32 * 288 + 36 * 64 * 288)
    ach = (f_bayes + f_prefix) / us / 1e6
    return {"kernel": "examples/MNIST/model.py BCNN, batch 1024, 8 MC samples per forward (mc_batched, eval): stock conv prefix once + "
                      "NormalConv2d 64->64 k3 s2 p1 + ELU + NormalLinear 576->10 + softmax on all samples",
            "mc_samples_per_s": round(SAMPLES / us * 1e6, 1), "avg_forward_us": round(us, 1), "prefix_us": round(us_prefix, 1),
            "bayesian_part_us": round(us - us_prefix, 1),
            "bound": "mfma", "achieved": round(ach, 2), "peak": PEAK[mode], "unit": "TFLOP/s", "frac": round(ach / PEAK[mode], 5),
            "algorithmic_flop_per_forward": f_bayes + f_prefix, "bayesian_flop_per_forward": f_bayes,
            "bayesian_part_tflops": round(f_bayes / max(us - us_prefix, 1e-3) / 1e6, 2), "traffic": None,
            "note": "the stock prefix (MIOpen convolutions on 1-32-64 channels, torch.backends.cudnn.benchmark = True) is torch's; the Bayesian layers are this engine's"}


def wide_stack_roofline(dev):
    """configs[4] as the STACK it names: 8 x NormalLinear(4096, 4096) with ReLU between, batch 4096, fp32 parity mode, ONE MC
    sample per forward -- 1.0995 TFLOP per MC sample.  mc_batched pass with nn.fuse_activations: ONE draw launch for all
    eight layers (three bf16 planes each), the input split once, hidden activations handed on as three planes."""
    import bayesianneuralnetworks_amd as bnn
    from bayesianneuralnetworks_amd.nn import BayesianNetworkModule, NormalLinear, fuse_activations
    D, B, L = 4096, 4096, 8

    class Stack(BayesianNetworkModule):
        def __init__(self):
            super().__init__(D, D, 1)
            mods = []
            for i in range(L):
                mods.append(NormalLinear(D, D))
                if i < L - 1:
                    mods.append(torch.nn.ReLU())
            self.layers = torch.nn.Sequential(*mods)

        def _forward(self, x):
            return self.layers(x)

    torch.manual_seed(14)
    net = Stack().to(dev)
    net.mc_batched = True
    fuse_activations(net, bf16_activations=True)
    x = torch.randn(B, D, device=dev)
    prev = bnn.get_compute()
    bnn.set_compute("f32")
    try:
        with torch.no_grad():
            ms = _time_launches(lambda: net._forward_batched_stacked(x, 1, 0), dev, 4, warm=2)
    finally:
        bnn.set_compute(prev)
    flops = 2.0 * B * L * D * D
    ach = flops / (ms * 1e-3) / 1e12
    return {"kernel": "8 x NormalLinear(4096, 4096) + ReLU, batch 4096, fp32 parity mode, 1 MC sample per forward: draw launches (three bf16 "
                      "planes) + one split of the input + 8 x k_dense_bf16<4,8,4,1,3> on three-plane operands",
            "mc_samples_per_s": round(1e3 / ms, 2), "avg_forward_ms": round(ms, 3),
            "bound": "mfma", "achieved": round(ach, 2), "peak": round(PEAK_X3, 1), "unit": "TFLOP/s", "frac": round(ach / PEAK_X3, 4),
            "peak_note": "fp32-equivalent FLOP/s of the whole forward; peak = bf16 dense MFMA peak / 6; native fp32 MFMA peak %.1f" % PEAK["f32"],
            "algorithmic_flop_per_forward": flops, "algorithmic_bytes_per_forward": 8.0 * L * (D * D + D) + 4.0 * B * D * 2, "traffic": None}


def sampler_roofline(dev, iters=20):
    """K1 alone on a working set beyond the Infinity Cache (64 Mi scalars = 768 MiB of traffic):
    HBM-bound, 8 B read + 4 B written per scalar."""
    from bayesianneuralnetworks_amd import ops
    from bayesianneuralnetworks_amd._rng import DrawKey
    n = 64 << 20
    mu = torch.zeros(n, device=dev)
    rho = torch.full((n,), -2.0, device=dev)
    key = DrawKey(1, 1, 0, 1, 0)
    ms = _time_launches(lambda: ops._sample_affine_philox_raw(mu, rho, key), dev, iters, warm=3)
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
        ms = _time_launches(lambda: ops.kl_normal([mu], [rho], [(0.0, 0.1)], 1.0, out=out), dev, iters, warm=3)
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


# ------------------------------------------------------------------------------------------ rank spawn
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def spawn_ranks(n, argv):
    """`bench.py --gpus N` outside a launcher: start N fresh rank processes of this script (one per GPU, RANK /
    LOCAL_RANK / WORLD_SIZE / MASTER_* in their environment) and relay rank 0's JSON line.  Called BEFORE this
    process has made any GPU call: the parent never initialises the GPU and nothing is re-exec'ed."""
    port = _free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ)
        env.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(n),
                    "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port),
                    "HSA_ENABLE_IPC_MODE_LEGACY": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")})
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    out, _ = procs[0].communicate()
    rc = procs[0].returncode
    for p in procs[1:]:
        rc = p.wait() or rc
    sys.stdout.write(out.decode())
    sys.stdout.flush()
    return rc


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--mode", default="forward", choices=["forward", "train"])
    ap.add_argument("--scaling", default="strong", choices=["strong", "weak"],
                    help="N > 1 headline: strong = the 8 global MC samples sharded 8/N per GPU (SURVEY 8e); weak = 8 per GPU")
    ap.add_argument("--inflight", type=int, default=int(os.environ.get("BNN_BENCH_INFLIGHT", "4")),
                    help="N = 1 forward: independent steps in flight on separate streams (1 = one stream)")
    ap.add_argument("--windows", type=int, default=int(os.environ.get("BNN_BENCH_WINDOWS", "25")),
                    help="repetitions of the timed window of --steps steps inside this run; the line reports the median window")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-legs", action="store_true", help="skip the configs[2..4] / K1 / K3 roofline legs")
    argv = sys.argv[1:] if argv is None else list(argv)
    args = ap.parse_args(argv)

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return spawn_ranks(args.gpus, argv)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("BNN_BENCH_DRYRUN"):
        # spawn rehearsal for the CPU test-suite: no GPU call, no process group
        print(json.dumps({"dryrun": True, "rank": rank, "world": world, "gpus": args.gpus}), flush=True)
        return 0
    ndev = torch.cuda.device_count()        # (counting devices does not initialise the GPU)
    rehearsal = False
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # "nccl" = RCCL over xGMI.  With fewer GPUs than ranks (or BNN_BENCH_BACKEND=gloo) the N > 1 code path is only
        # REHEARSED over gloo, ranks sharing a card: the number is then not a result and the line says so.
        backend = os.environ.get("BNN_BENCH_BACKEND", "nccl" if ndev >= world else "gloo")
        rehearsal = backend != "nccl"
        torch.distributed.init_process_group(backend)
    local = local % max(1, ndev)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    import bayesianneuralnetworks_amd as bnn
    from bayesianneuralnetworks_amd import _lib, distributed as bd
    _lib.load()                                     # fail loudly if the HIP library is missing
    post = posteriors(0)
    net = build_net(dev, post)
    x_cpu = torch.randn(BATCH, DIMS[0], generator=torch.Generator().manual_seed(1))
    x = x_cpu.to(dev)
    bnn.manual_seed(2)

    strong = world > 1 and args.scaling == "strong" and SAMPLES % world == 0
    results, checked, weak, pipeline, spread = {}, {}, None, None, {}
    failures = []

    def summarise(dts, total, steps):
        """median / fastest / slowest window -> (value, ms_per_step, steps), {window statistics}"""
        med, lo, hi = _median(dts), min(dts), max(dts)
        return (total * steps / med, med / steps * 1e3, steps), {
            "windows": len(dts), "steps_per_window": steps, "ms_per_step_median": round(med / steps * 1e3, 4),
            "ms_per_step_min": round(lo / steps * 1e3, 4), "ms_per_step_max": round(hi / steps * 1e3, 4)}

    for mode in ([args.dtype] + (["f32"] if (args.dtype != "f32" and world == 1) else [])):
        bnn.set_compute(mode)
        # bf16 mode: the synthetic batch is resident in HBM as bf16 (the first layer would round its
        # A operand to bf16 anyway -- identical results, half the input stream); fp32 mode: fp32.
        x_in = resident_input(x, mode)
        use_pipe = args.inflight > 1 and not args.no_graph and args.mode == "forward" and mode == args.dtype
        skw = {}
        if strong:
            s0, cnt = bd.shard_samples(SAMPLES, rank, world)
            skw = dict(samples=cnt, sample0=s0, total_samples=SAMPLES)
        step = Step(net, x_in, rank, world, not args.no_graph, **skw)
        steps = args.steps if mode == args.dtype else max(10, args.steps // 4)
        total = SAMPLES if strong else world * SAMPLES
        windows = args.windows if mode == args.dtype else max(3, args.windows // 5)
        results[mode], spread[mode] = summarise(time_windows(step, steps, args.warmup, world, dev, windows), total, steps)
        # one more replay of the TIMED object against the oracle -- both modes, every N (N > 1: the all-reduced buffer, global
        # sample ids; every rank replays, rank 0 compares)
        chk = oracle_check(step, post, x_cpu, mode, evaluate=(rank == 0))
        if rank == 0:
            checked[mode] = chk
            if not chk["ok"]:
                failures.append("oracle check of the timed step failed (%s mode)" % mode)
        if use_pipe:
            # throughput pipeline: `inflight` steps on as many streams; the single-stream number above becomes the latency
            single, single_spread = results[mode], spread[mode]
            pipe = PipelinedSteps(net, x_in, args.inflight, rank, world, **skw)
            piped, pipe_spread = summarise(time_windows(pipe, steps, args.warmup, world, dev, windows), total, steps)
            # both are windows of K timed steps between barriers (max over ranks, so every rank decides alike): the line reports
            # the higher throughput and says which arrangement it was
            use_piped = piped[0] >= single[0]
            results[mode], spread[mode] = (piped, pipe_spread) if use_piped else (single, single_spread)
            pipeline = {"steps_in_flight": args.inflight, "untimed_preroll_steps": int(os.environ.get("BNN_BENCH_PREROLL", "256")),
                        "reported": "pipelined" if use_piped else "single_stream",
                        "pipelined_value": round(piped[0], 1), "pipelined_ms_per_step": round(piped[1], 4),
                        "pipelined_windows": pipe_spread,
                        "single_stream_value": round(single[0], 1),
                        "single_stream_ms_per_step": round(single[1], 4),
                        "single_stream_windows": single_spread,
                        "note": "pipelined = `steps_in_flight` complete steps in flight on as many streams: THROUGHPUT (K complete "
                                "steps / wall time); one step's LATENCY is single_stream_ms_per_step"}
            if world == 1:
                # what the steps' LAST replays -- executed while the others were in flight -- left behind, against the oracle
                pchks = [oracle_check(st, post, x_cpu, mode, replay=False) for st in pipe.steps]
                pipeline.update({"checked_in_flight_results_ok": all(c["ok"] for c in pchks),
                                 "in_flight_pred_max_err": max(c["pred_max_err"] for c in pchks),
                                 "in_flight_epochs": [c["epoch_dev"] for c in pchks]})
                if not pipeline["checked_in_flight_results_ok"]:
                    failures.append("oracle check of the in-flight results failed")
            del pipe
        if mode == args.dtype and strong:
            if use_pipe:
                wstep = PipelinedSteps(net, x_in, args.inflight, rank, world)
            else:
                wstep = Step(net, x_in, rank, world, not args.no_graph)
            wres, _ = summarise(time_windows(wstep, steps, args.warmup, world, dev, max(3, windows // 5)), world * SAMPLES, steps)
            weak = (wres[0], wres[1])
            del wstep
        del step
    bnn.set_compute(args.dtype)
    # the device error word: a kernel whose bounded hand-off wait gave up has skipped its stores and said so there
    try:
        _lib.check_device(dev)
        device_word = "clear"
    except Exception as e:      # BnnHipError(BNN_E_DEVICE)
        device_word = "SET: %s" % e
        failures.append("device error word set after the timed runs")

    # training step (own copy of the model: Adam moves the parameters)
    train = None
    if args.mode == "train" or world == 1:
        tnet = build_net(dev, post)
        tsteps = args.steps if args.mode == "train" else max(10, args.steps // 4)
        tstep = TrainStep(tnet, x.bfloat16() if args.dtype == "bf16" else x, rank, world, not args.no_graph)   # (dense rows: autograd saves x)
        dt = time_steps(tstep, tsteps, args.warmup, world, dev)
        train = (world * SAMPLES * tsteps / dt, dt / tsteps * 1e3, tsteps, float(tstep.loss))
        del tstep, tnet

    if rank == 0:
        val, ms, steps = results[args.dtype]
        per_gpu = SAMPLES // world if strong else SAMPLES
        line = {
            "metric": "MC-samples/sec (node), 784-1200-1200-10 BayesianLinear MLP, batch 512",
            "value": round(val, 1), "unit": "MC-samples/s", "n_gpus": world, "steps": steps,
            "warmup": args.warmup, "ms_per_step": round(ms, 4), "higher_is_better": True,
            "scaling": "strong" if strong else "weak", "vs_baseline": None,
            "dtype": "bf16" if args.dtype == "bf16" else "f32", "data": "synthetic",
            "config": {"workload": ("configs[1]: MNIST-shape 784-1200-1200-10 NormalLinear MLP, batch 512, "
                                    "%d MC samples per forward per GPU (%d in all), KL once per forward, predictive mean; "
                                    % (per_gpu, SAMPLES if strong else world * SAMPLES)) +
                                   ("bf16 operands / fp32 accumulate" if args.dtype == "bf16" else
                                    "fp32 parity mode (wide layers: bf16x3 splits on the bf16 MFMA, fp32-accurate)"),
                       "samples_per_step_per_gpu": per_gpu, "batch": BATCH, "hip_graph": not args.no_graph,
                       "collective": "one all-reduce of [6 KL sums, KL scalar, 512x10 prediction sum] fp32" if world > 1 else None},
        }
        if rehearsal:
            line["rehearsal"] = "ranks share %d GPU(s) over gloo: code-path rehearsal, not a result" % ndev
        if weak is not None:
            line["weak"] = {"value": round(weak[0], 1), "ms_per_step": round(weak[1], 4), "samples_per_step_per_gpu": SAMPLES,
                            "note": "weak scaling: 8 MC samples on every GPU (global predictive mean over 8 N)"}
        line["timing"] = dict(spread[args.dtype], note="value / ms_per_step = the MEDIAN of `windows` timed windows of `steps` steps each "
                              "(barrier + synchronize around every window)")
        line["device_error_word"] = device_word
        if args.dtype in checked:
            line["checked"] = checked[args.dtype]
        if "f32" in checked and args.dtype != "f32":
            line["checked_f32"] = checked["f32"]
        if pipeline is not None:
            line["config"]["pipeline"] = pipeline
        if train is not None:
            line["train"] = {"value": round(train[0], 1), "unit": "MC-samples/s", "ms_per_step": round(train[1], 4),
                             "steps": train[2], "loss": round(train[3], 4), "hip_graph": (not args.no_graph) and world == 1,
                             "note": "training-loop body of examples/MNIST/train.py:53-65 (fwd + KL + CE + HIP bwd + Adam)"}
            if args.mode == "train":
                line["forward"] = {"value": line["value"], "ms_per_step": line["ms_per_step"]}
                line["metric"] = "MC-samples/sec (node), TRAINING step, 784-1200-1200-10 BayesianLinear MLP, batch 512"
                line["value"], line["ms_per_step"], line["steps"] = line["train"]["value"], line["train"]["ms_per_step"], train[2]
                line["scaling"] = "weak"
        if "f32" in results and args.dtype != "f32":
            line["f32"] = {"value": round(results["f32"][0], 1), "ms_per_step": round(results["f32"][1], 4),
                           "note": "same step in the 1e-5 parity mode: every operand as three bf16 planes on the dense kernel (six plane-pair k-steps per k-block), fp32 accumulate"}
        line["roofline"] = kernel_roofline(net, x, args.dtype, dev)
        if args.dtype == "bf16":
            line["roofline_draw"] = draw_roofline(net, dev)
        if world == 1 and not args.no_legs:
            line["roofline_sampler"] = sampler_roofline(dev)
            line["roofline_kl"] = kl_roofline(dev)
            line["roofline_conv_lenet"] = conv_roofline("lenet", args.dtype, dev)
            line["roofline_conv_cifar"] = conv_roofline("cifar", args.dtype, dev)
            line["roofline_wide_f32"] = wide_roofline(dev)
            line["roofline_lenet_net"] = lenet_net_roofline(args.dtype, dev)
            line["roofline_wide_stack"] = wide_stack_roofline(dev)
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(post, x_cpu)
        if failures:
            line["failed"] = failures       # a wrong-result run publishes no headline: the process exits non-zero
        print(json.dumps(line), flush=True)
    rc = 1 if failures else 0
    if world > 1:
        t = torch.tensor([rc], device=dev)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)      # every rank leaves with rank 0's verdict
        rc = int(t.item())
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()
    return rc


if __name__ == "__main__":
    sys.exit(main())
