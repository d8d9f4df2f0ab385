"""The path bench.py TIMES, pinned to the oracle.

bench.Step for configs[1] -- 784-1200-1200-10 NormalLinear MLP, batch 512, 8 MC samples, bf16 operands with
bf16 hidden activations, KL in two halves carried by the step's own launches, the device-epoch bump, ONE
captured HIP graph -- is replayed three times.  After replay k the KL scalar, 64 rows of the predictive mean
and 64 rows of the layer-2 output (a forward hook's copy, captured with the graph) are compared with the
CPU oracle fed what the kernels feed the MFMA (bf16-rounded inputs / drawn weights / hidden activations) on
the draw keys of THAT replay (epoch_dev = k).  Tolerances are stated in bench.oracle_check.

Also here (CPU): `bench.py --gpus N` really starts N ranks.
"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_gpus_flag_spawns_that_many_ranks():
    """`bench.py --gpus 2` with no launcher: two rank processes (the dry-run rehearsal makes no GPU call)."""
    env = dict(os.environ, BNN_BENCH_DRYRUN="1")
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    out = subprocess.check_output([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3"],
                                  env=env, timeout=300)
    line = json.loads(out.decode().strip().splitlines()[-1])
    assert line == {"dryrun": True, "rank": 0, "world": 2, "gpus": 2}
    # under a launcher (WORLD_SIZE set) the script is one rank and spawns nothing
    env2 = dict(env, WORLD_SIZE="4", RANK="3", LOCAL_RANK="3")
    out = subprocess.check_output([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4"], env=env2, timeout=300)
    assert json.loads(out.decode().strip().splitlines()[-1]) == {"dryrun": True, "rank": 3, "world": 4, "gpus": 4}


def test_bench_strong_scaling_shards_cover_the_samples():
    import bench
    from bayesianneuralnetworks_amd import distributed as bd
    for world in (1, 2, 4, 8):
        ids = []
        for r in range(world):
            s0, cnt = bd.shard_samples(bench.SAMPLES, r, world)
            ids += list(range(s0, s0 + cnt))
        assert ids == list(range(bench.SAMPLES))


def test_multi_gpu_check_arithmetic_on_an_emulated_all_reduce():
    """bench.oracle_evaluate for N > 1, on CPU: the packed buffers of two ranks of a strong-scaling run are built from the
    oracle itself -- rank r holds global MC samples [4 r, 4 r + 4) scaled by 1 / 8 and the KL SUMS of its shard of every
    posterior tensor (distributed.shard_range) -- and added as the one all-reduce adds them.  The check must accept that
    buffer (KL scalar from the six sums, predictive mean over global sample ids 0..7) and reject a buffer with one rank's
    samples drawn on the wrong ids, or with one tensor's KL shard missing."""
    sys.path.insert(0, ROOT)
    import bench
    from bayesianneuralnetworks_amd import distributed as bd
    from bayesianneuralnetworks_amd._rng import DrawKey, GEN_PHILOX7_U16
    from oracle import oracle as orc
    post = bench.posteriors(0)
    x_cpu = torch.randn(bench.BATCH, bench.DIMS[0], generator=torch.Generator().manual_seed(1))
    rows, world, S, e_dev = 4, 2, bench.SAMPLES, 3
    T = 2 * len(post)
    keys = [(DrawKey(2, 2 * i + 1, 0, S // world, 11, gen=GEN_PHILOX7_U16), DrawKey(2, 2 * i + 2, 0, S // world, 11, gen=GEN_PHILOX7_U16))
            for i in range(len(post))]                           # rank 0's keys: sample0 = 0, 4 samples

    def rank_buffer(r, id_shift=0, drop_kl=None):
        buf = np.zeros(T + 1 + bench.BATCH * bench.DIMS[-1], np.float64)
        t = 0
        for mw, rw, mb, rb in post:
            for mu, rho in ((mw, rw), (mb, rb)):
                lo, hi = bd.shard_range(mu.numel(), r, world)
                if hi > lo and t != drop_kl:
                    buf[t] = orc.kl_sum(mu.reshape(-1)[lo:hi].numpy(), rho.reshape(-1)[lo:hi].numpy(), 0.0, 0.1)
                t += 1
        pred = np.zeros((bench.BATCH, bench.DIMS[-1]))
        s0, cnt = bd.shard_samples(S, r, world)
        for s in range(s0 + id_shift, s0 + id_shift + cnt):
            h = orc.bf16_round(x_cpu[:rows].numpy())
            for li, ((mw, rw, mb, rb), (kw, kb)) in enumerate(zip(post, keys)):
                w = orc.bf16_round(orc.sample_affine(mw.numpy(), rw.numpy(), orc.eps_fill(kw.seed, kw.stream, s, kw.epoch_host, e_dev, tuple(mw.shape), kw.gen)))
                b = orc.sample_affine(mb.numpy(), rb.numpy(), orc.eps_fill(kb.seed, kb.stream, s, kb.epoch_host, e_dev, tuple(mb.shape), kb.gen))
                h = orc.linear(h, w, b)
                if li < len(post) - 1:
                    h = orc.bf16_round(np.maximum(h, 0.0))
            pred[:rows] += h / S
        buf[T + 1:] = pred.reshape(-1)
        return buf

    r0, r1 = rank_buffer(0), rank_buffer(1)
    good = bench.oracle_evaluate((r0 + r1).astype(np.float32), keys, e_dev, e_dev + 1, post, x_cpu, "bf16", S, world, rows=rows)
    assert good["ok"] and good["world"] == 2 and good["samples"] == 8, good
    assert good["kl_rel_err"] < 1e-6 and good["pred_max_err"] < 1e-3
    bad = bench.oracle_evaluate((r0 + rank_buffer(1, id_shift=1)).astype(np.float32), keys, e_dev, e_dev + 1, post, x_cpu, "bf16", S, world, rows=rows)
    assert not bad["ok"] and bad["kl_rel_err"] < 1e-6            # wrong sample ids on rank 1: the predictive mean is off, the KL is not
    bad = bench.oracle_evaluate((r0 + rank_buffer(1, drop_kl=2)).astype(np.float32), keys, e_dev, e_dev + 1, post, x_cpu, "bf16", S, world, rows=rows)
    assert not bad["ok"] and bad["kl_rel_err"] > 1e-3
    stale = bench.oracle_evaluate((r0 + r1).astype(np.float32), keys, e_dev, e_dev, post, x_cpu, "bf16", S, world, rows=rows)
    assert not stale["ok"]                                       # the replay did not advance the device epoch


@pytest.mark.gpu
@pytest.mark.parametrize("mode,fuse_head", [("bf16", False), ("bf16", True), ("f32", False), ("f32", True)])
def test_timed_graph_replays_match_oracle(mode, fuse_head):
    sys.path.insert(0, ROOT)
    import bench
    import bayesianneuralnetworks_amd as bnn
    from bayesianneuralnetworks_amd import _lib
    lib = _lib.load()
    dev = torch.device("cuda:0")
    post = bench.posteriors(0)
    net = bench.build_net(dev, post)
    x_cpu = torch.randn(bench.BATCH, bench.DIMS[0], generator=torch.Generator().manual_seed(1))
    x = x_cpu.to(dev)
    bnn.manual_seed(2)
    bnn.set_compute(mode)
    rows = 64
    tap = torch.zeros(bench.SAMPLES, rows, bench.DIMS[2], device=dev)

    def hook(_m, _i, out):
        from bayesianneuralnetworks_amd import ops as _ops
        if isinstance(out, _ops.HeadPartials):      # layer 2 fused with the head: its output does not exist
            return
        if not torch.is_tensor(out):                # fp32 mode: a hidden activation travels as three bf16 planes (ops.X3Activation)
            out = out.float()
        tap.copy_(out.reshape(bench.SAMPLES, bench.BATCH, -1)[:, :rows])

    h = net.layers[2].register_forward_hook(hook)
    try:
        n0 = lib.bnn_launch_count()
        # fuse_head: layer 2 and the head are ONE launch and layer 2's output is never stored -- nothing to tap; the
        # predictive mean (through all three layers) and the KL are checked all the same, and the step is 4 launches, not 5
        step = bench.Step(net, bench.resident_input(x, mode), 0, 1, True, fuse_head=fuse_head)
        assert step.graph is not None and lib.bnn_launch_count() > n0
        n1 = lib.bnn_launch_count()
        step._body()
        torch.cuda.synchronize()
        # draw, layer 1, layer 2 (+ head | , head), reduction -- in the fp32 mode too: the split of the input into planes rides
        # in the draw launch (kind 3)
        assert lib.bnn_launch_count() - n1 == (4 if fuse_head else 5)
        prev = None
        for k in range(3):
            res = bench.oracle_check(step, post, x_cpu, mode, rows=rows, tap=None if fuse_head else tap)
            print("replay %d: %s" % (k, json.dumps(res)))
            out_dir = os.path.join(ROOT, "gpurun_out")
            if os.path.isdir(out_dir):              # measured errors next to the tolerances, for the record
                with open(os.path.join(out_dir, "bench_path_check.jsonl"), "a") as f:
                    f.write(json.dumps(dict(res, replay=k)) + "\n")
            assert res["epoch_advanced"]
            assert res["kl_rel_err"] <= 1e-5, res
            assert res["pred_max_err"] <= res["pred_tol_abs"], res
            assert res["ok"], res
            cur = step.packed.detach().cpu().numpy().copy()
            if prev is not None:
                # fresh noise on every replay: the predictions differ, the KL (eps-independent) does not
                assert np.abs(cur[step.T + 1:] - prev[step.T + 1:]).max() > 1e-3
                assert cur[step.T] == prev[step.T]
            prev = cur
    finally:
        h.remove()
        bnn.set_compute("f32")
        step.gen.epoch_dev(dev).zero_()         # later tests address draws with epoch_dev = 0


@pytest.mark.gpu
def test_pipelined_steps_in_flight_match_oracle():
    """bench.PipelinedSteps: independent captured steps replayed round-robin on separate streams (what the N = 1 headline
    times).  After 7 replays the result each step's LAST replay left behind -- computed while the other step was in flight
    -- is checked against the oracle on that replay's epoch; the steps own their epoch words, so they never draw the same
    noise."""
    sys.path.insert(0, ROOT)
    import bench
    import bayesianneuralnetworks_amd as bnn
    dev = torch.device("cuda:0")
    post = bench.posteriors(0)
    net = bench.build_net(dev, post)
    x_cpu = torch.randn(bench.BATCH, bench.DIMS[0], generator=torch.Generator().manual_seed(1))
    bnn.manual_seed(2)
    bnn.set_compute("bf16")
    try:
        pipe = bench.PipelinedSteps(net, bench.resident_input(x_cpu.to(dev), "bf16"), 2)
        for _ in range(7):
            pipe.run()
        pipe.finish()
        torch.cuda.synchronize()
        outs = []
        for st in pipe.steps:
            res = bench.oracle_check(st, post, x_cpu, "bf16", replay=False)
            print(json.dumps(res))
            assert res["ok"] and res["kl_rel_err"] <= 1e-5, res
            outs.append(st.packed.detach().cpu().numpy().copy())
        # 7 replays dealt 4 + 3, on top of the two eager warm-up runs each capture makes and the pipeline's untimed pre-roll
        pre = int(os.environ.get("BNN_BENCH_PREROLL", "256")) // 2
        assert [int(st.cell[0].item()) for st in pipe.steps] == [2 + pre + 4, 2 + pre + 3]
        assert np.abs(outs[0][pipe.steps[0].T + 1:] - outs[1][pipe.steps[1].T + 1:]).max() > 1e-3
        # the device-wide epoch word was not touched by the private steps
        from bayesianneuralnetworks_amd._rng import default_generator
        assert int(default_generator.epoch_dev(dev)[0].item()) == 0
    finally:
        bnn.set_compute("f32")


@pytest.mark.gpu
def test_strong_scaling_shards_reproduce_the_single_gpu_step():
    """What `bench.py --gpus G` runs on rank r (S / G samples, global ids from r S / G; KL slice r of G) -- executed here rank
    after rank on one GPU, the all-reduce replaced by a sum -- gives the single-GPU 8-sample step: the eps stream is addressed
    by the GLOBAL sample id, so the union of the ranks' draws does not depend on G.  Covers the 4-, 2- and 1-sample launches
    of the draw / dense kernels that the driver's 2-, 4- and 8-GPU runs use."""
    sys.path.insert(0, ROOT)
    import bench
    import bayesianneuralnetworks_amd as bnn
    from bayesianneuralnetworks_amd import distributed as bd
    dev = torch.device("cuda:0")
    post = bench.posteriors(0)
    net = bench.build_net(dev, post)
    x = bench.resident_input(torch.randn(bench.BATCH, bench.DIMS[0], generator=torch.Generator().manual_seed(1)).to(dev), "bf16")
    bnn.set_compute("bf16")
    try:
        bnn.manual_seed(2)
        full = bench.Step(net, x, 0, 1, True)
        ref = full.packed.detach().cpu().numpy().copy()           # result of the capture's own first replay (epoch_dev = 2)
        T = full.T
        for G in (2, 4, 8):
            acc = np.zeros_like(ref)
            for r in range(G):
                s0, cnt = bd.shard_samples(bench.SAMPLES, r, G)
                bnn.manual_seed(2)                                 # same host epochs -> the same draw keys as `full`
                st = bench.Step(net, x, r, G, True, samples=cnt, sample0=s0, total_samples=bench.SAMPLES)
                torch.cuda.synchronize()
                acc += st.packed.detach().cpu().numpy()
                del st
            # predictions: sum over ranks of (1 / 8) sum over local samples == the 8-sample mean (fp32 sums in another order)
            assert np.abs(acc[T + 1:] - ref[T + 1:]).max() <= 1e-4 * max(1.0, float(np.abs(ref[T + 1:]).max())), G
            # KL: per-tensor SUMS add up over the ranks' slices
            assert np.allclose(acc[:T], ref[:T], rtol=1e-5), G
    finally:
        bnn.set_compute("f32")
        from bayesianneuralnetworks_amd._rng import default_generator
        default_generator.epoch_dev(dev).zero_()
