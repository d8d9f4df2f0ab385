"""world_size-2 gloo test (CPU) of the N > 1 path: sample sharding + the ONE packed all-reduce
reproduce the single-process KL and predictive mean."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from bayesianneuralnetworks_amd import distributed as bd


def test_shard_arithmetic():
    assert bd.shard_samples(8, 3, 4) == (6, 2)
    with pytest.raises(ValueError):
        bd.shard_samples(10, 0, 4)
    cover = []
    for r in range(3):
        lo, hi = bd.shard_range(1201, r, 3)
        assert lo % 4 == 0
        cover += list(range(lo, hi))
    assert cover == list(range(1201))
    assert bd.shard_range(10, 7, 8) == (10, 10)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _build():
    from bayesianneuralnetworks_amd.nn import NormalLinear, BayesianNetworkModule

    class Net(BayesianNetworkModule):
        def __init__(self):
            super().__init__(12, 5, samples=4)
            self.layers = torch.nn.Sequential(NormalLinear(12, 16), torch.nn.ReLU(), NormalLinear(16, 5))

        def _forward(self, x):
            return self.layers(x)

        def forward_stacked(self, x, samples=None, sample0=0):
            # CPU tensors: make the draw a function of the GLOBAL sample id, like the Philox
            # stream does on the GPU, so that the union over ranks is rank-count independent.
            outs = []
            for s in range(sample0, sample0 + samples):
                g = torch.Generator().manual_seed(1000 + s)
                h = x
                for m in self.layers:
                    if isinstance(m, NormalLinear):
                        m.weight.sample_with_eps(torch.randn(m.weight.shape, generator=g))
                        m.bias.sample_with_eps(torch.randn(m.bias.shape, generator=g))
                        h = m(h, sample=False)
                    else:
                        h = m(h)
                outs.append(h)
            return torch.stack(outs)

    torch.manual_seed(7)
    return Net()


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    net = _build()
    x = torch.randn(6, 12, generator=torch.Generator().manual_seed(3))
    kl_tensors = [(p, (0.0, 0.1)) for L in (net.layers[0], net.layers[2]) for p in (L.weight, L.bias)]
    with torch.no_grad():
        kl, pred, ys = bd.forward_sharded(net, x, 4, kl_tensors, n_batches=2.0)
    q.put((rank, float(kl), pred.numpy(), ys.shape[0]))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_match_single_process():
    from bayesianneuralnetworks_amd.nn import KLDivergence
    net = _build()
    x = torch.randn(6, 12, generator=torch.Generator().manual_seed(3))
    with torch.no_grad():
        ref_pred = net.forward_stacked(x, 4).mean(0).numpy()
        ref_kl = float(KLDivergence(number_of_batches=2)(net))
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(2)])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, kl, pred, local in res:
        assert local == 2                                   # 4 samples over 2 ranks
        assert abs(kl - ref_kl) <= 1e-5 * (1 + abs(ref_kl))
        assert np.allclose(pred, ref_pred, atol=1e-5, rtol=1e-5)


# ------------------------------------------------------------------ gradient exchange (training)
def _grad_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    torch.manual_seed(11)
    net = torch.nn.Sequential(torch.nn.Linear(7, 33), torch.nn.Tanh(), torch.nn.Linear(33, 5))
    red = bd.GradAllReducer(net.parameters(), bucket_bytes=100)        # 3 buckets at this size
    outs = []
    for step in range(2):                                               # views must survive a second step
        red.zero_grad()
        x = torch.randn(4, 7, generator=torch.Generator().manual_seed(100 * step + rank))
        net(x).pow(2).sum().backward()
        red.finish()
        outs.append([p.grad.clone().numpy() for p in net.parameters()])
    q.put((rank, len(red.buckets), outs))
    dist.barrier()
    dist.destroy_process_group()


def test_gradient_allreduce_world2_equals_mean_of_rank_gradients():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_grad_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0][1] == 3
    # single-process restatement
    torch.manual_seed(11)
    net = torch.nn.Sequential(torch.nn.Linear(7, 33), torch.nn.Tanh(), torch.nn.Linear(33, 5))
    for step in range(2):
        want = None
        for r in range(world):
            net.zero_grad()
            x = torch.randn(4, 7, generator=torch.Generator().manual_seed(100 * step + r))
            net(x).pow(2).sum().backward()
            g = [p.grad.clone().numpy() / world for p in net.parameters()]
            want = g if want is None else [a + b for a, b in zip(want, g)]
        for r in range(world):
            for got, w in zip(res[r][2][step], want):
                assert np.allclose(got, w, rtol=1e-6, atol=1e-7)


def test_gradient_reducer_single_process_keeps_bucket_views():
    net = torch.nn.Linear(3, 2)
    red = bd.GradAllReducer(net.parameters())
    net(torch.ones(1, 3)).sum().backward()
    red.finish()
    assert net.weight.grad.data_ptr() == red.buckets[0][0].data_ptr() + 4 * 4   # bias (2 -> 4 floats) first
    assert torch.equal(net.weight.grad, torch.ones(2, 3))
