"""Host side of the eps stream (RNG contract in include/bnn_hip.h).

The reference draws eps from torch's global generator (pytorch_bayesian/nn/core.py:45).
Here every draw is addressed by (seed, tensor stream, MC sample, epoch): counter-based, so
the fused GEMM, the standalone sampler, the backward pass and every GPU of a sharded MC
run regenerate identical eps without ever storing it.
"""
import itertools
import threading

import torch

_stream_ids = itertools.count(1)
_lock = threading.Lock()


def new_stream_id():
    """A tensor-stream id (1 .. 65535: 16 bits of the Philox counter), handed out in construction order so
    that every rank of a multi-GPU run that builds the same model gets the same ids.  Ids are never reused:
    a wrapped id would make two tensors draw identical eps, so the 65 536th request raises."""
    with _lock:
        i = next(_stream_ids)
    if i > 0xFFFF:
        raise RuntimeError("eps stream ids exhausted: more than 65535 posterior tensors were created in this process "
                           "(the stream field of the Philox counter is 16 bits wide)")
    return i


class EpsGenerator:
    """seed + draw counter.  `epoch_host` advances once per draw call; `epoch_dev` is a
    device word a captured graph bumps itself (bnn_rng_advance) so replays differ."""

    def __init__(self):
        self._seed = None
        self._torch_seed = None
        self.epoch_host = 0
        self._epoch_dev = {}

    def manual_seed(self, seed):
        self._seed = int(seed) & (2 ** 64 - 1)
        self._torch_seed = torch.initial_seed()
        self.epoch_host = 0
        for t in self._epoch_dev.values():
            t.zero_()
        return self

    @property
    def seed(self):
        # Follow torch.manual_seed(): a new torch seed re-keys and restarts the stream.
        ts = torch.initial_seed()
        if self._seed is None or ts != self._torch_seed:
            self._seed = ts & (2 ** 64 - 1)
            self._torch_seed = ts
            self.epoch_host = 0
        return self._seed

    def next_epoch(self):
        _ = self.seed
        e = self.epoch_host
        self.epoch_host = (self.epoch_host + 1) & 0xFFFFFFFF
        return e

    def epoch_dev(self, device):
        key = (device.type, device.index)
        t = self._epoch_dev.get(key)
        if t is None:
            t = torch.zeros(4, dtype=torch.int32, device=device)
            self._epoch_dev[key] = t
        return t

    def private_epoch_cell(self, device):
        """Context manager: launches recorded inside read (and bump) a device epoch word of their own instead of the
        device-wide one -- for captured graphs that are replayed CONCURRENTLY on different streams (two steps in flight
        must not read one epoch word that either of them bumps).  Yields the cell (4 x int32, word 0 is the epoch)."""
        gen = self

        class _Cell:
            def __enter__(self_):
                self_.key = (device.type, device.index)
                self_.prev = gen._epoch_dev.get(self_.key)
                self_.cell = torch.zeros(4, dtype=torch.int32, device=device)
                gen._epoch_dev[self_.key] = self_.cell
                return self_.cell

            def __exit__(self_, *exc):
                if self_.prev is None:
                    gen._epoch_dev.pop(self_.key, None)
                else:
                    gen._epoch_dev[self_.key] = self_.prev
                return False

        return _Cell()


default_generator = EpsGenerator()


def manual_seed(seed):
    """Seed the eps stream (and restart its draw counter)."""
    default_generator.manual_seed(seed)


GEN_PHILOX10_U24 = 0        # include/bnn_hip.h BNN_GEN_*: the default stream (24-bit uniforms, 4 eps per Philox4x32-10 block)
GEN_PHILOX7_U16 = 1         # 8 eps per Philox4x32-7 block from 16-bit uniforms: for weights that are rounded to bf16 anyway

# which stream a layer's sample() keys its draw with in the bf16 compute mode (BNN_BF16_GENERATOR=philox10 keeps the default
# stream everywhere, for A/B runs); the fp32 parity mode always draws from the default stream
_bf16_gen = GEN_PHILOX10_U24 if __import__("os").environ.get("BNN_BF16_GENERATOR", "philox7") == "philox10" else GEN_PHILOX7_U16


def generator_for(compute_mode):
    return _bf16_gen if compute_mode == "bf16" else GEN_PHILOX10_U24


def set_bf16_generator(name):
    """'philox7' (default: BNN_GEN_PHILOX7_U16) or 'philox10' (the default stream) for draws keyed in the bf16 compute mode."""
    global _bf16_gen
    if name not in ("philox7", "philox10"):
        raise ValueError("generator must be 'philox7' or 'philox10'")
    _bf16_gen = GEN_PHILOX7_U16 if name == "philox7" else GEN_PHILOX10_U24


class DrawKey:
    """Everything needed to re-create one draw of one tensor (gen: which eps stream, part of the key)."""
    __slots__ = ("seed", "stream", "sample0", "nsamples", "epoch_host", "epoch_dev_delta", "gen")

    def __init__(self, seed, stream, sample0, nsamples, epoch_host, epoch_dev_delta=0, gen=GEN_PHILOX10_U24):
        self.seed = seed
        self.stream = stream
        self.sample0 = sample0
        self.nsamples = nsamples
        self.epoch_host = epoch_host
        self.epoch_dev_delta = epoch_dev_delta
        self.gen = gen

    def last_sample(self):
        return DrawKey(self.seed, self.stream, self.sample0 + self.nsamples - 1, 1,
                       self.epoch_host, self.epoch_dev_delta, self.gen)
