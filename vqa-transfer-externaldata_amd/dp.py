"""Single-node data parallelism for the fusion-model train step.

The reference has none (one GPU per process, independent seeds: run.py:25-43,
vqa/trainer.py:165-168).  Samples are independent (LayerNorm is per sample), so
the minibatch is sharded by sample, weights and the feature table are replicated
in every GPU's HBM, and the only exchange is ONE sum-all-reduce per step of the
flat gradient buffer (train-var gradients + the un-aggregated embedding-slice
sum of squares in the tail slot) over RCCL/xGMI.  Every rank then applies the
identical clip_by_global_norm + Adam update.

Gradients are produced already scaled by 1/global_batch (vqa_dims_t.inv_global_batch),
so a SUM reduce yields the gradient of the global-batch mean loss, and unequal
shards are weighted by their sample counts automatically.
"""
from __future__ import annotations

import time

import torch
import torch.distributed as dist


def shard_bounds(n_samples, rank, world):
    """Contiguous shard [lo, hi) of `n_samples` for `rank`; the first n % world ranks get one extra."""
    q, r = divmod(n_samples, world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def shard_batch(batch, rank, world):
    """Slices every per-sample entry of the reference batch dict
    (vqa/datasets/input_ops_vqa_tf_record_memft.py:47-71) along axis 0."""
    n = len(batch["image_idx"])
    if n < world:
        raise ValueError("global batch of %d samples is smaller than the world size %d: a rank without samples cannot "
                         "run the step" % (n, world))
    lo, hi = shard_bounds(n, rank, world)
    return {k: v[lo:hi] for k, v in batch.items()}, n


def allreduce_flat_(flat, group=None, bucket_floats=None):
    """In-place SUM all-reduce of a flat fp32 buffer.  xGMI is point-to-point, so a
    ring all-reduce is bound by one link (~153 GB/s): a 48.6 MB buffer costs
    ~0.6 ms -- small against the step, so the default is ONE collective; pass
    bucket_floats to split it (used to start reducing early-finished gradients)."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return flat
    if flat.is_cuda and dist.get_backend(group) == "gloo":
        # rehearsal path (several ranks sharing one GPU, no RCCL): stage through host memory
        host = flat.detach().cpu()
        dist.all_reduce(host, op=dist.ReduceOp.SUM, group=group)
        flat.copy_(host)
        return flat
    if bucket_floats is None or bucket_floats >= flat.numel():
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
        return flat
    works = []
    for lo in range(0, flat.numel(), bucket_floats):
        works.append(dist.all_reduce(flat[lo:lo + bucket_floats], op=dist.ReduceOp.SUM, group=group, async_op=True))
    for w in works:
        w.wait()
    return flat


def _keep_the_collectives_company(group=None, overlapped=True):
    """The weight-stationary GRU back-propagation (csrc/gru_ws.hip) is one launch of 256 workgroups that must ALL be resident
    (one per CU, 160 KB of LDS each) before any of them gets past its first step.  Under RCCL the first gradient bucket's
    all-reduce kernel is running on some CUs when that launch arrives: the recurrence would wait for the collective to end
    instead of running beside it.  So with more than one rank on RCCL the back-propagation stays on the per-step kernels
    (the forward form, which has no collective beside it, stays on); VQA_HOT_GRU_WS_DP=1 keeps both (unmeasured: no
    multi-GPU node was available to this build).  Ranks that SHARE a GPU switch the form off altogether."""
    import os
    if not (dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1):
        return
    from . import _lib
    if torch.cuda.is_available() and int(os.environ.get("LOCAL_WORLD_SIZE", "1")) > torch.cuda.device_count():
        # several ranks on ONE GPU (a rehearsal): two of those launches at once would each hold CUs the other needs and
        # wait for each other until their bounded spins give up -- per-step kernels in both directions
        _lib.load().vqa_gru_ws_set_mode(0)
    elif overlapped and dist.get_backend(group) == "nccl" and os.environ.get("VQA_HOT_GRU_WS_DP") != "1":
        _lib.load().vqa_gru_ws_set_mode(1)


class GradAllReduce:
    """Callable handed to FusionEngine.train_step(allreduce=...)."""

    def __init__(self, engine=None, group=None, bucket_floats=None):
        self.group = group
        self.bucket_floats = bucket_floats
        _keep_the_collectives_company(group, overlapped=False)      # (reduces after backward: only the shared-GPU case matters)

    def __call__(self, grad_flat):
        return allreduce_flat_(grad_flat, self.group, self.bucket_floats)


class BucketedAllReduce:
    """Overlapped gradient reduction: FusionEngine.backward(reducer=...) calls start(bucket) as soon as a
    backward phase has been enqueued and finish() at the end.  On RCCL each bucket is an async all-reduce:
    the collective stream waits (event) for the compute stream at the point of the call, i.e. for the phase
    that produced the bucket, and runs beside the next phase; finish() makes the compute stream wait for
    all of them before the optimizer.  Buckets: everything-but-GRU/embedding (9-60 MB), embedding table
    (19.7 MB at Vq = 16384) + slice sum of squares, GRU gate weights (10.8 MB), GRU candidate weights (5.4 MB, the
    only one with nothing left to hide behind)."""

    def __init__(self, group=None, timing=False):
        self.group = group
        _keep_the_collectives_company(group)
        self._works = []
        self.timing = timing          # bench.py: measure the time the compute stream spends blocked in finish()
        self._events = []             # (before, after) event pairs on the compute stream, one per finish()
        self._host_ms = []            # gloo rehearsal path: the reduction is synchronous, timed on the host
        self._host_acc = 0.0

    def start(self, bucket):
        if not dist.is_initialized() or dist.get_world_size(self.group) == 1:
            return
        if bucket.is_cuda and dist.get_backend(self.group) == "gloo":
            t0 = time.perf_counter()
            allreduce_flat_(bucket, self.group)            # rehearsal path: synchronous host staging
            self._host_acc += (time.perf_counter() - t0) * 1e3
            return
        self._works.append(dist.all_reduce(bucket, op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def finish(self):
        """Makes the compute stream wait for every bucket started since the last finish().  With timing on, two
        events bracket the waits on the compute stream: nothing else is enqueued between them, so their distance
        is exactly the reduction time that backward did NOT hide (the exposed all-reduce time of the step)."""
        ev = None
        if self.timing and self._works and self._works[0] is not None and torch.cuda.is_available():
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record()
        for w in self._works:
            w.wait()
        if ev is not None:
            ev[1].record()
            self._events.append(ev)
        if self.timing and self._host_acc:
            self._host_ms.append(self._host_acc)
        self._host_acc = 0.0
        self._works = []

    def reset_timing(self):
        self._events, self._host_ms, self._host_acc = [], [], 0.0

    def exposed_ms(self):
        """Per-step exposed reduction time in ms (synchronises the recorded events)."""
        if self._events:
            torch.cuda.synchronize()
            return [a.elapsed_time(b) for a, b in self._events]
        return list(self._host_ms)
