"""Batch provider of the region-feature extractor with the contract of vqa/datasets/input_ops_vfeat.py:15-81:
ids (optionally shuffled) -> dataset.get_data on 8 parallel workers -> padded batches -> a prefetch queue.

A batch is a dict of NumPy arrays: id i32[B], image f32[B,540,540,3], box / normal_box f32[B,maxn,4] zero padded to
the batch's longest box list (tf.data padded_batch), num_box i32[B], image_id list[str], image_id_len i32[B].
Decoding + resizing runs in a thread pool (PIL releases the GIL inside its C loops) `prefetch` batches ahead of the
consumer, so the GPU's conv stack does not wait for JPEG decoding; the last batch may be short.

`processes=N` (not in the reference): the same map on N forked worker PROCESSES that write their pixels into a ring of batch
blocks in shared memory -- no GIL between decoders.  The workers are forked inside create(), so call it BEFORE the process
touches the GPU (vfeat_extractor.run does when --loader_processes is given); the ring is page-locked afterwards
(cudaHostRegister) when `pinned` is set."""
from __future__ import annotations

import collections
import mmap
import multiprocessing
import queue as _queue
from concurrent.futures import ThreadPoolExecutor

import numpy as np


def _decode_worker(dataset, ring, tasks, results):
    """worker process: (seq, slot, j, id) -> pixels into ring[slot][j], the rest of the item back through `results`"""
    while True:
        t = tasks.get()
        if t is None:
            return
        seq, slot, j, idx = t
        try:
            item = dataset.get_data(idx, ring[slot][j])
            results.put((seq, j, {k: v for k, v in item.items() if k != "image"}, None))
        except BaseException as e:      # the parent re-raises
            results.put((seq, j, None, "%s: %s" % (type(e).__name__, e)))


def _create_mp(dataset, chunks, batch_size, shape, image_dtype, prefetch, processes, pinned):
    import torch
    if torch.cuda.is_initialized():
        raise RuntimeError("decoding processes must be forked before this process initialises the GPU runtime "
                           "(create the input pipeline before the first torch.cuda call)")
    n_ring = max(1, prefetch) + 2
    block = int(np.prod((batch_size,) + shape)) * np.dtype(image_dtype).itemsize
    shm = mmap.mmap(-1, n_ring * block)                       # anonymous shared mapping: inherited by the forked workers
    ring = [np.frombuffer(shm, dtype=image_dtype, count=block // np.dtype(image_dtype).itemsize, offset=i * block)
            .reshape((batch_size,) + shape) for i in range(n_ring)]
    ctx = multiprocessing.get_context("fork")
    tasks, results = ctx.Queue(), ctx.Queue()
    procs = [ctx.Process(target=_decode_worker, args=(dataset, ring, tasks, results), daemon=True) for _ in range(processes)]
    for p in procs:
        p.start()

    def gen():
        registered = False
        try:
            if pinned:      # page-lock the shared ring now that the consumer runs (the GPU runtime is up by then)
                import ctypes
                import torch
                if torch.cuda.is_available():
                    addr = ctypes.addressof(ctypes.c_char.from_buffer(shm))
                    registered = int(torch.cuda.cudart().cudaHostRegister(addr, n_ring * block, 0)) == 0
            pending, done = collections.deque(), {}
            it, seq, slot = iter(chunks), [0], [0]

            def submit():
                c = next(it, None)
                if c is None:
                    return
                s_, k = seq[0], slot[0] % n_ring
                seq[0] += 1
                slot[0] += 1
                for j, idx in enumerate(c):
                    tasks.put((s_, k, j, idx))
                pending.append((s_, c, k))
                done[s_] = {}
            for _ in range(max(1, prefetch)):
                submit()
            while pending:
                s_, c, k = pending[0]
                while len(done[s_]) < len(c):
                    try:
                        rs, j, item, err = results.get(timeout=5.0)
                    except _queue.Empty:
                        if not all(p.is_alive() for p in procs):
                            raise RuntimeError("an image decoding worker process died")
                        continue
                    if err is not None:
                        raise RuntimeError("image decoding failed in a worker: " + err)
                    done[rs][j] = item
                pending.popleft()
                items = [done[s_][j] for j in range(len(c))]
                del done[s_]
                submit()
                yield _collate(c, items, ring[k][:len(c)])
        finally:
            for _ in procs:
                tasks.put(None)
            for p in procs:
                p.join(timeout=5.0)
                if p.is_alive():
                    p.terminate()
            if registered:
                import ctypes
                import torch
                torch.cuda.cudart().cudaHostUnregister(ctypes.addressof(ctypes.c_char.from_buffer(shm)))
    return gen()


def _collate(ids, items, images=None):
    B = len(items)
    maxn = max([int(x["num_box"]) for x in items] + [1])
    box = np.zeros((B, maxn, 4), np.float32)
    nbox = np.zeros((B, maxn, 4), np.float32)
    for i, x in enumerate(items):
        n = int(x["num_box"])
        box[i, :n], nbox[i, :n] = x["box"], x["normal_box"]
    if images is None:
        images = np.stack([x["image"] for x in items], 0)
    return {"id": np.asarray(ids, np.int32), "image": images, "box": box,
            "normal_box": nbox, "num_box": np.array([int(x["num_box"]) for x in items], np.int32),
            "image_id": [x["image_id"] for x in items],
            "image_id_len": np.array([int(x["image_id_len"]) for x in items], np.int32)}


def create(dataset, batch_size, is_train=False, scope="vfeat_input", shuffle=True, seed=123, num_parallel_calls=8,
           prefetch=10, repeat=1000, reuse_buffers=False, pinned=False, image_dtype=np.float32, processes=0):
    """reuse_buffers (not in the reference): the image blocks of the batches come from a ring of `prefetch + 2`
    preallocated [B,H,W,3] buffers, so a batch's `image` is only valid until the second next batch is requested (its
    slot is refilled then) -- for consumers that upload each batch before asking for the next (the extractor).
    Fresh memory of that size (336 MB per batch of 96) is page-faulted in at well under 1 GB/s in these VMs, which
    otherwise dominates the loader: 54 -> 496 images/s on 8 cores together with the direct write of the pixels into
    the batch block (tools/vfeat_input_bench.py).  pinned: the ring is page-locked memory (needs the GPU runtime).
    image_dtype=np.uint8 (with a dataset that writes into the block): the pixels stay bytes on the host -- a quarter of
    the upload, and no uint8 -> float32 conversion under the GIL in the loader threads; the consumer converts on the
    device (vfeat_extractor.device_batches does).  The reference's batches are float32 (input_ops_vfeat.py:40-51)."""
    ids = list(dataset.ids)
    if is_train and shuffle:
        np.random.RandomState(seed).shuffle(ids)
    chunks = [ids[i:i + batch_size] for i in range(0, len(ids), batch_size)]

    ring_keepalive = []
    direct = bool(getattr(dataset, "supports_image_out", False))
    shape = (int(getattr(dataset, "height", 0)), int(getattr(dataset, "width", 0)), 3)
    if processes and processes > 0:
        if not (direct and reuse_buffers) or is_train:
            raise ValueError("processes > 0 needs a dataset that writes into a caller-owned image slot, reuse_buffers=True "
                             "and one pass over the ids (is_train=False)")
        return _create_mp(dataset, chunks, batch_size, shape, image_dtype, prefetch, int(processes), pinned)
    ring = []
    if direct and reuse_buffers:
        n_ring = max(1, prefetch) + 2
        if pinned:      # page-locked blocks: the consumer's .to(device, non_blocking=True) is an asynchronous DMA
            import torch
            tdt = torch.uint8 if np.dtype(image_dtype) == np.uint8 else torch.float32
            keep = [torch.empty((batch_size,) + shape, dtype=tdt, pin_memory=True) for _ in range(n_ring)]
            ring = [t.numpy() for t in keep]
            ring_keepalive.extend(keep)
        else:
            ring = [np.empty((batch_size,) + shape, image_dtype) for _ in range(n_ring)]
    slot = [0]

    def gen():
        _ = ring_keepalive          # the pinned tensors live as long as the generator
        with ThreadPoolExecutor(max_workers=max(1, num_parallel_calls)) as pool:
            for _ in range(repeat if is_train else 1):
                pending = collections.deque()
                it = iter(chunks)

                def submit():
                    c = next(it, None)
                    if c is None:
                        return
                    if direct:      # workers write their pixels straight into the batch block
                        if reuse_buffers:
                            images = ring[slot[0] % len(ring)][:len(c)]
                            slot[0] += 1
                        else:
                            images = np.empty((len(c),) + shape, image_dtype)
                        pending.append((c, [pool.submit(dataset.get_data, i, images[j]) for j, i in enumerate(c)], images))
                    else:
                        pending.append((c, [pool.submit(dataset.get_data, i) for i in c], None))
                for _ in range(max(1, prefetch)):
                    submit()
                while pending:
                    c, futs, images = pending.popleft()
                    items = [f.result() for f in futs]
                    submit()
                    yield _collate(c, items, images)
    return gen()
