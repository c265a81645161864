"""Host -> device input feeder for the fixed-field models: a background thread stages numpy batches through pinned buffers into
the engine's own input slots (dctr_input_slot) on a copy stream, up to DCTR_INPUT_SLOTS - 1 batches ahead of the step that
consumes them; the training loop then passes the slot tensors to train_step, which reads them in place (no staging copy, no
allocation per step).  This is the prefetch(500000) of the reference pipeline (DeepFM.py:84) at device granularity: at
~15 M examples/s a step lasts 0.26 ms, the feeder needs ~0.05 ms of one host thread per batch (one memcpy into pinned memory +
one library call that enqueues the three async H2D copies, 1.3 MB at B=4096) and ~5 GB/s of the link."""
from __future__ import annotations

import queue
import threading
from typing import Iterable, Iterator, Tuple

import numpy as np

from . import capi


class DeviceFeeder:
    def __init__(self, engine, batches: Iterable[Tuple[np.ndarray, np.ndarray, np.ndarray]], slots: int = capi.INPUT_SLOTS):
        import torch
        self._torch = torch
        self.eng = engine
        self.n_slots = int(slots)
        MB, F = engine.cfg.max_batch, engine.cfg.field_size
        self.dev = torch.device("cuda", torch.cuda.current_device())
        self.slot = [engine.input_slot(k) for k in range(self.n_slots)]                 # (ids, vals, labels) device views
        # pinned staging, one block per slot laid out like the device slot ([ids | vals | labels]): a whole batch is one H2D copy
        # (ids and vals padded to a multiple of 64 elements, as dctr_create lays the slot out)
        R = -(-MB * F // 64) * 64
        self._pin_block = [torch.zeros((2 * R + MB,), dtype=torch.float32).pin_memory() for _ in range(self.n_slots)]
        self.pin = [(blk[:MB * F].view(torch.int32).view(MB, F), blk[R:R + MB * F].view(MB, F), blk[2 * R:])
                    for blk in self._pin_block]
        self.pin_ptr = [(p[0].data_ptr(), p[1].data_ptr(), p[2].data_ptr()) for p in self.pin]
        self.pin_np = [(p[0].numpy(), p[1].numpy(), p[2].numpy()) for p in self.pin]
        self.full = [tuple(t[:MB] for t in s) for s in self.slot]                       # views of a whole-batch slot, made once
        self.MB = MB
        # The copies, their "filled" event, the consumer's device-side wait and its "consumed" record live in the library
        # (dctr_input_slot_fill / acquire / release / wait_released): per batch the feeder thread pays one C call instead of three
        # torch copies under a stream context, the training thread two ~3 us calls instead of a torch Event made per step (40 us)
        # and a wait_event -- at c2 the training thread was the bottleneck: 290 us of host time per 260 us step.
        # host-side handshake per slot: the feeder may refill slot k only after the consumer has ENQUEUED the step that reads
        # it (release) -- taking the item off the queue is not enough -- and then waits for that step's record on the device
        self.free = [threading.Semaphore(1) for _ in range(self.n_slots)]
        self._q: "queue.Queue" = queue.Queue(maxsize=self.n_slots - 1)
        self._err = None
        self._stop = False
        self._thread = threading.Thread(target=self._run, args=(iter(batches),), daemon=True)
        self._thread.start()

    def _run(self, it):
        try:
            k = 0
            for ids, vals, labels in it:
                if self._stop:
                    break
                B = int(labels.shape[0])
                while not self.free[k].acquire(timeout=0.2):
                    if self._stop:
                        return
                self.eng.input_slot_wait_released(k)      # the slot (and its pinned buffer) is free again
                p_i, p_v, p_l = self.pin_np[k]
                np.copyto(p_i[:B], ids, casting="same_kind")
                np.copyto(p_v[:B], vals, casting="same_kind")
                np.copyto(p_l[:B], labels, casting="same_kind")
                # (advances the slot's generation: a grouping prefetched from its old contents is stale from here on)
                self.eng.input_slot_fill(k, self.pin_ptr[k][0], self.pin_ptr[k][1], self.pin_ptr[k][2], B)
                self._q.put((k, B))
                k = (k + 1) % self.n_slots
        except BaseException as e:                        # noqa: BLE001  (surfaced in the consumer)
            self._err = e
        finally:
            self._q.put(None)

    def __iter__(self) -> Iterator[Tuple["object", "object", "object", int]]:
        while True:
            item = self._q.get()
            if item is None:
                if self._err is not None:
                    raise self._err
                return
            k, B = item
            self.eng.input_slot_acquire(k)                # the current stream waits for the slot's copies
            if B == self.MB:
                s_i, s_v, s_l = self.full[k]
                yield s_i, s_v, s_l, k
            else:
                s_i, s_v, s_l = self.slot[k]
                yield s_i[:B], s_v[:B], s_l[:B], k

    def peek_next_ids(self, wait: float = 0.0):
        """ids view of the NEXT staged batch, or None at the end of the stream -- what Engine.prefetch_ids wants right after the step
        that consumes the current batch has been enqueued.  With the consumer enqueueing faster than the device runs (c2: 0.15 ms
        of host time per 0.26 ms step) the queue is EMPTY at that moment -- every staged batch has been taken -- so the hint would
        never be given: `wait` > 0 blocks up to that many seconds for the feeder to stage one (the consumer could not go on without
        it anyway).  Its H2D copy may still be in flight: dctr_prefetch_ids makes the grouping stream wait for it."""
        with self._q.not_empty:
            if not self._q.queue and wait > 0.0:
                self._q.not_empty.wait(timeout=wait)
            item = self._q.queue[0] if self._q.queue else None
        if item is None:
            return None
        k, B = item
        return self.full[k][0] if B == self.MB else self.slot[k][0][:B]

    def release(self, k: int) -> None:
        """call after enqueueing the step that reads slot k (on the current stream)"""
        self.eng.input_slot_release(k)
        self.free[k].release()

    def close(self) -> None:
        self._stop = True
        self.eng.prefetch_cancel()                        # the batch announced last (peek_next_ids) will not be trained
        try:
            while self._q.get_nowait() is not None:
                pass
        except queue.Empty:
            pass
        self._thread.join(timeout=5)
