"""Input side of the training loop on MI355X: pinned, triple-buffered host-to-device copies on a side stream, one batch ahead of the
step that is computing (the reference relies on DataLoader(pin_memory=True) + its prefetching subclass and a blocking
``dict_to_device`` at the top of every iteration: dataloaders/dataloader.py:24-50, runners/holycow.py:233-236).

``DevicePrefetcher(loader, device)`` wraps any iterable of ``(data_dict, target_dict)`` host batches (the plugin dataloader contract) and
yields the same dicts with their tensors in DEVICE staging buffers:
  * batch k+1 is staged while step k computes: host tensors -> a pinned slot (plain memcpy; skipped when the loader already delivers pinned
    tensors) -> the slot's device buffers by an async copy on ``copy_stream``;
  * NO device-side dependency between the copy stream and the compute stream: measured on MI355X (profiles/README.md, r03 input path), an
    H2D copy running beside the captured step costs nothing (+0.07 ms for 25 MB), while every cross-stream event wait placed between
    graph replays costs ~1 ms per step.  Ordering is kept on the HOST instead, with events that are already complete in steady state:
      - a batch is handed out only after its H2D event completed (issued a whole step earlier);
      - three slots: slot s is refilled for batch k+3 only after a marker recorded on the compute stream behind step k -- the last reader
        of slot s -- completed (recorded when batch k+2 was staged), so the host never runs more than two steps ahead of the device.
A meta-training batch is 69 MB per step (8 x (8 + 1 + 1 + 1) frames of 3 x 256 x 256 fp32), a fine-tuning batch 25 MB: 1.6 / 0.55 ms at the
44 GB/s measured, hidden behind a 50 / 22 ms step.  The hipGraph step copies the staging tensors into its static inputs (device-to-device, ~10 us)."""
import ctypes
import os
import time

import torch

SLOTS = 3
_PIN_COPY = os.environ.get('LP_PIN_COPY', 'memmove')        # memmove | torch


def _host_copy(pin, v):
    """pageable host tensor -> pinned slot.  A plain single-threaded memmove (GIL released): ``Tensor.copy_`` between two host tensors runs
    on the intra-op thread pool, and on a host with few free cores its workers -- spinning after the copy -- starve the HIP runtime's
    own threads: measured +14 .. +23 % per captured fine-tuning step for a 25 MB batch, against +0.5 % with the copy below
    (profiles/r03_input_path.txt).  2.5 ms for 25 MB, hidden behind the step like the H2D itself."""
    if _PIN_COPY == 'memmove' and v.is_contiguous() and pin.is_contiguous() and v.dtype == pin.dtype:
        ctypes.memmove(pin.data_ptr(), v.data_ptr(), v.numel() * v.element_size())
    else:
        pin.copy_(v)


class DevicePrefetcher:
    def __init__(self, loader, device):
        self.loader = loader
        self.device = torch.device(device)
        self.copy_stream = torch.cuda.Stream(device=self.device, priority=int(os.environ.get('LP_COPY_PRIORITY', '0')))
        self.slots = [None] * SLOTS        # per slot: {'pinned': {...}, 'dev': {...}, 'reader_done': event | None}
        self.markers = []                  # (batch index the marker covers, event): everything enqueued on the compute stream so far
        self.waits = {'slot_free': 0.0, 'h2d_done': 0.0, 'pin_memcpy': 0.0}      # host seconds spent waiting / copying (diagnostics)

    def __len__(self):
        return len(self.loader)

    def _buffers(self, slot, key, t, need_pin):
        s = self.slots[slot]
        dev = s['dev'].get(key)
        if dev is None or dev.shape != t.shape or dev.dtype != t.dtype:
            s['dev'][key] = torch.empty(t.shape, dtype=t.dtype, device=self.device)
            s['pinned'].pop(key, None)
        pin = None
        if need_pin:
            pin = s['pinned'].get(key)
            if pin is None or pin.shape != t.shape or pin.dtype != t.dtype:
                pin = s['pinned'][key] = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
        return pin, s['dev'][key]

    def _stage(self, batch, index):
        slot = index % SLOTS
        if self.slots[slot] is None:
            self.slots[slot] = {'pinned': {}, 'dev': {}, 'reader_done': None}
        s = self.slots[slot]
        # marker: all compute-stream work enqueued up to now, i.e. every step up to batch index - 2 (batch index - 1 has not been handed out yet)
        mk = torch.cuda.Event()
        mk.record(torch.cuda.current_stream(self.device))
        self.markers.append((index - 2, mk))
        # the last reader of this slot was the step of batch index - SLOTS: wait (on the host) for a marker that covers it
        while self.markers and self.markers[0][0] < index - SLOTS:
            self.markers.pop(0)
        if self.markers and self.markers[0][0] >= index - SLOTS and index >= SLOTS:
            t0 = time.perf_counter(); self.markers[0][1].synchronize(); self.waits['slot_free'] += time.perf_counter() - t0
        out = []
        with torch.cuda.stream(self.copy_stream):
            for di, d in enumerate(batch):
                o = {}
                for k, v in d.items():
                    if torch.is_tensor(v) and not v.is_cuda:
                        pin, dev = self._buffers(slot, (di, k), v, need_pin=not v.is_pinned())
                        if pin is not None:
                            t0 = time.perf_counter()
                            _host_copy(pin, v)                          # host memcpy into pinned memory
                            self.waits['pin_memcpy'] += time.perf_counter() - t0
                            v = pin
                        dev.copy_(v, non_blocking=True)                 # async H2D on the copy stream (pinned source)
                        o[k] = dev
                    else:
                        o[k] = v
                out.append(o)
            ev = torch.cuda.Event()
            ev.record(self.copy_stream)
        return tuple(out), ev

    def __iter__(self):
        pending = None
        for index, batch in enumerate(self.loader):
            staged = self._stage(batch, index)
            if pending is not None:
                t0 = time.perf_counter(); pending[1].synchronize(); self.waits['h2d_done'] += time.perf_counter() - t0      # host-side: this H2D was issued a whole step ago
                yield pending[0]
            pending = staged
        if pending is not None:
            pending[1].synchronize()
            yield pending[0]
