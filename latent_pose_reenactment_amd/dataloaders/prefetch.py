"""Input side of the training loop on MI355X: pinned, double-buffered host-to-device copies on a side stream, one batch ahead of the
step that is computing (the reference relies on DataLoader(pin_memory=True) + its prefetching subclass and a blocking
``dict_to_device`` at the top of every iteration: dataloaders/dataloader.py:24-50, runners/holycow.py:233-236).

``DevicePrefetcher(loader, device)`` wraps any iterable of ``(data_dict, target_dict)`` host batches (the plugin dataloader contract) and
yields the same dicts with their tensors in DEVICE staging buffers:
  * batch k+1 is staged while step k computes: host tensors -> a pinned slot (plain memcpy) -> the slot's device buffers by an async copy
    on ``copy_stream``; an event orders the consumer's stream behind the copy (no host synchronisation on the hot path);
  * two slots: before slot s is refilled (batch k+2) the copy stream waits for everything the compute stream had enqueued up to that
    moment -- step k, the last reader of slot s -- and the host waits for the slot's previous H2D before overwriting its pinned memory.
A batch is 18.9 MB per step in meta-training (8 x (8 + 1 + 1 + 3/3) frames of 3 x 256 x 256 fp32): ~0.3 ms at PCIe Gen5 rates, hidden
behind a 40+ ms step.  The hipGraph step copies the staging tensors into its static inputs (device-to-device, ~10 us)."""
import torch


class DevicePrefetcher:
    def __init__(self, loader, device):
        self.loader = loader
        self.device = torch.device(device)
        self.copy_stream = torch.cuda.Stream(device=self.device)
        self.slots = [None, None]          # per slot: {'pinned': {...}, 'dev': {...}, 'event': cuda event of the last H2D}

    def __len__(self):
        return len(self.loader)

    def _buffers(self, slot, key, t):
        s = self.slots[slot]
        buf = s['pinned'].get(key)
        if buf is None or buf.shape != t.shape or buf.dtype != t.dtype:
            s['pinned'][key] = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
            s['dev'][key] = torch.empty(t.shape, dtype=t.dtype, device=self.device)
        return s['pinned'][key], s['dev'][key]

    def _stage(self, batch, slot):
        if self.slots[slot] is None:
            self.slots[slot] = {'pinned': {}, 'dev': {}, 'event': None}
        s = self.slots[slot]
        if s['event'] is not None:
            s['event'].synchronize()                                   # the pinned slot's previous H2D is done (it was issued a step ago)
        # the device buffers of this slot were last read by work already enqueued on the consumer's stream: let the copy wait for it
        self.copy_stream.wait_stream(torch.cuda.current_stream(self.device))
        out = []
        with torch.cuda.stream(self.copy_stream):
            for di, d in enumerate(batch):
                o = {}
                for k, v in d.items():
                    if torch.is_tensor(v) and not v.is_cuda:
                        pin, dev = self._buffers(slot, (di, k), v)
                        pin.copy_(v)                                    # host memcpy into pinned memory
                        dev.copy_(pin, non_blocking=True)               # async H2D on the copy stream
                        o[k] = dev
                    else:
                        o[k] = v
                out.append(o)
            ev = torch.cuda.Event()
            ev.record(self.copy_stream)
        s['event'] = ev
        return tuple(out), ev

    def __iter__(self):
        pending, slot = None, 0
        for batch in self.loader:
            staged = self._stage(batch, slot)
            slot ^= 1
            if pending is not None:
                torch.cuda.current_stream(self.device).wait_event(pending[1])
                yield pending[0]
            pending = staged
        if pending is not None:
            torch.cuda.current_stream(self.device).wait_event(pending[1])
            yield pending[0]
