"""Measurement knob of `DeviceResize(stager=)` / `tools/probe_input_pipeline.py e2e` (moved out of the product package in round 6: measured negative)."""
import torch


class PinnedStager:
    """OPTIONAL, off everywhere by default: host-to-device copies of large PAGEABLE batches through a small ring of pinned buffers (a host
    memcpy into pinned memory, GIL released, then an asynchronous DMA; a buffer is reused once the copy out of it has finished). Built to test
    whether the runtime's synchronous pageable copy was what held the file-fed loop back — it was not: 5.6-5.8 k img/s with the ring against
    6.4 k without (profiles/r05_input_pipeline_end_to_end.txt; the bound is the loaders' hand-over of the ragged batches). Kept as the
    measurement knob of ``DeviceResize(stager=)`` / ``tools/probe_input_pipeline.py e2e``. (The loader's own ``pin_memory`` would allocate a new
    pinned block per differently sized ragged batch: tens of seconds for the pool and a fragmented host allocator.)"""

    def __init__(self, nbuf=3):
        self.bufs, self.events, self.i = [None] * nbuf, [None] * nbuf, 0

    def to_device(self, t, device):
        if t.is_cuda or t.is_pinned():
            return t.to(device, non_blocking=True)
        k = self.i
        self.i = (k + 1) % len(self.bufs)
        if self.events[k] is not None:
            self.events[k].synchronize()
        n = t.numel() * t.element_size()
        if self.bufs[k] is None or self.bufs[k].numel() < n:
            self.bufs[k] = torch.empty(int(n * 1.25) + 4096, dtype=torch.uint8, pin_memory=True)
        p = self.bufs[k][:n].view(t.dtype).view(t.shape)
        p.copy_(t)
        d = p.to(device, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(device))
        self.events[k] = ev
        return d
