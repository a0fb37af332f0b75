"""Named, level-gated timing sections (API of the reference's ``timings`` singleton, utils/profiler.py:7-61:
``set_level / reset / add_cnt / start / stop / env(name, level)`` and the ms-per-image report).

MI355X-first difference: a section is bracketed by a pair of HIP events on the current stream instead of two
``torch.cuda.synchronize()`` calls, so enabling the profiler does not serialise the GPU; events are resolved
once, when the report is read.  On a machine without a GPU it falls back to host wall-clock."""
from __future__ import annotations

import time
from collections import defaultdict
from contextlib import contextmanager

import torch


class Timings:
    def __init__(self, level: int = 0):
        self.level = level
        self.average = True
        self.reset()

    def add_cnt(self, cnt: int = 1):
        if self.level >= 0:
            self.cnt += cnt

    def set_level(self, level: int):
        self.level = level

    def reset(self):
        self.records = defaultdict(float)   # name -> seconds (resolved)
        self.counts = defaultdict(int)
        self._open = {}                     # name -> start event or start time
        self._pending = defaultdict(list)   # name -> [(ev_start, ev_stop)]
        self.cnt = 0

    def _gpu(self) -> bool:
        return torch.cuda.is_available()

    def start(self, name: str, level: int = 0):
        if level <= self.level:
            if self._gpu() and torch.cuda.is_current_stream_capturing():
                return   # sections inside a hipGraph capture cannot be timed (they replay without Python)
            if self._gpu():
                ev = torch.cuda.Event(enable_timing=True)
                ev.record()
                self._open[name] = ev
            else:
                self._open[name] = time.perf_counter()
            self.counts[name] += 1

    def stop(self, name: str, level: int = 0):
        start = self._open.pop(name, None)
        if start is None:
            return
        if isinstance(start, float):
            self.records[name] += time.perf_counter() - start
        else:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
            self._pending[name].append((start, ev))

    def _resolve(self):
        if not any(self._pending.values()):
            return
        torch.cuda.synchronize()
        for name, pairs in self._pending.items():
            for a, b in pairs:
                try:
                    self.records[name] += a.elapsed_time(b) * 1e-3
                except RuntimeError:
                    pass   # an event that never executed
            pairs.clear()

    def __repr__(self):
        if self.cnt > 0:
            self._resolve()
            lines = [f"### Profiler (images: {self.cnt})###"]
            for name in sorted(self.records):
                ms = self.records[name] * 1000
                lines.append(f"# {name:20}: {ms / self.cnt:4.3f} ms per image (number of calls: {self.counts[name]}, "
                             f"per call: {ms / max(1, self.counts[name]):4.3f} ms) ")
            return "\n".join(lines) + "\n"
        if self.cnt == 0:
            return "## Profiler: no batches registered"
        return "## Profiler: disabled"

    @contextmanager
    def env(self, name: str, level: int = 0):
        if level > self.level:   # fast path: section disabled
            yield
            return
        self.start(name, level)
        try:
            yield
        finally:
            self.stop(name)


timings = Timings(level=0)
