"""Wall-clock stopwatch behind the reference's names (lib/utils/timer.py): callers read `total_time`, `calls`, `diff`, `average_time`
and call tic() / toc(average=True) (ctpn/demo.py:57-67). Measured on the monotonic clock (time.perf_counter: an NTP step in the middle of
a detection cannot produce a negative or inflated interval, which time.time() can); also usable as a context manager."""
import time


class Timer(object):
    __slots__ = ("total_time", "calls", "start_time", "diff")

    def __init__(self):
        self.total_time, self.calls, self.start_time, self.diff = 0.0, 0, 0.0, 0.0

    @property
    def average_time(self):
        return self.total_time / self.calls if self.calls else 0.0

    def tic(self):
        self.start_time = time.perf_counter()

    def toc(self, average=True):
        now = time.perf_counter()
        self.diff = now - self.start_time
        self.total_time += self.diff
        self.calls += 1
        return self.average_time if average else self.diff

    def __enter__(self):
        self.tic()
        return self

    def __exit__(self, *exc):
        self.toc()
        return False
