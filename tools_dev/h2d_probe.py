import os, sys, time, threading
import torch
dev = torch.device("cuda", 0)
torch.zeros(1, device=dev)
n = 24 << 20
pin = torch.empty(n, dtype=torch.uint8).pin_memory(); d = torch.empty(n, dtype=torch.uint8, device=dev)
def timed(stream, label, reps=20):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(stream):
        e0.record(stream)
        for _ in range(reps): d.copy_(pin, non_blocking=True)
        e1.record(stream)
    e1.synchronize()
    print(label, "GB/s", n * reps / (e0.elapsed_time(e1) * 1e-3) / 1e9)
timed(torch.cuda.current_stream(), "null stream, main thread")
s = torch.cuda.Stream()
timed(s, "side stream, main thread")
th = threading.Thread(target=lambda: (torch.cuda.set_device(dev), timed(s, "side stream, worker thread"))); th.start(); th.join()
# with a busy compute stream
x = torch.randn(8192, 8192, device=dev)
def busy():
    for _ in range(200): y = x @ x
busy(); torch.cuda.synchronize()
busy(); timed(s, "side stream while the null stream runs GEMMs"); torch.cuda.synchronize()
# event synchronize latency
t0 = time.perf_counter()
for _ in range(20):
    e = torch.cuda.Event(); e.record(torch.cuda.current_stream()); e.synchronize()
print("event record+synchronize on an idle stream: ms", (time.perf_counter() - t0) / 20 * 1e3)
sl = pin[: n // 2]
print("slice is_pinned", sl.is_pinned())
