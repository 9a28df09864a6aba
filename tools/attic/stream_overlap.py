#!/usr/bin/env python3
"""Do two kernels on two streams overlap on this box?  A long low-occupancy kernel (64 workgroups) on stream A and the same on
stream B: sequential time vs two-stream time, with the legacy default stream and with two non-default streams."""
import time, torch
dev = torch.device("cuda:0")
x = torch.randn(64, 4096, device=dev); y = torch.randn(64, 4096, device=dev)
def work(t):
    for _ in range(200): t = torch.sin(t) * 1.0001   # 200 tiny dependent kernels of 64 x 4096 elements
    return t
def timed(fn):
    torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); return (time.perf_counter() - t0) * 1e3
work(x); work(y)
print("sequential on the default stream: %.2f ms" % timed(lambda: (work(x), work(y))))
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def two():
    with torch.cuda.stream(s1): work(x)
    with torch.cuda.stream(s2): work(y)
print("two non-default streams:          %.2f ms" % timed(two))
def mixed():
    with torch.cuda.stream(s1): work(x)
    work(y)
print("default + one side stream:        %.2f ms" % timed(mixed))
# one big kernel each: matmul chains that do not fill the chip
a = torch.randn(512, 512, device=dev); b = torch.randn(512, 512, device=dev)
def mm(t):
    for _ in range(300): t = (t @ b) * 0.04
    return t
mm(a)
print("matmul chain x2 sequential:       %.2f ms" % timed(lambda: (mm(a), mm(a))))
def mm2():
    with torch.cuda.stream(s1): mm(a)
    with torch.cuda.stream(s2): mm(a)
print("matmul chain x2 on two streams:   %.2f ms" % timed(mm2))
