#!/usr/bin/env python3
"""Developer tool (GPU box): what the PCIe link gives for device -> pinned host copies (one 512 MiB copy; two halves on two streams)."""
import torch, time
x = torch.empty(512*1024*1024, dtype=torch.uint8, device="cuda")
h = torch.empty(512*1024*1024, dtype=torch.uint8, pin_memory=True)
for _ in range(2): h.copy_(x, non_blocking=True); torch.cuda.synchronize()
t0=time.perf_counter()
for _ in range(5): h.copy_(x, non_blocking=True)
torch.cuda.synchronize(); dt=(time.perf_counter()-t0)/5
print("D2H pinned 512 MiB: %.2f ms = %.1f GB/s" % (dt*1e3, 0.5368709/dt))
# two halves on two streams
s1,s2=torch.cuda.Stream(),torch.cuda.Stream()
torch.cuda.synchronize(); t0=time.perf_counter()
for _ in range(5):
    with torch.cuda.stream(s1): h[:256*1024*1024].copy_(x[:256*1024*1024], non_blocking=True)
    with torch.cuda.stream(s2): h[256*1024*1024:].copy_(x[256*1024*1024:], non_blocking=True)
torch.cuda.synchronize(); dt=(time.perf_counter()-t0)/5
print("D2H two streams: %.2f ms = %.1f GB/s" % (dt*1e3, 0.5368709/dt))
