#!/usr/bin/env python3
"""Calibration: what does a PURE WRITE stream of the render's size reach on this MI355X?

Times (hipEvents, 20 reps) filling a 693.6 MB buffer (= C4's observation batch) with
  * hipMemsetAsync (torch.Tensor.zero_ -> __amd_rocclr_fillBufferAligned)
  * torch.Tensor.fill_ (vectorised elementwise kernel)
  * a device-to-device copy of the same size (read + write) for reference
The render kernel's roofline fraction is quoted against the 8 TB/s spec; this tool shows where the
practical write-only ceiling sits so that fraction can be read in context (profiles/r1/hbm_write_ceiling.txt).
"""
import torch

n = 32768 * 21168
buf = torch.empty(n, dtype=torch.uint8, device="cuda")
src = torch.empty(n, dtype=torch.uint8, device="cuda")


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


print("bytes", n)
for name, fn, mult in (("memset (zero_)", lambda: buf.zero_(), 1), ("fill_(7)", lambda: buf.fill_(7), 1),
                       ("int32 view fill_", lambda: buf.view(torch.int32).fill_(7), 1),
                       ("copy_ (read+write)", lambda: buf.copy_(src), 2)):
    us = timeit(fn)
    print("%-22s %8.1f us  %7.1f GB/s (%s)" % (name, us, mult * n / us / 1e3, "read+write bytes" if mult == 2 else "written bytes"))
