#!/usr/bin/env python3
"""What the memory system of the GPU box gives plain streams, to read the Regrid kernels' byte rates against: a fill
(write only), a reduction (read only) and a copy (half / half), on buffers far larger than the 256 MiB Infinity Cache.
torch kernels, torch.cuda.Event timing, median of 7.  usage: python tools/stream_ceiling.py [GB]"""
import sys

import torch


def timed(fn, reps=7):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    return sorted(ts)[len(ts) // 2]


def main():
    gb = float(sys.argv[1]) if len(sys.argv) > 1 else 16.0
    n = int(gb * 1e9 / 4)
    x = torch.empty(n, dtype=torch.float32, device="cuda")
    y = torch.empty(n, dtype=torch.float32, device="cuda")
    x.fill_(1.0)
    ms = timed(lambda: y.fill_(2.0))
    print("fill   %6.2f GB written            %7.3f ms  %5.2f TB/s" % (gb, ms, gb / ms))
    ms = timed(lambda: x.sum())
    print("sum    %6.2f GB read               %7.3f ms  %5.2f TB/s" % (gb, ms, gb / ms))
    ms = timed(lambda: y.copy_(x))
    print("copy   %6.2f GB read + %6.2f written %7.3f ms  %5.2f TB/s" % (gb, gb, ms, 2 * gb / ms))
    xd = x.view(torch.float64)
    yd = y.view(torch.float64)
    ms = timed(lambda: torch.add(xd, 1.0, out=yd))
    print("add64  %6.2f GB read + %6.2f written %7.3f ms  %5.2f TB/s" % (gb, gb, ms, 2 * gb / ms))


if __name__ == "__main__":
    main()
