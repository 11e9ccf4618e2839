#!/usr/bin/env python3
"""What the card's HBM delivers for pure writes, pure reads and a copy (torch's own fill / sum / copy kernels on 8 GiB): the ceiling a
write-dominated Regrid (configuration 5: 71 % of its algorithmic bytes are stores) can be held against, beside the 8 TB/s of the data sheet.
usage (GPU box): python tools/hbm_mix_probe.py"""
import json

import torch


def timed(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3


def main():
    out = {}
    for dt, name in ((torch.float32, "f32"), (torch.float64, "f64")):
        n = (8 << 30) // torch.empty(0, dtype=dt).element_size()
        x = torch.empty(n, dtype=dt, device="cuda")
        y = torch.empty(n, dtype=dt, device="cuda")
        gb = x.numel() * x.element_size() / 1e9
        out[name] = {"fill_write_only_GBs": round(gb / timed(lambda: x.fill_(1.5)), 0),
                     "sum_read_only_GBs": round(gb / timed(lambda: x.sum()), 0),
                     "copy_read_plus_write_GBs": round(2 * gb / timed(lambda: y.copy_(x)), 0),
                     "add_2reads_1write_GBs": round(3 * gb / timed(lambda: torch.add(x, y, out=y)), 0)}
        del x, y
    print(json.dumps(out))


if __name__ == "__main__":
    main()
