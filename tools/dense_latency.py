#!/usr/bin/env python
"""Latency of the dense building blocks of the wide combine (gadfly_dense.hip) on the GPU:
gf_bgemm at the chunk-map sizes, gf_dense_solve, gf_lft_tree_scan.  python tools/dense_latency.py [W]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from gadfly_amd import _lib  # noqa: E402


def clock(fn, reps=50):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps        # us


def main():
    lib, p = _lib.load(), _lib.ptr
    W = int(sys.argv[1]) if len(sys.argv) > 1 else 172
    WP = lib.gf_dense_width(W)
    for batch in (1, 2, 8, 64):
        kw = dict(dtype=torch.float64, device="cuda")
        A = torch.randn((batch, WP, WP), **kw)
        B = torch.randn((batch, WP, WP), **kw)
        C = torch.empty((batch, WP, WP), **kw)
        v = torch.randn((batch, WP, 1), **kw)
        y = torch.empty((batch, WP, 1), **kw)
        for ta, tb in ((0, 0), (0, 1), (1, 0)):
            us = clock(lambda: lib.gf_bgemm(batch, ta, tb, WP, WP, WP, p(A), WP, WP * WP, p(B), WP, WP * WP,
                                            None, 0, 0, p(C), WP, WP * WP, None))
            fl = 2.0 * batch * WP ** 3 / us / 1e6
            print(f"bgemm W'={WP} batch={batch:3d} ta={ta} tb={tb}: {us:8.1f} us  {fl:7.2f} TFLOP/s")
        us = clock(lambda: lib.gf_bgemm(batch, 0, 0, WP, 1, WP, p(A), WP, WP * WP, p(v), 1, WP,
                                        None, 0, 0, p(y), 1, WP, None))
        print(f"bgemm mat-vec W'={WP} batch={batch:3d}: {us:8.1f} us")
        for nrhs in (WP + 16, 2 * WP + 16):
            Aa = torch.eye(WP, **kw).expand(batch, WP, WP).contiguous() + 0.1 * torch.randn((batch, WP, WP), **kw) / WP ** 0.5
            R = torch.randn((batch, WP, nrhs), **kw)
            us = clock(lambda: lib.gf_dense_solve(batch, WP, nrhs, p(Aa), p(R), None), reps=20)
            print(f"dense_solve n={WP} nrhs={nrhs} batch={batch:3d}: {us:8.1f} us")
    for B_, P in ((1, 128), (1, 256), (64, 16)):
        if B_ > 1 and WP > 96:
            continue
        kw = dict(dtype=torch.float64, device="cuda")
        n = B_ * P
        Ph = 0.5 * torch.randn((n, WP, WP), **kw) / WP ** 0.5
        L = torch.randn((n, WP, 6), **kw)
        Xb = L @ L.transpose(1, 2) / 6
        M = torch.randn((n, WP, 6), **kw)
        G = -(M @ M.transpose(1, 2)) / 6
        Yb, m = torch.randn((n, WP), **kw), torch.randn((n, WP), **kw)
        Xs, Ys = torch.empty((n, WP, WP), **kw), torch.empty((n, WP), **kw)
        work = torch.empty((int(lib.gf_lft_tree_work(B_, P, WP)),), **kw)
        us = clock(lambda: lib.gf_lft_tree_scan(B_, P, WP, p(Ph), p(G), p(Xb), p(Yb), p(m), p(Xs), p(Ys),
                                                p(work), None), reps=5)
        print(f"lft_tree_scan W'={WP} B={B_} P={P}: {us / 1e3:8.3f} ms")


if __name__ == "__main__":
    main()
