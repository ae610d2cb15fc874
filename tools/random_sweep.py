#!/usr/bin/env python
"""A longer run of tests/test_gpu_random.py::test_random_problem than the 40 seeds of the suite:
python tools/random_sweep.py FIRST COUNT   (prints the failures and the skip count; development)"""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_random as T  # noqa: E402
from gadfly_amd import _lib as hip  # noqa: E402

first, count = int(sys.argv[1]), int(sys.argv[2])
hip.require_device()
bad, skipped = [], 0
for seed in range(first, first + count):
    try:
        T.test_random_problem(hip, seed)
    except pytest.skip.Exception:
        skipped += 1
    except Exception as e:      # noqa: BLE001
        bad.append((seed, repr(e)[:300]))
        print("FAIL", seed, repr(e)[:300], flush=True)
    if (seed - first) % 50 == 49:
        print(f"... {seed - first + 1} seeds, {len(bad)} failures, {skipped} skipped", flush=True)
print(f"{count} seeds from {first}: {len(bad)} failures, {skipped} skipped")
sys.exit(1 if bad else 0)
