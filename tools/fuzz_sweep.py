#!/usr/bin/env python
"""Extended fuzz sweep (GPU box): the plan-vs-oracle fuzzers of tests/test_gpu_fuzz.py over seeds the suite does not hold.
   python tools/fuzz_sweep.py [first seed = 1000] [count = 300]     — prints the seeds that disagree (none expected)."""
import os, sys, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pytest
from frostdb_amd import physicalplan as pp
from tests import test_gpu_fuzz as F

first = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
count = int(sys.argv[2]) if len(sys.argv) > 2 else 300
bad = []
for seed in range(first, first + count):
    for fn, needs_mp in ((F.test_fuzz_plan_vs_oracle, True), (F.test_fuzz_plain_strings_and_bools_vs_oracle, False)):
        mp = pytest.MonkeyPatch()
        try:
            fn(pp, seed, mp) if needs_mp else fn(pp, seed)
        except Exception:  # noqa: BLE001
            bad.append((fn.__name__, seed))
            traceback.print_exc(limit=3)
        finally:
            mp.undo()
print(f"seeds {first} … {first + count - 1}: {2 * count} cases, {len(bad)} disagree {bad}")
