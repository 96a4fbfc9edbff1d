#!/usr/bin/env python3
"""Exhaustive check of csrc/rt_fastdiv.hpp on the GPU: the 5-instruction quotient equals IEEE n/d for EVERY
pair of fp32 significands (2^23 divisors x 2^23 numerators = 7.0e13 pairs).  The result depends on the
significands only as long as nothing over/underflows, which the regular class guarantees; a few corner
exponent pairs of that class are swept on a subset as well.  Prints a progress line per chunk.

    python tools/verify_fastdiv.py [--chunk 65536] > profiles/fastdiv_exhaustive_r01.txt
"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as G

ap = argparse.ArgumentParser()
ap.add_argument("--chunk", type=int, default=1 << 16)
ap.add_argument("--limit", type=int, default=1 << 23, help="number of divisor significands to sweep")
ap.add_argument("--four", action="store_true", help="check fast_div_exact4 (4 instructions, two-word reciprocal) instead of fast_div_exact")
args = ap.parse_args()
p = G.load_package()
t0 = time.time()
total_bad = 0
for first in range(0, args.limit, args.chunk):
    n = min(args.chunk, args.limit - first)
    bad, ex = p.api.selftest_fastdiv(first, n, 0, 0, four=args.four)
    total_bad += bad
    done = first + n
    print(f"divisor significands [{first:#08x}, {done:#08x}): mismatches {bad}  (elapsed {time.time()-t0:.1f}s, "
          f"{done * (1 << 23) / 1e12:.2f}e12 pairs)" + (f"  example n={ex[0]:#x} d={ex[1]:#x}" if bad else ""), flush=True)
for ne, de in [(-64, 39), (40, -40), (-64, -40), (40, 39), (-60, 0), (0, -40)]:
    bad, ex = p.api.selftest_fastdiv(0, min(args.chunk, 1 << 14), ne, de, four=args.four)
    total_bad += bad
    print(f"corner exponents num 2^{ne} den 2^{de}: first {min(args.chunk, 1 << 14)} divisor significands x 2^23: mismatches {bad}", flush=True)
print(f"TOTAL mismatches: {total_bad} over {args.limit} x 2^23 significand pairs in {time.time()-t0:.1f}s")
sys.exit(1 if total_bad else 0)
