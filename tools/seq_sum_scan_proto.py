"""Prototype (CPU, numpy): can the reference's SEQUENTIAL float32 sums be evaluated exactly, but in parallel?  

While the running sum s stays inside one binade, ulp = 2^(e-23) is constant and s = S * ulp with an integer S in +-[2^23, 2^24):
fl(s + t) = (S + rne(t / ulp)) * ulp, where the rounding of t / ulp depends on S only through its PARITY (ties go to the even result).  A run of terms
is therefore a function {even, odd} -> (increment, smallest / largest prefix), such functions compose associatively, and a block of 64 x 16 terms is
one lane-local pass + a 6-step wave scan.  A prefix that leaves the binade (checked on the smallest / largest prefix) ends the block early: the lane
where it happens is added sequentially, the scan restarts with the new ulp.

This script checks the construction bit for bit against the plain loop on adversarial inputs (ties, cancellation, huge / tiny / non-finite terms)
and prices it with a cost model (a pass 2200 shader cycles, a sequential lane 150, the sequential kernel 10.1 cycles per term).  Result: 2.3 cycles
per term for sums that drift (J J^T diagonals, correlated products), but a sum of zero-mean products crosses a binade boundary every ~30 terms
(8.5 k crossings in 250 k terms) and costs 8-28 cycles per term -- and J^T r hovers around zero at convergence.  The slowest accumulator decides, so the
kernel was not built.  usage: python tools/seq_sum_scan_proto.py
"""
import bisect
import struct
import warnings

import numpy as np

warnings.filterwarnings("ignore")
f32 = np.float32


def seq_sum(s, terms):
    for t in terms:
        s = f32(s + t)
    return s


def expo(s):
    b = struct.unpack("<I", struct.pack("<f", float(s)))[0]
    return ((b >> 23) & 0xFF) - 127


def lane_fn(q_terms):
    """A run of scaled terms as a function of the start parity: (K0, K1, lo0, hi0, lo1, hi1, bad)."""
    K, lo, hi, bad, first = [0, 0], [0, 0], [0, 0], False, True
    for q in q_terms:
        if not (abs(q) < 2.0 ** 25):
            bad = True
            continue
        f = np.floor(q)
        r = f32(q - f)
        fi = int(f)
        for p in (0, 1):
            c = p ^ (K[p] & 1)                       # parity of the running integer
            inc = fi + ((c ^ (fi & 1)) & 1) if r == f32(0.5) else (fi + 1 if r > f32(0.5) else fi)
            K[p] += inc
            lo[p], hi[p] = (K[p], K[p]) if first else (min(lo[p], K[p]), max(hi[p], K[p]))
        first = False
    return (K[0], K[1], lo[0], hi[0], lo[1], hi[1], bad)


def compose(A, B):
    out = [0] * 7
    for p in (0, 1):
        q = p ^ (A[p] & 1)
        out[p] = A[p] + B[q]
        out[2 + 2 * p] = min(A[2 + 2 * p], A[p] + B[2 + 2 * q])
        out[3 + 2 * p] = max(A[3 + 2 * p], A[p] + B[3 + 2 * q])
    out[6] = A[6] or B[6]
    return tuple(out)


STATS = {"pass": 0, "seqlane": 0}


def block(s, terms, TPL=16):
    L = len(terms) // TPL
    if not np.any(terms != 0):
        return s
    start = 0
    while start < L:
        if s == 0 or not np.isfinite(s) or expo(s) < -100:   # no binade to work in: one lane sequentially, then try again
            s = seq_sum(s, terms[start * TPL:(start + 1) * TPL]); start += 1; STATS["seqlane"] += 1
            continue
        e = expo(s)
        scale, ulp = f32(2.0) ** f32(23 - e), f32(2.0) ** f32(e - 23)
        S = int(f32(s * scale))
        q = (terms * scale).astype(np.float32)
        STATS["pass"] += 1
        G, cur = [], None
        for l in range(start, L):
            fn = lane_fn(q[l * TPL:(l + 1) * TPL])
            cur = fn if cur is None else compose(cur, fn)
            G.append(cur)
        p = S & 1

        def valid(F):
            if F[6]:
                return False
            lo, hi = F[2 + 2 * p], F[3 + 2 * p]
            return (S + lo > 2 ** 23 and S + hi < 2 ** 24) if S > 0 else (S + hi < -2 ** 23 and S + lo > -2 ** 24)
        bad = [not valid(F) for F in G]
        if not any(bad):
            return f32(f32(S + G[-1][p]) * ulp)
        ls = bad.index(True)
        s = f32(f32(S + (G[ls - 1][p] if ls > 0 else 0)) * ulp)
        l = start + ls
        s = seq_sum(s, terms[l * TPL:(l + 1) * TPL]); STATS["seqlane"] += 1
        start = l + 1
    return s


def run(terms, B=1024):
    s = f32(0)
    for i in range(0, len(terms), B):
        blk = terms[i:i + B]
        if len(blk) < B:
            blk = np.concatenate([blk, np.zeros(B - len(blk), np.float32)])
        s = block(s, blk)
    return s


def crossings(t):
    acc, e = f32(0), []
    for x in t:
        acc = f32(acc + x)
        e.append(np.frexp(acc)[1])
    e = np.array(e)
    return np.nonzero(e[1:] != e[:-1])[0] + 1


def modelled_cycles_per_term(t, backoff_max, PASS=2200, SEQL=150, TPL=16, B=1024):
    chg, n, pos, cost, k = crossings(t), len(t), 0, 0, 1
    while pos < n:
        blk_end = min(n, (pos // B + 1) * B)
        j = bisect.bisect_left(chg, pos)
        nxt = chg[j] if j < len(chg) else n + 10
        cost += PASS
        if nxt >= blk_end:
            pos, k = blk_end, 1
        else:
            seq_end = min(blk_end, (nxt // TPL) * TPL + k * TPL)
            cost += SEQL * ((seq_end - (nxt // TPL) * TPL) // TPL)
            pos, k = seq_end, min(backoff_max, k * 2)
    return cost / n


if __name__ == "__main__":
    rng = np.random.default_rng(1)
    N = 60000
    tests = [("positive products", rng.normal(size=N).astype(f32) ** 2),
             ("signed products", (rng.normal(size=N) * rng.normal(size=N)).astype(f32)),
             ("wide dynamic range", (rng.normal(size=N) * np.exp(rng.normal(size=N) * 5)).astype(f32)),
             ("ties (halves)", rng.integers(-8, 9, size=N).astype(f32) * f32(0.5)),
             ("ties (powers of two)", rng.integers(-3, 4, size=N).astype(f32) * np.exp2(rng.integers(-6, 3, size=N)).astype(f32)),
             ("cancellation", np.concatenate([np.full(5000, 1.5, f32), np.full(5000, -1.5, f32), rng.normal(size=5000).astype(f32) * f32(1e-3)])),
             ("huge terms", np.concatenate([rng.normal(size=3000).astype(f32), np.array([1e30, -1e30, 3e38, 3e38], f32), rng.normal(size=3000).astype(f32)])),
             ("NaN", np.concatenate([rng.normal(size=3000).astype(f32), np.array([np.nan], f32), rng.normal(size=100).astype(f32)])),
             ("denormal terms", (rng.normal(size=N) * 1e-42).astype(f32)),
             ("tiny, then large", np.concatenate([(rng.normal(size=5000) * 1e-30).astype(f32), rng.normal(size=5000).astype(f32)]))]
    ok_all = True
    for name, t in tests:
        STATS["pass"] = STATS["seqlane"] = 0
        ref, got = seq_sum(f32(0), t), run(t)
        ok = (np.isnan(ref) and np.isnan(got)) or struct.pack("<f", float(ref)) == struct.pack("<f", float(got))
        ok_all &= ok
        print("%-22s %s  %d blocks, %d passes, %d sequential lanes" % (name, "bit-equal" if ok else "DIFFERENT", (len(t) + 1023) // 1024, STATS["pass"], STATS["seqlane"]))
    print("all bit-equal" if ok_all else "FAILURES")
    rng = np.random.default_rng(3)
    N = 250000
    a, b = rng.normal(size=N), rng.normal(size=N)
    for name, t in (("J J^T diagonal (squares)", (a * a).astype(f32)), ("products, correlation 0.3", (a * (0.3 * a + 0.95 * b)).astype(f32)),
                    ("products, correlation 0.05", (a * (0.05 * a + 0.999 * b)).astype(f32)), ("zero-mean products", (a * b).astype(f32))):
        print("%-28s %5d binade crossings in %d terms; modelled cycles per term: %.2f (restart only), %.2f (sequential back-off up to 64 lanes); sequential kernel 10.1"
              % (name, len(crossings(t)), N, modelled_cycles_per_term(t, 1), modelled_cycles_per_term(t, 64)))
