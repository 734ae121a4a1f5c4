"""Bank-conflict model of the LDS accesses of acq_corr2500_kernel (64 banks x 4 bytes, 8-byte accesses served 32 lanes at a time):
cycles per instruction = max over banks of the distinct dwords a 32-lane group asks of it.  Used to choose layouts (development aid)."""
import itertools, sys
import numpy as np
def cycles(addrs_f2, active=None):
    """addrs_f2: float2 index per lane (64 lanes; None = inactive)"""
    tot = 0
    for g in range(0, len(addrs_f2), 32):
        banks = {}
        for a in addrs_f2[g:g + 32]:
            if a is None: continue
            for d in (2 * a, 2 * a + 1):
                banks.setdefault(d % 64, set()).add(d)
        tot += max((len(v) for v in banks.values()), default=0)
    return tot
def sim(SS=281, P=25, map3="t2fast", PA=25):
    T = 250; total = 0; ideal = 0; per = {}
    lanes = list(range(256))
    def run(name, fn, n):      # fn(t, i) -> float2 index or None, for i < n instructions
        nonlocal total, ideal
        c = 0; idl = 0
        for i in range(n):
            for w in range(4):
                ad = [fn(t, i) if t < T else None for t in range(64 * w, 64 * w + 64)]
                c += cycles(ad); idl += sum(1 for g in (0, 32) if any(a is not None for a in ad[g:g + 32]))
        per[name] = (c, idl); total += c; ideal += idl
    run("p1w", lambda t, k: k * SS + t, 10)
    run("p2r", lambda t, q: (t // 25) * SS + (t % 25) + 25 * q, 10)
    run("p2w", lambda t, k: (t // 25) * SS + k * P + (t % 25), 10)
    def m3(t, h):
        g = t + 250 * h
        if map3 == "t2fast": return g // 5, g % 5
        return g % 100, g // 100        # sq fastest
    for h in range(2):
        run("p3r%d" % h, lambda t, q, h=h: ((m3(t, h)[0] // 10) * SS + (m3(t, h)[0] % 10) * P + m3(t, h)[1] + 5 * q), 5)
        run("p3w%d" % h, lambda t, k, h=h: (m3(t, h)[0] * PA + k * 5 + m3(t, h)[1]), 5)
        run("p4r%d" % h, lambda t, q, h=h: (((t + 250 * h) // 5) * PA + ((t + 250 * h) % 5) * 5 + q), 5)
    return total, ideal, per
if __name__ == "__main__":
    for kw in (dict(), dict(map3="sqfast"), dict(P=27), dict(P=26), dict(P=28), dict(P=27, PA=27), dict(PA=27), dict(PA=26), dict(P=27, map3="sqfast")):
        t, i, per = sim(**kw)
        print(kw, "cycles", t, "ideal", i, "conflict share %.2f" % (1 - i / t), {k: v[0] - v[1] for k, v in per.items() if v[0] != v[1]})

def search():
    """pass 3 / pass 4 only: lane g in [0, 500) -> digits (t2 in 0..4, k1 in 0..9, k2 in 0..9) in any digit order; B layout k1 SS + k2 P + t1
    (t1 = t2 + 5 q), A layout for pass 3 -> 4: sq PA + k3 5 + t2 (pass 4 reads sq PA + k3a 5 + q with its own lane order)."""
    import itertools
    T = 250
    def decomp(g, order):      # order: tuple of (name, radix) from fastest to slowest
        out = {}
        for name, r in order:
            out[name] = g % r; g //= r
        return out
    orders = [p for p in itertools.permutations((("t", 5), ("k1", 10), ("k2", 10)))]
    best = []
    for SS in (281,):
        for P in (25, 26, 27, 28):
            for PA in (25, 26, 27, 28, 29, 30, 31):
                for o3 in orders:
                    for o4 in orders:
                        tot = 0; idl = 0
                        for h in range(2):
                            for w in range(4):
                                lanes = [decomp(t + 250 * h, o3) if t < T else None for t in range(64 * w, 64 * w + 64)]
                                for q in range(5):
                                    ad = [None if d is None else d["k1"] * SS + d["k2"] * P + d["t"] + 5 * q for d in lanes]
                                    tot += cycles(ad); idl += sum(1 for g in (0, 32) if any(a is not None for a in ad[g:g + 32]))
                                    ad = [None if d is None else (10 * d["k1"] + d["k2"]) * PA + q * 5 + d["t"] for d in lanes]
                                    tot += cycles(ad); idl += sum(1 for g in (0, 32) if any(a is not None for a in ad[g:g + 32]))
                                lanes = [decomp(t + 250 * h, o4) if t < T else None for t in range(64 * w, 64 * w + 64)]
                                for q in range(5):
                                    ad = [None if d is None else (10 * d["k1"] + d["k2"]) * PA + d["t"] * 5 + q for d in lanes]
                                    tot += cycles(ad); idl += sum(1 for g in (0, 32) if any(a is not None for a in ad[g:g + 32]))
                        best.append((tot - idl, P, PA, [n for n, _ in o3], [n for n, _ in o4]))
    best.sort(key=lambda x: x[0])
    for b in best[:12]: print(b)
    print("current:", [b for b in best if b[1] == 25 and b[2] == 25 and b[3] == ["t", "k2", "k1"] and b[4] == ["t", "k2", "k1"]])
if len(sys.argv) > 1 and sys.argv[1] == "search": search()

def search_pad():
    import itertools
    T = 250
    def decomp(g, order):
        out = {}
        for name, r in order:
            out[name] = g % r; g //= r
        return out
    orders = [p for p in itertools.permutations((("t", 5), ("k1", 10), ("k2", 10)))]
    pads = [(m, c) for m in (5, 8, 10, 16, 25, 32, 50, 64) for c in (0, 1, 2, 3)]
    res = []
    for (mb, cb) in pads:
        padB = lambda i: i + (i // mb) * cb
        if padB(249) >= 281: continue
        for (ma, ca) in pads:
            padA = lambda i: i + (i // ma) * ca
            for o3 in orders:
                # pass 4 keeps its order (t fastest: the thread's outputs stay where the accumulators expect them)
                tot = 0; idl = 0
                for h in range(2):
                    for w in range(4):
                        lanes = [decomp(t + 250 * h, o3) if t < T else None for t in range(64 * w, 64 * w + 64)]
                        for q in range(5):
                            ad = [None if d is None else d["k1"] * 281 + padB(d["k2"] * 25 + d["t"] + 5 * q) for d in lanes]
                            tot += cycles(ad); idl += sum(1 for g in (0, 32) if any(a is not None for a in ad[g:g + 32]))
                            ad = [None if d is None else padA((10 * d["k1"] + d["k2"]) * 25 + q * 5 + d["t"]) for d in lanes]
                            tot += cycles(ad); idl += sum(1 for g in (0, 32) if any(a is not None for a in ad[g:g + 32]))
                        lanes = [decomp(t + 250 * h, orders[0]) if t < T else None for t in range(64 * w, 64 * w + 64)]
                        for q in range(5):
                            ad = [None if d is None else padA((10 * d["k1"] + d["k2"]) * 25 + d["t"] * 5 + q) for d in lanes]
                            tot += cycles(ad); idl += sum(1 for g in (0, 32) if any(a is not None for a in ad[g:g + 32]))
                res.append((tot - idl, (mb, cb), (ma, ca), [n for n, _ in o3]))
    res.sort(key=lambda x: x[0])
    for b in res[:10]: print(b)
    # pass 2 writes with the padded B layout must stay conflict-free too: check the best
if len(sys.argv) > 1 and sys.argv[1] == "pad": search_pad()
