#!/usr/bin/env python3
"""One-off stress of the HIP path against the oracle: many random configurations with arrays large enough
for multi-tile workgroup ranges (the regime the per-commit tests touch only in a few fixed cases).
  python tests/stress_parity.py [--cases 150] [--max-n 6000000] [--seed 1]
Exits non-zero at the first mismatch."""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=150)
    ap.add_argument("--max-n", type=int, default=6_000_000)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--nd-cases", type=int, default=60, help="random 2-D / 3-D shapes through the tile mode (in place and gather paths)")
    ap.add_argument("--split", type=int, default=0, help="1 / 3: flat fp64 arrays on the chain of kernels go through k_compress_eo (lists / single-pass "
                    "placement); the one-launch kernels are turned off so that every array takes the chain")
    a = ap.parse_args()
    import numpy as np
    import torch
    import dctz_amd
    from oracle import oracle as O

    rng = np.random.default_rng(a.seed)
    ctx = dctz_amd.Context(0)
    ctx.set_speculation(True, 1 << 18)
    if a.split:
        ctx.set_one_launch(False)
        ctx.set_split(a.split)
    t0 = time.time()
    flags_seen = {}
    for k in range(a.cases):
        n = int(rng.integers(1, a.max_n)) if rng.random() < 0.7 else int(rng.integers(1, 5000))
        dtype = np.float64 if rng.random() < 0.6 else np.float32
        mode = O.QT if rng.random() < 0.4 else O.EC
        eb = float(rng.choice([1e-2, 1e-3, 1e-4, 1e-5, 1e-6]))
        amp = 10.0 ** rng.uniform(-5, 7)
        noise = float(rng.choice([0.0, 1e-5, 1e-3, 0.05, 1.0]))
        t = np.arange(n) / rng.uniform(5, 500)
        x = amp * (np.sin(t) + 0.3 * np.cos(4.7 * t + 1.0)) + noise * amp * rng.standard_normal(n)
        if rng.random() < 0.3:
            x[rng.random(n) < rng.uniform(0.001, 0.5)] = 0.0
        if rng.random() < 0.15 and n > 10:
            x[rng.integers(0, n, size=3)] *= 1e3            # spikes: wrong sampled decade now and then
        x = x.astype(dtype)
        c = O.compress(x, eb, mode, O.FAST)
        ref = O.decompress(c, O.FAST)
        xd = torch.from_numpy(x).cuda()
        # every other case also asks for the scaled copy in a buffer of its own: the variant of k_compress that writes x / sf back
        sc = torch.empty_like(xd) if k % 2 else None
        out, info = ctx.compress(xd, eb, mode, scaled=sc)
        flags_seen[info.flags] = flags_seen.get(info.flags, 0) + 1
        it = np.uint64 if dtype == np.float64 else np.uint32
        if sc is not None and not np.array_equal(sc.cpu().numpy().view(it), c.scaled.view(it)):
            print(f"MISMATCH (scaled copy) case {k}: n={n} dtype={dtype.__name__} mode={mode} eb={eb} amp={amp:g}")
            sys.exit(1)
        ok = (info.sf == c.sf and info.cnt == c.cnt and np.array_equal(out["bin_index"].cpu().numpy(), c.bin_index)
              and np.array_equal(out["dc"].cpu().numpy().view(np.uint32), c.dc.view(np.uint32))
              and np.array_equal(out["ac_exact"][:c.cnt].cpu().numpy().view(np.uint32), c.ac_exact.view(np.uint32)))
        tdt = torch.float64 if dtype == np.float64 else torch.float32
        for _ in range(2):
            r = ctx.decompress(out, info.cnt, n, tdt, eb, info.sf, mode, qtable=np.array(info.qtable[:])).cpu().numpy()
            ok = ok and np.array_equal(r.view(it), ref.view(it))
        # the entropy stage on the same streams: zlib's inflate is the checker; every other case also through the device decoder
        import zlib
        secs = [out["bin_index"], out["dc"], out["ac_exact"][:c.cnt]]
        zs, index = ctx.deflate(secs, want_index=True)
        for z, want in zip(zs, (c.bin_index, c.dc, c.ac_exact)):
            ok = ok and zlib.decompress(z.cpu().numpy().tobytes()) == want.tobytes()
        if k % 2 == 0:
            back, good = ctx.inflate(zs, index, [t.numel() * t.element_size() for t in secs])
            ok = ok and good and all(torch.equal(b, t.view(torch.uint8).reshape(-1)[:b.numel()]) for b, t in zip(back, secs))
        if not ok:
            print(f"MISMATCH case {k}: n={n} dtype={dtype.__name__} mode={mode} eb={eb} amp={amp:g} noise={noise}")
            sys.exit(1)
        if k % 10 == 9:
            print(f"{k + 1} cases ok, {time.time() - t0:.0f} s, flags {flags_seen}", flush=True)
    print(f"all {a.cases} cases bit-exact; statistics paths seen (flags -> count): {flags_seen}")
    # multi-dimensional blocks: random shapes, a third of them with every extent a multiple of the tile edge (read and
    # written in place), the rest ragged (gather / scatter passes)
    nd_seen = {"in_place": 0, "gather": 0}
    for k in range(a.nd_cases):
        nd = int(rng.integers(2, 4))
        e = 8 if nd == 2 else 4
        cap = 3_000_000
        while True:
            shape = tuple(int(rng.integers(1, 2000 if nd == 2 else 200)) for _ in range(nd))
            if rng.random() < 0.4:
                shape = tuple(max(e, (d // e) * e) for d in shape)
            if int(np.prod(shape)) <= cap:
                break
        dtype = np.float64 if rng.random() < 0.5 else np.float32
        mode = O.QT if rng.random() < 0.4 else O.EC
        eb = float(rng.choice([1e-2, 1e-3, 1e-4, 1e-5]))
        amp = 10.0 ** rng.uniform(-3, 4)
        axes = np.meshgrid(*[np.linspace(0, rng.uniform(1, 9), d) for d in shape], indexing="ij")
        x = amp * (np.sin(axes[0] * 3.1 + 0.2) * np.cos(axes[1] * 2.3) + (0.3 * np.sin(axes[2] * 5.0) if nd == 3 else 0.0)
                   + float(rng.choice([0.0, 1e-4, 0.02, 0.5])) * rng.standard_normal(shape))
        x = np.ascontiguousarray(x.astype(dtype))
        c = O.compress_nd(x, eb, mode, O.FAST)
        ref = O.decompress_nd(c, shape, O.FAST)
        out, info = ctx.compress_nd(torch.from_numpy(x).cuda(), eb, mode)
        nd_seen["in_place" if all(d % e == 0 for d in shape) else "gather"] += 1
        tdt = torch.float64 if dtype == np.float64 else torch.float32
        r = ctx.decompress_nd(out, info.cnt, shape, tdt, eb, info.sf, mode, qtable=np.array(info.qtable[:])).cpu().numpy()
        it = np.uint64 if dtype == np.float64 else np.uint32
        ok = (info.sf == c.sf and info.cnt == c.cnt and np.array_equal(out["bin_index"].cpu().numpy(), c.bin_index)
              and np.array_equal(out["dc"].cpu().numpy().view(np.uint32), c.dc.view(np.uint32))
              and np.array_equal(out["ac_exact"][:c.cnt].cpu().numpy().view(np.uint32), c.ac_exact.view(np.uint32))
              and np.array_equal(r.view(it), ref.view(it)))
        if not ok:
            print(f"MISMATCH nd case {k}: shape={shape} dtype={dtype.__name__} mode={mode} eb={eb} amp={amp:g}")
            sys.exit(1)
    print(f"all {a.nd_cases} multi-dimensional cases bit-exact ({nd_seen})")


if __name__ == "__main__":
    main()
