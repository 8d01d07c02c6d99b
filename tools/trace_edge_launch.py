"""Where the time of a slab's edge launch goes: start, end and placement of every wave.

Needs a build of mgx_stream.hip with -DMGX_STREAM_TRACE (tools/build_variant.sh edgetrace
"-DMGX_STREAM_TRACE"), then
   MGX_LIB=$PWD/variants/libmgx_edgetrace.so python tools/trace_edge_launch.py [--ranks 8] [--rank 3]
The slab driver is built for rank g of R with a communicator that moves nothing
(tools/rank_share.py); the marks are those of the last two-range launch."""
import argparse
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import torch

from multigrid_amd import _lib, slab
from rank_share import QuietComm


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=16385)
    ap.add_argument("--ranks", type=int, default=8)
    ap.add_argument("--rank", type=int, default=3)
    ap.add_argument("--interior", action="store_true", help="the one-range launch (a slab's interior columns; with "
                                                            "--ranks 1 the whole grid) instead of the edge bands")
    args = ap.parse_args()
    L = _lib.load()
    L.mgx_debug_stream_trace.argtypes = [ctypes.c_int, ctypes.c_void_p]
    M = args.size
    if args.ranks > 1:
        slab._Comm = lambda group: QuietComm(args.rank, args.ranks)
        S = slab.SlabSolver(M, 1.0 / (M - 1), 2)
        S.set_random_rhs()
    else:
        import multigrid_amd as mg
        from multigrid_amd import rhs
        S = mg.Plan(M, 1.0 / (M - 1), 2)
        S.set(rhs.random_torch(M), mg.colmajor(M, M))
    S.tune(2, 2, 0.6)
    S.cycles(3, 2, 2, 0.6)
    torch.cuda.synchronize()
    L.mgx_debug_stream_trace(1 if args.interior else 2, None)
    S.cycles(3, 2, 2, 0.6)
    torch.cuda.synchronize()
    n = 16384
    buf = (ctypes.c_ulonglong * (4 * n))()
    L.mgx_debug_stream_trace(0, buf)
    t = np.frombuffer(buf, dtype=np.uint64).reshape(n, 4).astype(np.int64)
    on = t[:, 3] != 0
    t = t[on]
    if not len(t):
        print("no marks")
        return
    t0 = t[:, 0].min()
    st, en = (t[:, 0] - t0) / 100.0, (t[:, 1] - t0) / 100.0
    dur = en - st
    hw, xcc = t[:, 2] & 0xffffffff, (t[:, 2] >> 32) & 0xf
    simd, cu, sh, se = (hw >> 4) & 3, (hw >> 8) & 15, (hw >> 12) & 1, (hw >> 13) & 7
    where = xcc * 100000 + se * 10000 + sh * 1000 + cu * 10 + simd
    kinds = t[:, 3] & 0xff
    width = (t[:, 3] >> 8) & 0xff          # modulo 256: the interior's chunks are longer
    strip, chunk = (t[:, 3] >> 16) & 0xffff, (t[:, 3] >> 32) & 0xffff
    print("waves %d, launch spans %.1f us (first start to last end)" % (len(t), en.max()))
    pct = lambda a: "min %.1f  p10 %.1f  median %.1f  p90 %.1f  max %.1f" % (a.min(), np.percentile(a, 10), np.median(a),
                                                                              np.percentile(a, 90), a.max())
    print("start    ", pct(st))
    print("duration ", pct(dur))
    print("end      ", pct(en))
    u, c = np.unique(where, return_counts=True)
    print("SIMDs used %d; waves per SIMD: %s" % (len(u), dict(zip(*np.unique(c, return_counts=True)))))
    cuid = where // 10
    u2, c2 = np.unique(cuid, return_counts=True)
    print("CUs used %d; waves per CU: %s" % (len(u2), dict(zip(*np.unique(c2, return_counts=True)))))
    for k in np.unique(kinds):
        m = kinds == k
        print("kind %d (1 plain, +1 row edge, +2 column tests): %d waves, median duration %.1f us, columns %s"
              % (k, m.sum(), np.median(dur[m]), dict(zip(*np.unique(width[m], return_counts=True)))))
    for w in np.unique(width):
        m = width == w
        print("chunks of %d columns: %d waves, median duration %.1f us (%.2f us per column produced)"
              % (w, m.sum(), np.median(dur[m]), np.median(dur[m]) / max(w, 1)))
    for name, key in (("XCD", xcc), ("chunk", chunk), ("strip group of 16", strip // 16)):
        print("median / max duration by %s: %s" % (name, {int(k): (round(float(np.median(dur[key == k])), 1),
                                                                    round(float(dur[key == k].max()), 1)) for k in np.unique(key)}))
    per_simd = dict(zip(u, c))
    share_ = np.array([per_simd[x] for x in where])
    print("by chunk: waves, share of them on a SIMD with company, median duration alone / with company")
    for k in np.unique(chunk):
        m = chunk == k
        a, b = dur[m & (share_ == 1)], dur[m & (share_ > 1)]
        print("   chunk %2d: %4d  %.2f   %s / %s" % (k, m.sum(), (share_[m] > 1).mean(), "%.1f" % np.median(a) if len(a) else "-",
                                                  "%.1f" % np.median(b) if len(b) else "-"))
    print("CUs with two workgroups: (chunk, first strip, median duration) of each")
    rows = []
    for cid in u2[c2 >= 8]:
        m = np.where(cuid == cid)[0]
        groups = {}
        for i in m:
            groups.setdefault((int(chunk[i]), int(strip[i]) // 4), []).append(dur[i])
        rows.append(sorted((k[0], k[1] * 4, round(float(np.median(v)), 1)) for k, v in groups.items()))
    rows.sort(key=lambda r: -max(x[2] for x in r))
    for r in rows[:12] + rows[-12:]:
        print("   ", r)
    slow = np.argsort(-dur)[:8]
    print("slowest waves (duration, strip, chunk, columns, xcd, cu, simd, waves on its CU):")
    percu = dict(zip(u2, c2))
    for i in slow:
        print("   %.1f us  strip %d chunk %d cols %d  xcd %d se %d cu %d simd %d  (%d)" % (dur[i], strip[i], chunk[i], width[i], xcc[i], se[i],
                                                                                      cu[i], simd[i], percu[cuid[i]]))
    # how long a wave takes against how many waves shared its SIMD
    per = dict(zip(u, c))
    share = np.array([per[x] for x in where])
    for k in np.unique(share):
        m = share == k
        print("waves on a SIMD with %d of them: %d, median duration %.1f us, median end %.1f us" % (k, m.sum(), np.median(dur[m]),
                                                                                                  np.median(en[m])))


if __name__ == "__main__":
    main()
