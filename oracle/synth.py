"""NumPy restatement of the on-device synthetic data generator (csrc/synth.hip + the column
statistics / standardisation of csrc/sweep.hip and api.hip: rbl_synth_local / rbl_synth_finish).

The generator reproduces the STATISTICS of the reference's synthetic branch
(src/util/load_data.py:101-116: sklearn make_classification defaults - 2 informative + 2 redundant
columns, the other d-4 columns N(0,1) noise, 2 clusters per class, class_sep, flip_y label noise,
shuffled columns - followed by preprocessing.scale), not sklearn's bits: BASELINE configs C2-C5 are
generated in HBM and never exist on the host.  It is counter based (Philox4x32-10 keyed by the seed,
indexed by (global row, column packet)), so any row range can be regenerated independently - which is
what lets the tests hand device-generated rows to the CPU oracle.

What is exact and what is not: the Philox stream, the labels, the cluster / flip draws and the positions
of the special columns are integer work and match the device bit for bit.  The Gaussian values go through
the device's fast float32 intrinsics (__logf, __sincosf), which NumPy's float32 log / sin / cos only match
to a few ulp: values agree to ~1e-6, the test states 2e-5.  Test infrastructure only.
"""
import numpy as np

_M0, _M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
_W0, _W1 = 0x9E3779B9, 0xBB67AE85
_MASK = np.uint64(0xFFFFFFFF)


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Philox4x32 with 10 rounds on arrays of uint32 counters (synth.hip: philox4x32_10)."""
    c0, c1, c2, c3 = (np.asarray(c, dtype=np.uint64) & _MASK for c in (c0, c1, c2, c3))
    k0, k1 = int(k0) & 0xFFFFFFFF, int(k1) & 0xFFFFFFFF
    for _ in range(10):
        p0, p1 = _M0 * c0, _M1 * c2
        hi0, lo0, hi1, lo1 = p0 >> np.uint64(32), p0 & _MASK, p1 >> np.uint64(32), p1 & _MASK
        c0, c1, c2, c3 = hi1 ^ c1 ^ np.uint64(k0), lo1, hi0 ^ c3 ^ np.uint64(k1), lo0
        k0, k1 = (k0 + _W0) & 0xFFFFFFFF, (k1 + _W1) & 0xFFFFFFFF
    return tuple(c.astype(np.uint32) for c in (c0, c1, c2, c3))


def _u01(x):
    # synth.hip: u01 - 24 high bits, centred: (0, 1)
    return ((x >> np.uint32(8)).astype(np.float32) + np.float32(0.5)) * np.float32(1.0 / 16777216.0)


def _box_muller(a, b):
    r = np.sqrt(np.float32(-2.0) * np.log(_u01(a)), dtype=np.float32)
    ang = np.float32(6.28318530717958647692) * _u01(b)
    return (r * np.cos(ang, dtype=np.float32)).astype(np.float32), (r * np.sin(ang, dtype=np.float32)).astype(np.float32)


def special_columns(seed, d):
    """positions of the 2 informative + 2 redundant columns, the 2x2 mixing matrix, the four clusters' 2x2
    covariance matrices A_k (row-major) and the hypercube vertex of each cluster (bit 0 / bit 1 = sign of the first /
    second informative coordinate; cluster k belongs to class k % 2, as in make_classification): the host-side LCG
    of api.hip: rbl_synth_local (identical on every rank)"""
    m64 = (1 << 64) - 1
    st = (seed * 6364136223846793005 + 1442695040888963407) & m64

    def nxt():
        nonlocal st
        st = (st * 6364136223846793005 + 1442695040888963407) & m64
        return (st >> 33) & 0xFFFFFFFF

    special = [-1, -1, -1, -1]
    for k in range(4 if d >= 4 else d):
        while True:
            c = nxt() % d
            if c not in special[:k]:
                special[k] = c
                break
    mix = [2.0 * (nxt() / 2147483648.0) - 1.0 for _ in range(4)]
    A = [2.0 * (nxt() / 2147483648.0) - 1.0 for _ in range(16)]
    vertex = [0, 1, 2, 3]
    for i in range(3, 0, -1):
        j = nxt() % (i + 1)
        vertex[i], vertex[j] = vertex[j], vertex[i]
    return special, np.array(mix, dtype=np.float32), np.array(A, dtype=np.float32).reshape(4, 4), vertex


def raw_rows(seed, d, row_lo, row_hi, class_sep=1.0, flip_y=0.01):
    """rows [row_lo, row_hi) of the RAW matrix (float32 values as k_synth computes them) and their labels"""
    ld = (d + 3) // 4 * 4
    packets = ld // 4
    rows = np.arange(row_lo, row_hi, dtype=np.uint64)
    n = rows.shape[0]
    k0, k1 = seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF
    glo, ghi = (rows & _MASK), (rows >> np.uint64(32))
    X = np.zeros((n, ld), dtype=np.float32)
    pk = np.arange(packets, dtype=np.uint64)
    r0, r1, r2, r3 = philox4x32_10(np.repeat(glo, packets), np.repeat(ghi, packets), np.tile(pk, n),
                                   np.ones(n * packets, dtype=np.uint64), k0, k1)
    x0, x1 = _box_muller(r0, r1)
    x2, x3 = _box_muller(r2, r3)
    X[:] = np.stack([x0, x1, x2, x3], axis=1).reshape(n, ld)
    # per-row draw: label, cluster, flip, informative noise
    q0, q1, q2, q3 = philox4x32_10(glo, ghi, np.full(n, 0xFFFFFFFF, dtype=np.uint64), np.zeros(n, dtype=np.uint64), k0, k1)
    y01 = (q0 & np.uint32(1)).astype(np.int64)
    cl = ((q0 >> np.uint32(1)) & np.uint32(1)).astype(np.int64)
    ylab = np.where(_u01(q1) < np.float32(flip_y), ((q0 >> np.uint32(2)) & np.uint32(1)).astype(np.int64), y01)
    special, mix, A, vertex = special_columns(seed, d)
    g0, g1 = _box_muller(q2, q3)
    cs = np.float32(class_sep)
    c = (cl << 1) | y01                                    # cluster: class = c % 2
    vx = np.array(vertex)
    cen0 = cs * np.where(vx[c] & 1, np.float32(1), np.float32(-1)).astype(np.float32)
    cen1 = cs * np.where(vx[c] & 2, np.float32(1), np.float32(-1)).astype(np.float32)
    f0 = (g0 * A[c, 0] + g1 * A[c, 2] + cen0).astype(np.float32)
    f1 = (g0 * A[c, 1] + g1 * A[c, 3] + cen1).astype(np.float32)
    feat = [f0, f1, (f0 * mix[0] + f1 * mix[2]).astype(np.float32), (f0 * mix[1] + f1 * mix[3]).astype(np.float32)]
    for k in range(4):
        if special[k] >= 0:
            X[:, special[k]] = feat[k]
    X[:, d:] = 0
    return X[:, :d], (2 * ylab - 1).astype(np.float64)


def standardized_D(seed, d, n_total, row_lo=0, row_hi=None, class_sep=1.0, flip_y=0.01, storage="f64", chunk=200_000):
    """D = -y * scale(X) for rows [row_lo, row_hi) of the n_total-row problem: column mean / population std over
    ALL n_total rows (preprocessing.scale, load_data.py:115), fp64 sums as on the device; `storage` rounds
    the raw and the final values to the device's storage type."""
    row_hi = n_total if row_hi is None else row_hi
    st = np.float32 if storage == "f32" else np.float64
    s1, s2 = np.zeros(d), np.zeros(d)
    for lo in range(0, n_total, chunk):
        X, _ = raw_rows(seed, d, lo, min(lo + chunk, n_total), class_sep, flip_y)
        X = X.astype(st).astype(np.float64)
        s1 += X.sum(axis=0)
        s2 += (X * X).sum(axis=0)
    mean = s1 / n_total
    var = s2 / n_total - mean * mean
    var = np.where(var > 0, var, 1.0)
    X, y = raw_rows(seed, d, row_lo, row_hi, class_sep, flip_y)
    X = X.astype(st).astype(np.float64)
    D = (-y[:, None] * ((X - mean) * (1.0 / np.sqrt(var)))).astype(st).astype(np.float64)
    return D, y
