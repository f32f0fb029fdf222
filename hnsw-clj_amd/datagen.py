"""Mirror of the reference's synthetic data generator (test/data_generator.clj:28-87), vectorised.

``generate_dataset(size, dim, distribution=..., seed=42)`` reproduces, value for value, what the
Clojure generator draws from ``java.util.Random(seed)`` -- so a JVM user regenerates the identical
dataset.  java.util.Random is the JDK's documented 48-bit LCG; nextGaussian is the polar method with
StrictMath (fdlibm) log/sqrt.  The LCG is jumped ahead with composed affine maps so the whole stream
is produced by numpy array ops.  (Data preparation only: no distance arithmetic lives here.)
"""
import numpy as np

_MULT = np.uint64(0x5DEECE66D)
_ADD = np.uint64(0xB)
_MASK = np.uint64((1 << 48) - 1)


_TAB = {"n": 0, "A": None, "C": None}


def _lcg_stream(seed_state, count):
    """States after 1..count steps of s -> (a*s + c) mod 2^48."""
    if _TAB["n"] < count:
        n = max(count, 1 << 16)
        _TAB["A"], _TAB["C"] = _lcg_tables(n)
        _TAB["n"] = n
    with np.errstate(over="ignore"):
        return (_TAB["A"][1:count + 1] * np.uint64(seed_state) + _TAB["C"][1:count + 1]) & _MASK


def _lcg_tables(count):
    """A[k], C[k] with f^k(s) = A[k]*s + C[k] (mod 2^48) for k = 0..count."""
    A = np.empty(count + 1, np.uint64)
    Cc = np.empty(count + 1, np.uint64)
    A[0], Cc[0] = 1, 0
    if count >= 1:
        A[1], Cc[1] = _MULT, _ADD
    m = 1
    with np.errstate(over="ignore"):
        while m < count:
            # f^(m+j) = f^j o f^m  for j = 1..m
            j = min(m, count - m)
            A[m + 1:m + 1 + j] = (A[1:1 + j] * A[m]) & _MASK
            Cc[m + 1:m + 1 + j] = (A[1:1 + j] * Cc[m] + Cc[1:1 + j]) & _MASK
            m += j
    return A, Cc


def _fdlibm_log(x):
    """StrictMath.log (fdlibm __ieee754_log) for positive normal doubles, vectorised."""
    ln2_hi, ln2_lo = 6.93147180369123816490e-01, 1.90821492927058770002e-10
    Lg = (6.666666666666735130e-01, 3.999999999940941908e-01, 2.857142874366239149e-01, 2.222219843214978396e-01,
          1.818357216161805012e-01, 1.531383769920937332e-01, 1.479819860511658591e-01)
    x = np.ascontiguousarray(x, np.float64)
    u = x.view(np.uint64)
    hx = (u >> np.uint64(32)).astype(np.int64)
    assert np.all(hx >= 0x00100000) and np.all(hx < 0x7ff00000)
    k = (hx >> 20) - 1023
    hx = hx & 0x000fffff
    i = (hx + 0x95f64) & 0x100000
    hi = (hx | (i ^ 0x3ff00000)).astype(np.uint64)
    xm = ((u & np.uint64(0xffffffff)) | (hi << np.uint64(32))).view(np.float64)
    k = k + (i >> 20)
    f = xm - 1.0
    dk = k.astype(np.float64)
    # |f| < 2^-20 branch
    small = (0x000fffff & (2 + hx)) < 3
    Rs = f * f * (0.5 - 0.33333333333333333 * f)
    res_small = np.where(f == 0.0, np.where(k == 0, 0.0, dk * ln2_hi + dk * ln2_lo),
                         np.where(k == 0, f - Rs, dk * ln2_hi - ((Rs - dk * ln2_lo) - f)))
    s = f / (2.0 + f)
    z = s * s
    ii = hx - 0x6147a
    w = z * z
    jj = 0x6b851 - hx
    t1 = w * (Lg[1] + w * (Lg[3] + w * Lg[5]))
    t2 = z * (Lg[0] + w * (Lg[2] + w * (Lg[4] + w * Lg[6])))
    ii = ii | jj
    R = t2 + t1
    hfsq = 0.5 * f * f
    res_a = np.where(k == 0, f - (hfsq - s * (hfsq + R)), dk * ln2_hi - ((hfsq - (s * (hfsq + R) + dk * ln2_lo)) - f))
    res_b = np.where(k == 0, f - s * (f - R), dk * ln2_hi - ((s * (f - R) - dk * ln2_lo) - f))
    return np.where(small, res_small, np.where(ii > 0, res_a, res_b))


class JavaRandom:
    """java.util.Random with bulk draws."""

    def __init__(self, seed):
        self.state = (int(seed) ^ 0x5DEECE66D) & ((1 << 48) - 1)
        self._have = False
        self._next_g = 0.0

    def _states(self, count):
        st = _lcg_stream(self.state, count)
        if count:
            self.state = int(st[-1])
        return st

    def next_int(self, bound):
        bound = int(bound)
        r = int(self._states(1)[0] >> np.uint64(17))  # next(31)
        m = bound - 1
        if bound & m == 0:
            return (bound * r) >> 31
        u = r
        while True:
            r = u % bound
            if u - r + m < (1 << 31):  # no int overflow
                return r
            u = int(self._states(1)[0] >> np.uint64(17))

    def next_doubles(self, count):
        st = self._states(2 * count)
        hi = (st[0::2] >> np.uint64(22)).astype(np.int64)  # next(26)
        lo = (st[1::2] >> np.uint64(21)).astype(np.int64)  # next(27)
        return ((hi << 27) + lo).astype(np.float64) * (2.0 ** -53)

    def next_gaussians(self, count):
        """``count`` successive nextGaussian() values (honours the cached second variate)."""
        out = np.empty(count, np.float64)
        pos = 0
        if self._have and count > 0:
            out[0] = self._next_g
            self._have = False
            pos = 1
        while pos < count:
            need = min(count - pos, 1 << 19)  # block the stream so the temporaries stay cache-sized
            attempts = int((need + 1) // 2 * 1.35) + 8
            save = self.state
            d = self.next_doubles(2 * attempts)
            v1 = 2 * d[0::2] - 1
            v2 = 2 * d[1::2] - 1
            s = v1 * v1 + v2 * v2
            ok = (s < 1) & (s != 0)
            idx = np.flatnonzero(ok)
            pairs_needed = (need + 1) // 2
            if len(idx) >= pairs_needed:
                last = idx[pairs_needed - 1]
                # rewind to just after the last attempt that is actually consumed
                self.state = save
                self._states(4 * (int(last) + 1))
                idx = idx[:pairs_needed]
            sv = s[idx]
            mul = np.sqrt(-2 * _fdlibm_log(sv) / sv)
            g = np.empty(2 * len(idx), np.float64)
            g[0::2] = v1[idx] * mul
            g[1::2] = v2[idx] * mul
            take = min(need, len(g))
            out[pos:pos + take] = g[:take]
            pos += take
            if take < len(g):  # odd request: cache the second variate of the last pair
                self._have = True
                self._next_g = float(g[take])
        return out


def generate_dataset(size, dim, distribution="gaussian", num_clusters=10, noise_level=0.1, seed=42,
                     dtype=np.float32):
    """test/data_generator.clj:50-87 -> array (size, dim).  Values are computed in f64 exactly as the
    JVM would and rounded once to ``dtype`` (the engine stores float32)."""
    rng = JavaRandom(seed)
    if distribution == "gaussian":  # :28-31, :69
        out = rng.next_gaussians(size * dim).reshape(size, dim)
    elif distribution == "uniform":  # :71
        out = (2 * rng.next_doubles(size * dim) - 1).reshape(size, dim)
    elif distribution == "unit":  # :33-40: norm = sqrt(reduce + squares), left to right
        out = rng.next_gaussians(size * dim).reshape(size, dim)
        nrm = np.sqrt(np.cumsum(out * out, axis=1)[:, -1])
        nz = nrm != 0
        out[nz] = out[nz] / nrm[nz, None]
    elif distribution == "clustered":  # :42-48, :73-79 (nextInt is drawn before the noise of each vector)
        centers = rng.next_gaussians(num_clusters * dim).reshape(num_clusters, dim)
        out = np.empty((size, dim), np.float64)
        for i in range(size):
            c = rng.next_int(num_clusters)
            out[i] = centers[c] + noise_level * rng.next_gaussians(dim)
    else:
        raise ValueError("unknown :distribution %r" % (distribution,))
    return out.astype(dtype)


def indexed(vectors, prefix="vec_"):
    """:format :indexed (:84-86): [[\"vec_<i>\", vector], ...]"""
    return [[prefix + str(i), v] for i, v in enumerate(vectors)]


def generate_query_set(num_queries, dim, **opts):
    """test/data_generator.clj:170-173"""
    return generate_dataset(num_queries, dim, **opts)
