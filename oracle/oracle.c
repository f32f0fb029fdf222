/*
 * oracle.c -- CPU restatement of hnsw-clj's distance / HNSW-search / IVF-FLAT hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under hnsw-clj_amd/ (the product) may link, import or
 * call this file.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it,
 * and there only as the checker / the CPU baseline, never as the thing shipped.
 *
 * Parity status: the reference is Clojure on the JVM and cannot be run in this image (no JVM).
 * This restatement is pinned against every numeric known-answer test the reference's own test
 * suite holds for this path (test/hnsw/core_test.clj:9-31, test/simple_test.clj:33-41,
 * test/hnsw/graph_test.clj:11-22, test-functional.sh:53-71) -- see tests/test_oracle_kat.py.
 * The reference holds NO golden top-k lists, so search-result parity against the real JVM is
 * unpinned beyond those KATs ("parity partially pinned"); DESIGN.md says the same.
 *
 * Every function cites the reference file:line it follows (paths relative to /root/reference).
 * Compile: gcc -O2 -ffp-contract=off -fno-fast-math (see oracle/Makefile).  JVM semantics are
 * IEEE-754 binary64, no FMA contraction, no re-association, so plain C loops reproduce them.
 */
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_COSINE 0
#define ORC_L2 1
#define ORC_DOT 2

/* arithmetic modes for the search/scan drivers */
#define ORC_MODE_F64 0  /* reference order, f64 (the truth)                                  */
#define ORC_MODE_DEV 1  /* bit-mimic of the HIP kernels' f32 lane/butterfly order            */
#define ORC_MODE_FAST 2 /* "fair-fight" CPU baseline: f32, 8 partial sums, precomputed norms */
#define ORC_MODE_MFMA 3 /* bit-mimic of the tiled MFMA scan path (tile_kernels.hpp); norms as DEV      */

/* ------------------------------------------------------------------------------------------ */
/* 1. Distances, f64, strictly left-to-right                                                  */
/* ------------------------------------------------------------------------------------------ */

/* src/hnsw/ultra_fast.clj:53-95  cosine-distance-ultra.  The 4x unrolled `(+ dot p0 p1 p2 p3)`
 * expands left-associatively, so the sums equal the plain loop's.  Zero guard :92-95.
 * Length = (alength v1). */
double orc_cosine_ultra(const double *a, const double *b, int n) {
    double dot = 0.0, n1 = 0.0, n2 = 0.0;
    for (int i = 0; i < n; i++) {
        dot = dot + a[i] * b[i];
        n1 = n1 + a[i] * a[i];
        n2 = n2 + b[i] * b[i];
    }
    if (n1 > 0.0 && n2 > 0.0) return 1.0 - dot / (sqrt(n1) * sqrt(n2));
    return 1.0;
}

/* src/hnsw/simd.clj:129-147  cosine-distance-direct -- what simd-optimized/cosine-distance
 * (simd_optimized.clj:145-153) resolves to for double[] when the Vector API is present.
 * Same sums; guard is (zero? magnitude). */
double orc_cosine_direct(const double *a, const double *b, int n) {
    double dot = 0.0, na = 0.0, nb = 0.0;
    for (int i = 0; i < n; i++) {
        dot = dot + a[i] * b[i];
        na = na + a[i] * a[i];
        nb = nb + b[i] * b[i];
    }
    double mag = sqrt(na) * sqrt(nb);
    if (mag == 0.0) return 1.0;
    return 1.0 - dot / mag;
}

/* src/hnsw/ultra_fast.clj:43-51 euclidean-distance-ultra == src/hnsw/simd.clj:149-160.
 * NOTE: rooted distance. */
double orc_euclid(const double *a, const double *b, int n) {
    double s = 0.0;
    for (int i = 0; i < n; i++) {
        double d = a[i] - b[i];
        s = s + d * d;
    }
    return sqrt(s);
}

/* src/hnsw/simd_optimized.clj:283-293 dot-product (fallback form; plain loop). */
double orc_dot(const double *a, const double *b, int n) {
    double s = 0.0;
    for (int i = 0; i < n; i++) s = s + a[i] * b[i];
    return s;
}

/* Same formulas on float32-valued inputs widened to f64 (the GPU stores f32; a JVM user would
 * hold exactly these values as doubles).  metric DOT distance := -dot (API definition of this
 * build; the reference has no index that uses dot as a metric -- SURVEY.md section 8 a3). */
static inline double dist_f64(int metric, const float *a, const float *b, int n) {
    if (metric == ORC_L2) {
        double s = 0.0;
        for (int i = 0; i < n; i++) {
            double d = (double)a[i] - (double)b[i];
            s = s + d * d;
        }
        return sqrt(s);
    } else if (metric == ORC_DOT) {
        double s = 0.0;
        for (int i = 0; i < n; i++) s = s + (double)a[i] * (double)b[i];
        return -s;
    } else {
        double dot = 0.0, n1 = 0.0, n2 = 0.0;
        for (int i = 0; i < n; i++) {
            double x = (double)a[i], y = (double)b[i];
            dot = dot + x * y;
            n1 = n1 + x * x;
            n2 = n2 + y * y;
        }
        if (n1 > 0.0 && n2 > 0.0) return 1.0 - dot / (sqrt(n1) * sqrt(n2));
        return 1.0;
    }
}
double orc_dist_f32in(int metric, const float *a, const float *b, int n) { return dist_f64(metric, a, b, n); }

/* ------------------------------------------------------------------------------------------ */
/* 2. Device-order f32 mimic (bit-for-bit what the HIP kernels compute; hnsw-clj_amd/csrc)     */
/*    lane l of a 64-lane wavefront accumulates elements {256c + 4l + j}, c ascending, j=0..3, */
/*    with one fmaf per element; the 64 partials are combined by an xor butterfly with offsets */
/*    1,2,4,8,16,32 (s[l] = s[l] + s[l^off]).                                                  */
/* ------------------------------------------------------------------------------------------ */
static inline float dev_reduce64(float *s) {
    float t[64];
    for (int off = 1; off < 64; off <<= 1) {
        for (int l = 0; l < 64; l++) t[l] = s[l] + s[l ^ off];
        memcpy(s, t, sizeof(t));
    }
    return s[0];
}
static inline float dev_dot(const float *q, const float *v, int n) {
    float s[64];
    for (int l = 0; l < 64; l++) {
        float acc = 0.0f;
        for (int base = 4 * l; base < n; base += 256)
            for (int j = 0; j < 4 && base + j < n; j++) acc = fmaf(q[base + j], v[base + j], acc);
        s[l] = acc;
    }
    return dev_reduce64(s);
}
static inline float dev_l2sq(const float *q, const float *v, int n) {
    float s[64];
    for (int l = 0; l < 64; l++) {
        float acc = 0.0f;
        for (int base = 4 * l; base < n; base += 256)
            for (int j = 0; j < 4 && base + j < n; j++) {
                float d = q[base + j] - v[base + j];
                acc = fmaf(d, d, acc);
            }
        s[l] = acc;
    }
    return dev_reduce64(s);
}
float orc_norm_dev(const float *v, int n) { return sqrtf(dev_dot(v, v, n)); }

/* Tiled path (hnsw-clj_amd/csrc/tile_kernels.hpp): v_mfma_f32_32x32x2_f32 is a k-ordered f32 fmaf chain;
 * the kernel feeds k in the order 8t+j, 8t+4+j (j = 0..3, t ascending). */
static inline float mfma_dot(const float *q, const float *v, int n) {
    float acc = 0.0f;
    for (int t = 0; 8 * t < n; t++)
        for (int j = 0; j < 4; j++) {
            int k0 = 8 * t + j, k1 = 8 * t + 4 + j;
            if (k0 < n) acc = fmaf(v[k0], q[k0], acc);
            if (k1 < n) acc = fmaf(v[k1], q[k1], acc);
        }
    return acc;
}
float orc_dist_mfma(int metric, const float *q, const float *v, int n, float qnorm, float vnorm) {
    float dot = mfma_dot(q, v, n);
    if (metric == ORC_DOT) return -dot + 0.0f;
    if (qnorm > 0.0f && vnorm > 0.0f) return 1.0f - dot / (qnorm * vnorm) + 0.0f;
    return 1.0f;
}
/* qnorm / vnorm are orc_norm_dev values (cosine only). */
float orc_dist_dev(int metric, const float *q, const float *v, int n, float qnorm, float vnorm) {
    if (metric == ORC_L2) return sqrtf(dev_l2sq(q, v, n));
    float dot = dev_dot(q, v, n);
    if (metric == ORC_DOT) return -dot;
    if (qnorm > 0.0f && vnorm > 0.0f) return 1.0f - dot / (qnorm * vnorm);
    return 1.0f;
}

/* "fair-fight" f32: 16 independent partial sums (no re-association needed to vectorise them), norms precomputed.
 * The file is built -O2, where gcc 11 does not run the vectoriser: these two functions ask for it themselves and are
 * cloned per ISA (resolved once at load time), so the CPU baseline's f32 variant really is AVX2 / AVX-512 code on the
 * GPU box's host cores (`gcc -fopt-info-vec` reports both loops vectorised; round 1's 8-sum form ran scalar). */
#define ORC_VEC __attribute__((optimize("O3", "tree-vectorize"), target_clones("avx512f", "avx2", "default"), noinline))
ORC_VEC static float fast_dot(const float *a, const float *b, int n) {
    float s[16] = {0};
    int i = 0;
    for (; i + 16 <= n; i += 16)
        for (int j = 0; j < 16; j++) s[j] += a[i + j] * b[i + j];
    float t = 0.0f;
    for (; i < n; i++) t += a[i] * b[i];
    for (int w = 8; w >= 1; w >>= 1)
        for (int j = 0; j < w; j++) s[j] += s[j + w];
    return s[0] + t;
}
ORC_VEC static float fast_l2sq(const float *a, const float *b, int n) {
    float s[16] = {0};
    int i = 0;
    for (; i + 16 <= n; i += 16)
        for (int j = 0; j < 16; j++) {
            float d = a[i + j] - b[i + j];
            s[j] += d * d;
        }
    float t = 0.0f;
    for (; i < n; i++) {
        float d = a[i] - b[i];
        t += d * d;
    }
    for (int w = 8; w >= 1; w >>= 1)
        for (int j = 0; j < w; j++) s[j] += s[j + w];
    return s[0] + t;
}

/* ------------------------------------------------------------------------------------------ */
/* 2b. The reference's float32 Vector-API forms -- reached only from     */
/*     P-HNSW/PCAF (ann/dimreduct/pcaf.clj:226,243); they document the reference's OWN f32        */
/*     semantics: hardware-dependent summation order, hence the 1e-4 tolerance of north_star.     */
/* ------------------------------------------------------------------------------------------ */

/* FloatVector.reduceLanes(ADD) over L f32 lanes.  The JDK leaves the association unspecified ("the order of
 * operations may vary"): assoc 0 = lanes left to right, assoc 1 = pairwise tree (what a horizontal-add lowering does). */
static float reduce_lanes(const float *v, int L, int assoc) {
    if (assoc == 0) {
        float s = v[0];
        for (int j = 1; j < L; j++) s = s + v[j];
        return s;
    }
    float t[16];
    for (int j = 0; j < L; j++) t[j] = v[j];
    for (int w = L / 2; w >= 1; w /= 2)
        for (int j = 0; j < w; j++) t[j] = t[j] + t[j + w];
    return t[0];
}

/* src/hnsw/simd.clj:26-43 (dot), :52-71 (euclidean), :81-115 (cosine): chunks of L = SPECIES_PREFERRED.length()
 * (4 NEON, 8 AVX2, 16 AVX-512); per chunk f32 lane products, reduceLanes(ADD) in f32, the chunk sum widened to f64 and
 * added to an f64 running sum; the tail (len mod L elements) multiplied and added in f64.  Cosine guard: (zero? magnitude). */
double orc_f32_vector(int metric, const float *a, const float *b, int n, int L, int assoc) {
    const int ub = n - n % L;
    double dot = 0.0, na = 0.0, nb = 0.0;
    float p[16], qa[16], qb[16];
    for (int i = 0; i < ub; i += L) {
        for (int j = 0; j < L; j++) {
            if (metric == ORC_L2) {
                const float d = a[i + j] - b[i + j];
                p[j] = d * d;
            } else {
                p[j] = a[i + j] * b[i + j];
                qa[j] = a[i + j] * a[i + j];
                qb[j] = b[i + j] * b[i + j];
            }
        }
        dot = dot + (double)reduce_lanes(p, L, assoc);
        if (metric == ORC_COSINE) {
            na = na + (double)reduce_lanes(qa, L, assoc);
            nb = nb + (double)reduce_lanes(qb, L, assoc);
        }
    }
    for (int j = ub; j < n; j++) {
        const double aj = (double)a[j], bj = (double)b[j];
        if (metric == ORC_L2) {
            const double d = aj - bj;
            dot = dot + d * d;
        } else {
            dot = dot + aj * bj;
            na = na + aj * aj;
            nb = nb + bj * bj;
        }
    }
    if (metric == ORC_L2) return sqrt(dot);
    if (metric == ORC_DOT) return dot;
    const double mag = sqrt(na) * sqrt(nb);
    return mag == 0.0 ? 1.0 : 1.0 - dot / mag;
}

/* src/hnsw/wip/vector.clj:21-45 (euclidean), :47-86 (cosine): the lane-accumulating twin -- L f32 lane accumulators
 * (.add acc (.mul va vb): two roundings, no fma), ONE reduceLanes at the end, the tail in f64 (Clojure arithmetic on
 * float widens), sqrt / divide in f64.  Cosine guard: (and (> n1 0) (> n2 0)).  (dot: the same accumulation.) */
double orc_f32_lane_accumulate(int metric, const float *a, const float *b, int n, int L, int assoc) {
    const int ub = n - n % L;
    float sd[16] = {0}, sa[16] = {0}, sb[16] = {0};
    for (int i = 0; i < ub; i += L)
        for (int j = 0; j < L; j++) {
            if (metric == ORC_L2) {
                const float d = a[i + j] - b[i + j];
                const float m = d * d;
                sd[j] = sd[j] + m;
            } else {
                const float m = a[i + j] * b[i + j], ma = a[i + j] * a[i + j], mb = b[i + j] * b[i + j];
                sd[j] = sd[j] + m;
                sa[j] = sa[j] + ma;
                sb[j] = sb[j] + mb;
            }
        }
    double dot = (double)reduce_lanes(sd, L, assoc), n1 = 0.0, n2 = 0.0;
    if (metric == ORC_COSINE) {
        n1 = (double)reduce_lanes(sa, L, assoc);
        n2 = (double)reduce_lanes(sb, L, assoc);
    }
    for (int j = ub; j < n; j++) {
        const double aj = (double)a[j], bj = (double)b[j];
        if (metric == ORC_L2) {
            const double d = aj - bj;
            dot = dot + d * d;
        } else {
            dot = dot + aj * bj;
            n1 = n1 + aj * aj;
            n2 = n2 + bj * bj;
        }
    }
    if (metric == ORC_L2) return sqrt(dot);
    if (metric == ORC_DOT) return dot;
    return (n1 > 0.0 && n2 > 0.0) ? 1.0 - dot / (sqrt(n1) * sqrt(n2)) : 1.0;
}

/* One distance evaluator used by every driver below. */
typedef struct {
    int metric, mode, dim;
    const float *base;   /* n x dim, row major */
    const float *norms;  /* n, only for DEV/FAST cosine (device-order / fast norms) */
    const float *q;
    float qnorm;
} dist_ctx;

static void ctx_set_query(dist_ctx *c, const float *q) {
    c->q = q;
    c->qnorm = 0.0f;
    if (c->metric == ORC_COSINE) {
        if (c->mode == ORC_MODE_DEV || c->mode == ORC_MODE_MFMA) c->qnorm = orc_norm_dev(q, c->dim);
        else if (c->mode == ORC_MODE_FAST) c->qnorm = sqrtf(fast_dot(q, q, c->dim));
    }
}
static inline double ctx_dist(const dist_ctx *c, int64_t row) {
    const float *v = c->base + row * (int64_t)c->dim;
    if (c->mode == ORC_MODE_F64) return dist_f64(c->metric, c->q, v, c->dim);
    if (c->mode == ORC_MODE_DEV || (c->mode == ORC_MODE_MFMA && c->metric == ORC_L2))
        return (double)orc_dist_dev(c->metric, c->q, v, c->dim, c->qnorm, c->norms ? c->norms[row] : 0.0f);
    if (c->mode == ORC_MODE_MFMA)
        return (double)orc_dist_mfma(c->metric, c->q, v, c->dim, c->qnorm, c->norms ? c->norms[row] : 0.0f);
    /* FAST */
    if (c->metric == ORC_L2) return (double)sqrtf(fast_l2sq(c->q, v, c->dim));
    float dot = fast_dot(c->q, v, c->dim);
    if (c->metric == ORC_DOT) return (double)-dot;
    float vn = c->norms[row];
    if (c->qnorm > 0.0f && vn > 0.0f) return (double)(1.0f - dot / (c->qnorm * vn));
    return 1.0;
}

void orc_norms(const float *base, int64_t n, int dim, int mode, float *out) {
    for (int64_t i = 0; i < n; i++) {
        const float *v = base + i * dim;
        if (mode == ORC_MODE_DEV || mode == ORC_MODE_MFMA) out[i] = orc_norm_dev(v, dim);
        else if (mode == ORC_MODE_FAST) out[i] = sqrtf(fast_dot(v, v, dim));
        else { /* src/hnsw/ann/partition/ivf_flat.clj:171-177 (f64), rounded for storage */
            double s = 0.0;
            for (int j = 0; j < dim; j++) s = s + (double)v[j] * (double)v[j];
            out[i] = (float)sqrt(s);
        }
    }
}

/* ------------------------------------------------------------------------------------------ */
/* 3. java.util.Random (public JDK specification) -- needed by test/data_generator.clj:28-87   */
/*    and by k-means++ (src/hnsw/ann/partition/ivf_flat.clj:36).                               */
/* ------------------------------------------------------------------------------------------ */
typedef struct {
    uint64_t seed;
    int have_next;
    double next_gauss;
} jrandom;

void jr_init(jrandom *r, int64_t seed) {
    r->seed = ((uint64_t)seed ^ 0x5DEECE66DULL) & ((1ULL << 48) - 1);
    r->have_next = 0;
    r->next_gauss = 0.0;
}
static inline int32_t jr_next(jrandom *r, int bits) {
    r->seed = (r->seed * 0x5DEECE66DULL + 0xBULL) & ((1ULL << 48) - 1);
    return (int32_t)(int64_t)(r->seed >> (48 - bits));
}
int32_t jr_next_int(jrandom *r) { return jr_next(r, 32); }
int32_t jr_next_int_bound(jrandom *r, int32_t bound) {
    int32_t rr = jr_next(r, 31);
    int32_t m = bound - 1;
    if ((bound & m) == 0) return (int32_t)(((int64_t)bound * (int64_t)rr) >> 31);
    for (int32_t u = rr; (int32_t)((uint32_t)u - (uint32_t)(rr = u % bound) + (uint32_t)m) < 0; u = jr_next(r, 31)) {
    }
    return rr;
}
double jr_next_double(jrandom *r) {
    int64_t hi = (int64_t)jr_next(r, 26);
    int64_t lo = (int64_t)jr_next(r, 27);
    return (double)((hi << 27) + lo) * 0x1.0p-53;
}

/* fdlibm __ieee754_log (what java.lang.StrictMath.log specifies), so nextGaussian is bit-exact
 * and does not depend on the host libm. */
static double fdlibm_log(double x) {
    static const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10,
                        two54 = 1.80143985094819840000e+16, Lg1 = 6.666666666666735130e-01,
                        Lg2 = 3.999999999940941908e-01, Lg3 = 2.857142874366239149e-01,
                        Lg4 = 2.222219843214978396e-01, Lg5 = 1.818357216161805012e-01,
                        Lg6 = 1.531383769920937332e-01, Lg7 = 1.479819860511658591e-01;
    union { double d; uint64_t u; } w;
    w.d = x;
    int32_t hx = (int32_t)(w.u >> 32);
    uint32_t lx = (uint32_t)w.u;
    int32_t k = 0, i, j;
    if (hx < 0x00100000) {
        if (((hx & 0x7fffffff) | lx) == 0) return -two54 / 0.0;
        if (hx < 0) return (x - x) / 0.0;
        k -= 54;
        x *= two54;
        w.d = x;
        hx = (int32_t)(w.u >> 32);
    }
    if (hx >= 0x7ff00000) return x + x;
    k += (hx >> 20) - 1023;
    hx &= 0x000fffff;
    i = (hx + 0x95f64) & 0x100000;
    w.d = x;
    w.u = (w.u & 0xffffffffULL) | ((uint64_t)(uint32_t)(hx | (i ^ 0x3ff00000)) << 32);
    x = w.d;
    k += (i >> 20);
    double f = x - 1.0, hfsq, s, z, R, ww, t1, t2, dk;
    if ((0x000fffff & (2 + hx)) < 3) {
        if (f == 0.0) {
            if (k == 0) return 0.0;
            dk = (double)k;
            return dk * ln2_hi + dk * ln2_lo;
        }
        R = f * f * (0.5 - 0.33333333333333333 * f);
        if (k == 0) return f - R;
        dk = (double)k;
        return dk * ln2_hi - ((R - dk * ln2_lo) - f);
    }
    s = f / (2.0 + f);
    dk = (double)k;
    z = s * s;
    i = hx - 0x6147a;
    ww = z * z;
    j = 0x6b851 - hx;
    t1 = ww * (Lg2 + ww * (Lg4 + ww * Lg6));
    t2 = z * (Lg1 + ww * (Lg3 + ww * (Lg5 + ww * Lg7)));
    i |= j;
    R = t2 + t1;
    if (i > 0) {
        hfsq = 0.5 * f * f;
        if (k == 0) return f - (hfsq - s * (hfsq + R));
        return dk * ln2_hi - ((hfsq - (s * (hfsq + R) + dk * ln2_lo)) - f);
    }
    if (k == 0) return f - s * (f - R);
    return dk * ln2_hi - ((s * (f - R) - dk * ln2_lo) - f);
}
double orc_fdlibm_log(double x) { return fdlibm_log(x); }

double jr_next_gaussian(jrandom *r) {
    if (r->have_next) {
        r->have_next = 0;
        return r->next_gauss;
    }
    double v1, v2, s;
    do {
        v1 = 2 * jr_next_double(r) - 1;
        v2 = 2 * jr_next_double(r) - 1;
        s = v1 * v1 + v2 * v2;
    } while (s >= 1 || s == 0);
    double mul = sqrt(-2 * fdlibm_log(s) / s);
    r->next_gauss = v2 * mul;
    r->have_next = 1;
    return v1 * mul;
}

/* test/data_generator.clj:50-87 generate-dataset (values as f64; caller rounds to f32).
 * distribution: 0 gaussian (:28-31), 1 uniform (:71), 2 unit (:33-40), 3 clustered (:42-48,:73-79). */
void orc_generate_dataset(int64_t size, int dim, int distribution, int num_clusters, double noise, int64_t seed,
                          double *out) {
    jrandom r;
    jr_init(&r, seed);
    double *centers = NULL;
    if (distribution == 3) {
        centers = (double *)malloc(sizeof(double) * (size_t)num_clusters * dim);
        for (int64_t i = 0; i < (int64_t)num_clusters * dim; i++) centers[i] = jr_next_gaussian(&r);
    }
    for (int64_t v = 0; v < size; v++) {
        double *o = out + v * dim;
        if (distribution == 0) {
            for (int j = 0; j < dim; j++) o[j] = jr_next_gaussian(&r);
        } else if (distribution == 1) {
            for (int j = 0; j < dim; j++) o[j] = 2 * jr_next_double(&r) - 1;
        } else if (distribution == 2) {
            for (int j = 0; j < dim; j++) o[j] = jr_next_gaussian(&r);
            double s = 0.0;
            for (int j = 0; j < dim; j++) s = s + o[j] * o[j];
            double nrm = sqrt(s);
            if (nrm != 0.0)
                for (int j = 0; j < dim; j++) o[j] = o[j] / nrm;
        } else {
            /* argument order: (.nextInt rng num-clusters) is evaluated before the noise draws */
            int c = jr_next_int_bound(&r, num_clusters);
            for (int j = 0; j < dim; j++) o[j] = centers[(int64_t)c * dim + j] + noise * jr_next_gaussian(&r);
        }
    }
    free(centers);
}

/* ------------------------------------------------------------------------------------------ */
/* 4. HNSW on an array-indexed graph                                                           */
/* ------------------------------------------------------------------------------------------ */
typedef struct {
    double d;
    int64_t seq; /* admission order: the deterministic tie-break this build defines (SURVEY B) */
    int32_t id;
} cand_t;

static inline int cand_less(const cand_t *a, const cand_t *b) {
    if (a->d < b->d) return 1;
    if (a->d > b->d) return 0;
    return a->seq < b->seq;
}
typedef struct {
    cand_t *a;
    int n, cap, is_max;
} heap_t;
static void heap_init(heap_t *h, int cap, int is_max) {
    h->cap = cap < 16 ? 16 : cap;
    h->a = (cand_t *)malloc(sizeof(cand_t) * (size_t)h->cap);
    h->n = 0;
    h->is_max = is_max;
}
static inline int heap_before(const heap_t *h, const cand_t *x, const cand_t *y) {
    return h->is_max ? cand_less(y, x) : cand_less(x, y);
}
static void heap_push(heap_t *h, cand_t c) {
    if (h->n == h->cap) {
        h->cap *= 2;
        h->a = (cand_t *)realloc(h->a, sizeof(cand_t) * (size_t)h->cap);
    }
    int i = h->n++;
    while (i > 0) {
        int p = (i - 1) / 2;
        if (!heap_before(h, &c, &h->a[p])) break;
        h->a[i] = h->a[p];
        i = p;
    }
    h->a[i] = c;
}
static cand_t heap_pop(heap_t *h) {
    cand_t top = h->a[0];
    cand_t last = h->a[--h->n];
    int i = 0;
    for (;;) {
        int l = 2 * i + 1, r = l + 1, m = l;
        if (l >= h->n) break;
        if (r < h->n && heap_before(h, &h->a[r], &h->a[l])) m = r;
        if (!heap_before(h, &h->a[m], &last)) break;
        h->a[i] = h->a[m];
        i = m;
    }
    if (h->n > 0) h->a[i] = last;
    return top;
}

typedef struct {
    int64_t n;
    int M, M0, max_level, entry;
    const int32_t *levels;  /* n */
    const int32_t *l0_adj;  /* n*M0, -1 padded, adjacency-array order */
    const int64_t *up_off;  /* n+1 : start (in level-blocks) of node's upper levels 1..level */
    const int32_t *up_adj;  /* up_off[n]*M, -1 padded */
} graph_view;

static inline const int32_t *gv_adj(const graph_view *g, int32_t node, int level, int *deg) {
    if (level == 0) {
        *deg = g->M0;
        return g->l0_adj + (int64_t)node * g->M0;
    }
    *deg = g->M;
    return g->up_adj + (g->up_off[node] + (level - 1)) * (int64_t)g->M;
}

typedef struct {
    uint64_t *bits;
    int32_t *touched;
    int nt;
} visited_t;

/* src/hnsw/ultra_fast.clj:151-212 search-layer-ultra.  Admission rules :175-178 (expand iff
 * |nearest| < ef or c.dist <= worst) and :195-204 (admit iff |nearest| < ef or d < worst).
 * Deviations defined by this build: neighbours are visited in adjacency-array
 * order instead of HashSet<String> hash order; PQ ties break on admission sequence.
 * eps/ep_d: entry points with their distances (the reference recomputes them :162-167; same
 * values).  Output: `nearest` ascending by (dist, seq).  Returns count. */
static int search_layer(const graph_view *g, const dist_ctx *c, const int32_t *eps, const double *ep_d, int neps,
                        int ef, int level, visited_t *vis, cand_t *out, int64_t *n_eval, int64_t *n_hop) {
    heap_t cand, near;
    heap_init(&cand, ef * 2, 0);
    heap_init(&near, ef + 2, 1);
    int64_t seq = 0;
    vis->nt = 0;
    for (int i = 0; i < neps; i++) {
        int32_t p = eps[i];
        double d = ep_d ? ep_d[i] : ctx_dist(c, p);
        if (!ep_d) (*n_eval)++;
        vis->bits[p >> 6] |= 1ULL << (p & 63);
        vis->touched[vis->nt++] = p;
        cand_t x = {d, seq++, p};
        heap_push(&cand, x);
        heap_push(&near, x);
    }
    while (cand.n > 0) {
        cand_t cur = heap_pop(&cand);
        double worst = near.n == 0 ? 1.7976931348623157e308 : near.a[0].d;
        if (!(near.n < ef || cur.d <= worst)) continue; /* the reference drains the queue (:170-178) */
        if (g->levels[cur.id] < level) continue;        /* :181 level guard */
        (*n_hop)++;
        int deg;
        const int32_t *adj = gv_adj(g, cur.id, level, &deg);
        for (int j = 0; j < deg; j++) {
            int32_t nb = adj[j];
            if (nb < 0) continue;
            if (vis->bits[nb >> 6] & (1ULL << (nb & 63))) continue;
            vis->bits[nb >> 6] |= 1ULL << (nb & 63);
            vis->touched[vis->nt++] = nb;
            double d = ctx_dist(c, nb);
            (*n_eval)++;
            double w = near.n == 0 ? 1.7976931348623157e308 : near.a[0].d;
            if (near.n < ef || d < w) {
                cand_t x = {d, seq++, nb};
                heap_push(&cand, x);
                heap_push(&near, x);
                if (near.n > ef) heap_pop(&near);
            }
        }
    }
    int cnt = near.n;
    for (int i = cnt - 1; i >= 0; i--) out[i] = heap_pop(&near);
    for (int i = 0; i < vis->nt; i++) vis->bits[vis->touched[i] >> 6] = 0;
    free(cand.a);
    free(near.a);
    return cnt;
}

/* src/hnsw/ultra_fast.clj:346-374 search-knn.  ef is explicit here (reference: (max k 50) :355);
 * upper layers use num-closest 1 (:373-374).  Final: ascending by distance, take k (:362-370). */
static int search_knn_one(const graph_view *g, const dist_ctx *c, int k, int ef, visited_t *vis, cand_t *buf,
                          int32_t *out_ids, double *out_d, int64_t *n_eval, int64_t *n_hop) {
    if (g->n == 0 || g->entry < 0) return 0;
    int32_t eps[1] = {g->entry};
    int neps = 1;
    int cnt = 0;
    for (int level = g->max_level; level >= 0; level--) {
        int efl = level > 0 ? 1 : ef;
        cnt = search_layer(g, c, eps, NULL, neps, efl, level, vis, buf, n_eval, n_hop);
        if (level > 0) {
            neps = cnt < 1 ? 0 : 1;
            if (cnt > 0) eps[0] = buf[0].id;
        }
    }
    int m = cnt < k ? cnt : k;
    for (int i = 0; i < m; i++) {
        out_ids[i] = buf[i].id;
        out_d[i] = buf[i].d;
    }
    return m;
}

typedef struct {
    graph_view g;
    dist_ctx c;
    const float *Q;
    int nq, k, ef;
    int32_t *out_ids;
    double *out_d;
    int64_t *out_stats; /* nq x 2: evals, hops (may be NULL) */
    double *out_lat_ms; /* nq per-query latency (may be NULL) */
    volatile int next;
    pthread_mutex_t mu;
} hnsw_job;

#include <time.h>
static double now_ms(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
}

static void *hnsw_worker(void *arg) {
    hnsw_job *job = (hnsw_job *)arg;
    visited_t vis;
    vis.bits = (uint64_t *)calloc((size_t)(job->g.n + 63) / 64 + 1, sizeof(uint64_t));
    vis.touched = (int32_t *)malloc(sizeof(int32_t) * (size_t)(job->g.n + 1));
    cand_t *buf = (cand_t *)malloc(sizeof(cand_t) * (size_t)(job->ef + 4));
    dist_ctx c = job->c;
    for (;;) {
        int qi = __sync_fetch_and_add(&job->next, 1);
        if (qi >= job->nq) break;
        double t0 = now_ms();
        ctx_set_query(&c, job->Q + (int64_t)qi * c.dim);
        int64_t ev = 0, hp = 0;
        int32_t *oi = job->out_ids + (int64_t)qi * job->k;
        double *od = job->out_d + (int64_t)qi * job->k;
        int m = search_knn_one(&job->g, &c, job->k, job->ef, &vis, buf, oi, od, &ev, &hp);
        for (int i = m; i < job->k; i++) {
            oi[i] = -1;
            od[i] = INFINITY;
        }
        if (job->out_stats) {
            job->out_stats[2 * qi] = ev;
            job->out_stats[2 * qi + 1] = hp;
        }
        if (job->out_lat_ms) job->out_lat_ms[qi] = now_ms() - t0;
    }
    free(vis.bits);
    free(vis.touched);
    free(buf);
    return NULL;
}

/* Batch driver: one task per query on a pool of T threads, results in query order --
 * src/hnsw/helper/parallel_search.clj:15-49 parallel-search-futures.  Returns wall ms. */
double orc_hnsw_search_batch(const float *base, int64_t n, int dim, int metric, int mode, const float *norms,
                             const int32_t *levels, const int32_t *l0_adj, int M0, const int64_t *up_off,
                             const int32_t *up_adj, int M, int entry, int max_level, const float *Q, int nq, int k,
                             int ef, int nthreads, int32_t *out_ids, double *out_d, int64_t *out_stats,
                             double *out_lat_ms) {
    hnsw_job job;
    memset(&job, 0, sizeof(job));
    job.g.n = n;
    job.g.M = M;
    job.g.M0 = M0;
    job.g.max_level = max_level;
    job.g.entry = entry;
    job.g.levels = levels;
    job.g.l0_adj = l0_adj;
    job.g.up_off = up_off;
    job.g.up_adj = up_adj;
    job.c.metric = metric;
    job.c.mode = mode;
    job.c.dim = dim;
    job.c.base = base;
    job.c.norms = norms;
    job.Q = Q;
    job.nq = nq;
    job.k = k;
    job.ef = ef < k ? k : ef;
    job.out_ids = out_ids;
    job.out_d = out_d;
    job.out_stats = out_stats;
    job.out_lat_ms = out_lat_ms;
    job.next = 0;
    if (nthreads < 1) nthreads = 1;
    double t0 = now_ms();
    if (nthreads == 1) {
        hnsw_worker(&job);
    } else {
        pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)nthreads);
        for (int i = 0; i < nthreads; i++) pthread_create(&th[i], NULL, hnsw_worker, &job);
        for (int i = 0; i < nthreads; i++) pthread_join(th[i], NULL);
        free(th);
    }
    return now_ms() - t0;
}

/* ---- graph builder (src/hnsw/ultra_fast.clj:122-147, 216-330; src/hnsw/graph.clj:162-295) --
 * Follows the reference's structure: level = floor(ml * -ln U) with ml = 1/ln 2 (:133,143-147);
 * start at lc = min(level, entry-level) with nearest = [entry], no greedy descent (:247-248);
 * per layer search with ef = (if lc>0 1 ef-construction) (:250-251); connect to m candidates,
 * m = 2M at layer 0 else M (:252-255); an over-full neighbour keeps its m closest by plain sort,
 * the dropped reverse edge is not removed (:264-266, 279-299); entry moves up when level > entry
 * level (:271-273).  Deliberate differences ("N"): the level RNG is a seeded
 * java.util.Random (the reference's is unseeded, so its graphs are not reproducible anyway), and
 * `(take m candidates)` takes the m CLOSEST candidates -- the reference takes the first m in
 * PriorityQueue array order, which is unspecified.  ORC_BUILD_FARTHEST takes them from the far end
 * instead (a max-heap's array starts with its worst elements), to study that quirk.
 *
 * `flags` select the pieces of src/hnsw/graph.clj's builder (the namespace behind README's
 * hnsw.hnsw-search, reached through ann/graph/pure_hnsw.clj) on the same skeleton:
 *   ORC_BUILD_HEURISTIC  neighbour selection by get-neighbors-heuristic (graph.clj:162-198): candidates
 *                        ascending by (distance, id) (:165-171); the closest is taken; every further one is
 *                        taken unless it is closer to an already taken one than to the base node (:181-188),
 *                        until M are taken (:177).  Used for the new node's own links and -- as
 *                        prune-connections does (graph.clj:208-232, extend-candidates? false) -- for an over-full
 *                        neighbour list.  The reference's insert links the new node to EVERY candidate its second
 *                        search returns (graph.clj:280-287: with ef-construction entry points and num-closest M,
 *                        search-layer returns all of them) and only prunes the neighbours; an adjacency row of this
 *                        build holds at most m = 2M / M ids (graph.clj:281 computes that m and never uses it), so
 *                        the new node's links are chosen by the same heuristic -- what prune-connections would make
 *                        of its list at the first later insert that touches it.
 *   ORC_BUILD_EXTEND     extend-candidates? = true for the new node's selection: discarded candidates fill the list
 *                        up to M in their order (graph.clj:191-195)
 *   ORC_BUILD_UPPER_EFC  ef-construction on every layer from min(level, entry-level) down (graph.clj:275-278;
 *                        `(if (> lc level) 1 ef-construction)` is always ef-construction there) instead of 1 above
 *                        layer 0 (ultra_fast.clj:250-251)
 *   ORC_BUILD_SYMMETRIC  an edge dropped by a pruning is removed from BOTH lists (graph.clj:226-231); without it the
 *                        reverse edge stays, as in prune-connections-ultra (ultra_fast.clj:279-299)
 *   ORC_BUILD_DESCEND    a greedy ef = 1 walk from the entry's level down to level + 1 first (what the device's
 *                        batched build does, and what graph/search-knn does on the way down: graph.clj:307-311),
 *                        instead of starting at min(level, entry-level) with the entry point itself
 * The graph is an INPUT to the search-parity tests, so none of this affects search parity. */
#define ORC_BUILD_FARTHEST 1
#define ORC_BUILD_HEURISTIC 2
#define ORC_BUILD_EXTEND 4
#define ORC_BUILD_UPPER_EFC 8
#define ORC_BUILD_SYMMETRIC 16
#define ORC_BUILD_DESCEND 32

typedef struct {
    int64_t n;
    int dim, M, M0;
    int32_t *levels;
    int32_t *l0_adj; /* n*(M0+1) during build */
    int32_t *l0_cnt;
    int64_t *up_off;
    int32_t *up_adj; /* blocks of (M+1) during build */
    int32_t *up_cnt;
} build_t;

static int32_t *bld_adj(build_t *b, int32_t node, int level, int32_t **cnt, int *cap) {
    if (level == 0) {
        *cnt = &b->l0_cnt[node];
        *cap = b->M0 + 1;
        return b->l0_adj + (int64_t)node * (b->M0 + 1);
    }
    int64_t blk = b->up_off[node] + (level - 1);
    *cnt = &b->up_cnt[blk];
    *cap = b->M + 1;
    return b->up_adj + blk * (b->M + 1);
}

typedef struct {
    double d;
    int32_t id;
    int32_t ord;
} prune_t;
static int prune_cmp(const void *x, const void *y) {
    const prune_t *a = (const prune_t *)x, *b = (const prune_t *)y;
    if (a->d < b->d) return -1;
    if (a->d > b->d) return 1;
    return a->ord - b->ord; /* stable, like Clojure's sort-by */
}
/* graph.clj:165-171: the sorted set's comparator -- by distance, equal distances by id */
static int heur_cmp(const void *x, const void *y) {
    const prune_t *a = (const prune_t *)x, *b = (const prune_t *)y;
    if (a->d < b->d) return -1;
    if (a->d > b->d) return 1;
    return (a->id > b->id) - (a->id < b->id);
}

/* (distance graph id1 id2) graph.clj:109-115: dist-fn on the two stored vectors, id1's first */
static double pair_dist(const dist_ctx *c0, int32_t i, int32_t j) {
    dist_ctx c = *c0;
    const float *q = c.base + (int64_t)i * c.dim;
    if (c.norms) { /* the rows' stored norms (device order): what the engine's edge distances use for both sides */
        c.q = q;
        c.qnorm = c.norms[i];
    } else {
        ctx_set_query(&c, q);
    }
    return ctx_dist(&c, j);
}

/* get-neighbors-heuristic, graph.clj:162-198.  cand[0..n): (id, distance to the base node); sorted here.  Returns the
 * number selected (<= m), ids in selection order in out[], their distances in out_d[] (may be NULL).  n_eval counts the
 * (distance graph ...) calls. */
static int select_heuristic(const dist_ctx *c, prune_t *cand, int n, int m, int extend, int32_t *out, double *out_d,
                            prune_t *disc, int64_t *n_eval) {
    qsort(cand, (size_t)n, sizeof(prune_t), heur_cmp);
    int nres = 0, ndisc = 0;
    for (int i = 0; i < n && nres < m; i++) { /* (while (and (not (empty? @nearest)) (< (count @result) M)) :177 */
        int closer = 0;
        for (int r = 0; r < nres && !closer; r++) { /* (some ... @result) :183-186 */
            (*n_eval)++;
            if (pair_dist(c, cand[i].id, out[r]) < cand[i].d) closer = 1;
        }
        if (closer) {
            disc[ndisc++] = cand[i];
        } else {
            if (out_d) out_d[nres] = cand[i].d;
            out[nres++] = cand[i].id;
        }
    }
    if (extend) /* :191-195 */
        for (int i = 0; i < ndisc && nres < m; i++) {
            if (out_d) out_d[nres] = disc[i].d;
            out[nres++] = disc[i].id;
        }
    return nres;
}

static void list_remove(int32_t *a, int32_t *cnt, int32_t x) {
    for (int j = 0; j < *cnt; j++)
        if (a[j] == x) {
            for (int t = j; t + 1 < *cnt; t++) a[t] = a[t + 1];
            a[--(*cnt)] = -1;
            return;
        }
}

int orc_hnsw_build_ex(const float *base, int64_t n, int dim, int metric, int mode, int M, int efc, int64_t seed,
                      int flags, int32_t *levels_out, int32_t *l0_adj_out, int64_t *up_off_out, int32_t *up_adj_out,
                      int64_t up_adj_cap, int32_t *entry_out, int32_t *max_level_out, int64_t *counters) {
    const int farthest_quirk = flags & ORC_BUILD_FARTHEST, heur = flags & ORC_BUILD_HEURISTIC;
    const int extend = (flags & ORC_BUILD_EXTEND) != 0, upper_efc = flags & ORC_BUILD_UPPER_EFC;
    const int symmetric = flags & ORC_BUILD_SYMMETRIC, descend = flags & ORC_BUILD_DESCEND;
    build_t b;
    b.n = n;
    b.dim = dim;
    b.M = M;
    b.M0 = 2 * M;
    jrandom rng;
    jr_init(&rng, seed);
    double ml = 1.0 / log(2.0);
    b.levels = levels_out;
    b.up_off = up_off_out;
    int64_t tot = 0;
    for (int64_t i = 0; i < n; i++) {
        double u = jr_next_double(&rng);
        int lv = (int)(ml * (-log(u))); /* (long (* ml (- (Math/log U)))) :143-147 */
        if (u == 0.0) lv = 64;
        if (lv > 30) lv = 30; /* the engine's cap (hnsw.hip: draw_levels); 2^-31 per row */
        b.levels[i] = lv;
        b.up_off[i] = tot;
        tot += lv;
    }
    b.up_off[n] = tot;
    if (tot > up_adj_cap / M) return -1;
    b.l0_adj = (int32_t *)malloc(sizeof(int32_t) * (size_t)n * (b.M0 + 1) + 16);
    b.l0_cnt = (int32_t *)calloc((size_t)n + 1, sizeof(int32_t));
    b.up_adj = (int32_t *)malloc(sizeof(int32_t) * (size_t)(tot + 1) * (M + 1));
    b.up_cnt = (int32_t *)calloc((size_t)tot + 1, sizeof(int32_t));

    dist_ctx c;
    c.metric = metric;
    c.mode = mode;
    c.dim = dim;
    c.base = base;
    c.norms = NULL;
    float *bnorms = NULL;
    if (mode != ORC_MODE_F64 && metric == ORC_COSINE) {
        bnorms = (float *)malloc(sizeof(float) * (size_t)(n + 1));
        orc_norms(base, n, dim, mode, bnorms);
        c.norms = bnorms;
    }
    visited_t vis;
    vis.bits = (uint64_t *)calloc((size_t)(n + 63) / 64 + 1, sizeof(uint64_t));
    vis.touched = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n + 1));
    int efmax = efc > 1 ? efc : 1;
    cand_t *buf = (cand_t *)malloc(sizeof(cand_t) * (size_t)(efmax + 4));
    int32_t *eps = (int32_t *)malloc(sizeof(int32_t) * (size_t)(efmax + 4));
    double *epd = (double *)malloc(sizeof(double) * (size_t)(efmax + 4));
    int prcap = (efmax > b.M0 ? efmax : b.M0) + 4;
    prune_t *pr = (prune_t *)malloc(sizeof(prune_t) * (size_t)prcap);
    prune_t *disc = (prune_t *)malloc(sizeof(prune_t) * (size_t)prcap);
    int32_t *sel = (int32_t *)malloc(sizeof(int32_t) * (size_t)(b.M0 + 4));
    int32_t *keep = (int32_t *)malloc(sizeof(int32_t) * (size_t)(b.M0 + 4));
    int entry = -1, top = -1;
    int64_t ev = 0, hp = 0, hev = 0, nprune = 0;

    /* a graph_view over the build arrays needs fixed strides; search_layer reads through gv_adj,
     * so give it strides M0+1 / M+1 by lying about M0/M (unused slots hold -1). */
    for (int64_t i = 0; i < n * (b.M0 + 1); i++) b.l0_adj[i] = -1;
    for (int64_t i = 0; i < (tot + 1) * (M + 1); i++) b.up_adj[i] = -1;
    graph_view g;
    g.n = n;
    g.M = M + 1;
    g.M0 = b.M0 + 1;
    g.levels = b.levels;
    g.l0_adj = b.l0_adj;
    g.up_off = b.up_off;
    g.up_adj = b.up_adj;

    for (int64_t id = 0; id < n; id++) {
        int level = b.levels[id];
        if (entry < 0) { /* :229-231 */
            entry = (int)id;
            top = level;
            continue;
        }
        ctx_set_query(&c, base + id * dim);
        int neps = 1;
        eps[0] = entry;
        int use_d = 0;
        g.max_level = top;
        g.entry = entry;
        if (descend)
            for (int lc = top; lc > level; lc--) { /* graph.clj:307-311 on the way down: one nearest per layer */
                int cnt = search_layer(&g, &c, eps, use_d ? epd : NULL, neps, 1, lc, &vis, buf, &ev, &hp);
                neps = cnt < 1 ? 0 : 1;
                if (cnt > 0) {
                    eps[0] = buf[0].id;
                    epd[0] = buf[0].d;
                    use_d = 1;
                }
            }
        for (int lc = level < top ? level : top; lc >= 0; lc--) {
            int ef = (lc > 0 && !upper_efc) ? 1 : efc;
            int cnt = search_layer(&g, &c, eps, use_d ? epd : NULL, neps, ef, lc, &vis, buf, &ev, &hp);
            int m = lc == 0 ? b.M0 : M;
            int take;
            if (heur) {
                for (int t = 0; t < cnt; t++) {
                    pr[t].d = buf[t].d;
                    pr[t].id = buf[t].id;
                    pr[t].ord = t;
                }
                take = select_heuristic(&c, pr, cnt, m, extend, sel, NULL, disc, &hev);
            } else {
                take = cnt < m ? cnt : m;
                for (int t = 0; t < take; t++) sel[t] = farthest_quirk ? buf[cnt - 1 - t].id : buf[t].id;
            }
            for (int t = 0; t < take; t++) {
                int32_t nb = sel[t];
                if (b.levels[nb] < lc) continue; /* :258 */
                int32_t *cn, *cq;
                int capn, capq;
                int32_t *an = bld_adj(&b, nb, lc, &cn, &capn);
                int32_t *aq = bld_adj(&b, (int32_t)id, lc, &cq, &capq);
                if (*cq < capq) aq[(*cq)++] = nb; /* new node never exceeds m (take <= m) */
                an[(*cn)++] = (int32_t)id;
                if (*cn > m) {
                    nprune++;
                    dist_ctx c2 = c;
                    if (c2.norms) {
                        c2.q = base + (int64_t)nb * dim;
                        c2.qnorm = c2.norms[nb];
                    } else {
                        ctx_set_query(&c2, base + (int64_t)nb * dim);
                    }
                    const int have = *cn;
                    for (int j = 0; j < have; j++) {
                        pr[j].d = ctx_dist(&c2, an[j]);
                        pr[j].id = an[j];
                        pr[j].ord = j;
                    }
                    if (heur) { /* prune-connections graph.clj:208-232 */
                        int nk = select_heuristic(&c, pr, have, m, 0, keep, NULL, disc, &hev);
                        if (symmetric) /* :226-231: the dropped edge leaves the other list as well */
                            for (int j = 0; j < have; j++) {
                                int kept = 0;
                                for (int r = 0; r < nk; r++) kept |= keep[r] == pr[j].id;
                                if (kept) continue;
                                int32_t *cz;
                                int capz;
                                int32_t *az = bld_adj(&b, pr[j].id, lc, &cz, &capz);
                                list_remove(az, cz, nb);
                            }
                        for (int j = 0; j < nk; j++) an[j] = keep[j];
                        for (int j = nk; j < capn; j++) an[j] = -1;
                        *cn = nk;
                    } else { /* prune-connections-ultra ultra_fast.clj:279-299 */
                        qsort(pr, (size_t)have, sizeof(prune_t), prune_cmp);
                        for (int j = 0; j < m; j++) an[j] = pr[j].id;
                        for (int j = m; j < capn; j++) an[j] = -1;
                        *cn = m;
                    }
                }
            }
            /* (recur (dec lc) candidates) :268 -- all candidates become the next entry points */
            neps = cnt;
            for (int t = 0; t < cnt; t++) {
                eps[t] = buf[t].id;
                epd[t] = buf[t].d;
            }
            use_d = 1;
        }
        if (level > top) { /* :271-273 */
            entry = (int)id;
            top = level;
        }
    }
    /* compact to the exported layout: n*M0 and blocks of M, -1 padded */
    for (int64_t i = 0; i < n; i++)
        for (int j = 0; j < b.M0; j++) l0_adj_out[i * b.M0 + j] = b.l0_adj[i * (b.M0 + 1) + j];
    for (int64_t blk = 0; blk < tot; blk++)
        for (int j = 0; j < M; j++) up_adj_out[blk * M + j] = b.up_adj[blk * (M + 1) + j];
    *entry_out = entry;
    *max_level_out = top < 0 ? 0 : top;
    if (counters) {
        counters[0] = ev;     /* distance evaluations of the layer searches */
        counters[1] = hp;     /* expansions */
        counters[2] = hev;    /* (distance graph ...) calls of the heuristic */
        counters[3] = nprune; /* prunings */
    }
    free(b.l0_adj);
    free(b.l0_cnt);
    free(b.up_adj);
    free(b.up_cnt);
    free(vis.bits);
    free(vis.touched);
    free(buf);
    free(eps);
    free(epd);
    free(pr);
    free(disc);
    free(sel);
    free(keep);
    free(bnorms);
    return 0;
}

int orc_hnsw_build(const float *base, int64_t n, int dim, int metric, int mode, int M, int efc, int64_t seed,
                   int farthest_quirk, int32_t *levels_out, int32_t *l0_adj_out, int64_t *up_off_out,
                   int32_t *up_adj_out, int64_t up_adj_cap, int32_t *entry_out, int32_t *max_level_out) {
    return orc_hnsw_build_ex(base, n, dim, metric, mode, M, efc, seed, farthest_quirk ? ORC_BUILD_FARTHEST : 0, levels_out,
                             l0_adj_out, up_off_out, up_adj_out, up_adj_cap, entry_out, max_level_out, NULL);
}

/* ------------------------------------------------------------------------------------------ */
/* 5. Exact k-NN + recall (src/hnsw/bench.clj:72-92)                                            */
/* ------------------------------------------------------------------------------------------ */
typedef struct {
    double d;
    int64_t ord;
    int32_t id;
} srt_t;
static int srt_cmp(const void *x, const void *y) {
    const srt_t *a = (const srt_t *)x, *b = (const srt_t *)y;
    if (a->d < b->d) return -1;
    if (a->d > b->d) return 1;
    return (a->ord > b->ord) - (a->ord < b->ord); /* stable */
}

/* ------------------------------------------------------------------------------------------ */
/* 4b. graph/insert LITERALLY (src/hnsw/graph.clj:239-295) -- a CPU-only study, not a parity mode. */
/*                                                                                              */
/* What orc_hnsw_build_ex does NOT reproduce of hnsw.graph/insert: the walk starts at layer           */
/* min(level, entry-level) with the entry point itself (no descent from the top, :275-278), searches with ef-construction   */
/* on every layer it visits, then for EVERY layer 0..level runs a second search-layer with num-closest M over the layer-0    */
/* result as entry points -- which returns all of them and what the expansion admits (:280-282: `nearest` starts with every  */
/* entry point and is only pruned one entry per admission) -- links the new node to EVERY returned node (its own list stays  */
/* unpruned until a later insert prunes it as somebody's neighbour), also on layers above those nodes' own level, and         */
/* prunes only the neighbours (:283-287).  Adjacency sets are unbounded here, so the graph cannot live in the engine's        */
/* fixed rows; this section builds it on the heap and searches it with graph.clj:115-161,297-320 (ef a parameter), to put     */
/* its recall / ef curve beside the bounded variant's.  Set iteration order (Clojure hash sets: unspecified) = insertion       */
/* order here; (apply min-key / max-key ...) ties = the first.                                                                 */
/* ------------------------------------------------------------------------------------------ */
typedef struct {
    int32_t *a;
    int32_t n, cap;
} lset_t; /* a set of node ids in insertion order */
typedef struct {
    int64_t n;
    int maxlev;        /* levels 0..maxlev-1 have storage */
    lset_t *adj;       /* [n][maxlev] */
    int32_t *levels;
    int32_t entry;
} lit_t;
static inline lset_t *lit_set(lit_t *g, int32_t node, int level) { return &g->adj[(int64_t)node * g->maxlev + level]; }
static inline int vis_test_set(visited_t *v, int32_t p) { /* 1 if p was visited already */
    if (v->bits[p >> 6] & (1ULL << (p & 63))) return 1;
    v->bits[p >> 6] |= 1ULL << (p & 63);
    v->touched[v->nt++] = p;
    return 0;
}
static inline void vis_clear(visited_t *v) {
    for (int i = 0; i < v->nt; i++) v->bits[v->touched[i] >> 6] = 0;
    v->nt = 0;
}
static int lset_has(const lset_t *s, int32_t x) {
    for (int i = 0; i < s->n; i++)
        if (s->a[i] == x) return 1;
    return 0;
}
static void lset_add(lset_t *s, int32_t x) {
    if (lset_has(s, x)) return;
    if (s->n == s->cap) {
        s->cap = s->cap ? 2 * s->cap : 8;
        s->a = (int32_t *)realloc(s->a, sizeof(int32_t) * (size_t)s->cap);
    }
    s->a[s->n++] = x;
}
static void lset_del(lset_t *s, int32_t x) {
    for (int i = 0; i < s->n; i++)
        if (s->a[i] == x) {
            for (int t = i; t + 1 < s->n; t++) s->a[t] = s->a[t + 1];
            s->n--;
            return;
        }
}
/* search-layer, graph.clj:115-161: returns (keys @nearest) in out[] (unordered), their distances in out_d[] */
static int lit_search_layer(lit_t *g, const dist_ctx *c, const int32_t *eps, int neps, int num_closest, int level, visited_t *vis,
                            heap_t *cand, heap_t *near, int32_t *out, double *out_d, int64_t *n_eval) {
    cand->n = 0;
    near->n = 0;
    vis->nt = 0;
    int64_t seq = 0;
    for (int i = 0; i < neps; i++) { /* :127-132: every entry point into both maps, whatever num-closest is */
        if (vis_test_set(vis, eps[i])) continue;
        cand_t e = {ctx_dist(c, eps[i]), seq++, eps[i]};
        (*n_eval)++;
        heap_push(cand, e);
        heap_push(near, e);
    }
    while (cand->n > 0) { /* :135 */
        cand_t cur = heap_pop(cand);
        if (!(near->n < num_closest || cur.d <= near->a[0].d)) continue; /* :141-144 (the loop drains) */
        const lset_t *nb = lit_set(g, cur.id, level);
        for (int j = 0; j < nb->n; j++) {
            const int32_t x = nb->a[j];
            if (vis_test_set(vis, x)) continue; /* :147-148 */
            const double d = ctx_dist(c, x);
            (*n_eval)++;
            if (near->n < num_closest || d < near->a[0].d) { /* :151-154 */
                cand_t e = {d, seq++, x};
                heap_push(cand, e);
                heap_push(near, e);
                if (near->n > num_closest) (void)heap_pop(near); /* :157-159: ONE entry leaves per admission */
            }
        }
    }
    const int cnt = near->n;
    for (int i = 0; i < cnt; i++) {
        out[i] = near->a[i].id;
        if (out_d) out_d[i] = near->a[i].d;
    }
    vis_clear(vis);
    return cnt;
}

typedef struct {
    lit_t g;
    dist_ctx c;
    float *norms;
    int M, efc;
} lit_index;

void *orc_lit_build(const float *base, int64_t n, int dim, int metric, int mode, int M, int efc, int64_t seed, int64_t *counters) {
    lit_index *L = (lit_index *)calloc(1, sizeof(lit_index));
    L->M = M;
    L->efc = efc;
    L->g.n = n;
    L->g.levels = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n + 1));
    jrandom rng;
    jr_init(&rng, seed);
    const double ml = 1.0 / log(2.0);
    int top = 0;
    for (int64_t i = 0; i < n; i++) { /* assign-level, graph.clj:93-98 */
        const double u = jr_next_double(&rng);
        int lv = (int)(ml * (-log(u)));
        if (u == 0.0 || lv > 30) lv = 30;
        L->g.levels[i] = lv;
        if (lv > top) top = lv;
    }
    L->g.maxlev = top + 1;
    L->g.adj = (lset_t *)calloc((size_t)n * (size_t)L->g.maxlev + 1, sizeof(lset_t));
    L->g.entry = -1;
    L->c.metric = metric;
    L->c.mode = mode;
    L->c.dim = dim;
    L->c.base = base;
    L->c.norms = NULL;
    if (mode != ORC_MODE_F64 && metric == ORC_COSINE) {
        L->norms = (float *)malloc(sizeof(float) * (size_t)(n + 1));
        orc_norms(base, n, dim, mode, L->norms);
        L->c.norms = L->norms;
    }
    visited_t vis;
    vis.bits = (uint64_t *)calloc((size_t)(n + 63) / 64 + 1, sizeof(uint64_t));
    vis.touched = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n + 1));
    heap_t cand, near;
    heap_init(&cand, 1024, 0);
    heap_init(&near, 1024, 1);
    int32_t *cur = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n + 8));
    int32_t *lnk = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n + 8));
    int prcap = 4096;
    prune_t *pr = (prune_t *)malloc(sizeof(prune_t) * (size_t)prcap);
    prune_t *disc = (prune_t *)malloc(sizeof(prune_t) * (size_t)prcap);
    int32_t *sel = (int32_t *)malloc(sizeof(int32_t) * (size_t)prcap);
    int64_t ev = 0, hev = 0, nprune = 0, maxdeg = 0;
    for (int64_t id = 0; id < n; id++) {
        const int level = L->g.levels[id];
        if (L->g.entry < 0) { /* :251-255 */
            L->g.entry = (int32_t)id;
            continue;
        }
        dist_ctx c = L->c;
        ctx_set_query(&c, base + id * (int64_t)dim);
        const int entry_level = L->g.levels[L->g.entry];
        int ncur = 1;
        cur[0] = L->g.entry;
        const int start = level < entry_level ? level : entry_level;
        for (int lc = start; lc >= 0; lc--) /* :275-278: (if (> lc level) 1 ef-construction) is always ef-construction here */
            ncur = lit_search_layer(&L->g, &c, cur, ncur, efc, lc, &vis, &cand, &near, cur, NULL, &ev);
        for (int lc = 0; lc <= level; lc++) { /* :280-287 */
            const int nl = lit_search_layer(&L->g, &c, cur, ncur, M, lc, &vis, &cand, &near, lnk, NULL, &ev);
            const int maxm = lc == 0 ? 2 * M : M; /* prune-connections, :208-232 (max-M = M as default-params has it) */
            for (int t = 0; t < nl; t++) {
                const int32_t nb = lnk[t];
                if (nb == (int32_t)id) continue;
                lset_add(lit_set(&L->g, (int32_t)id, lc), nb); /* connect-nodes :201-206 */
                lset_add(lit_set(&L->g, nb, lc), (int32_t)id);
                lset_t *s = lit_set(&L->g, nb, lc);
                if (s->n > maxdeg) maxdeg = s->n;
                if (s->n <= maxm) continue;
                if (s->n + 4 > prcap) {
                    prcap = 2 * s->n + 8;
                    pr = (prune_t *)realloc(pr, sizeof(prune_t) * (size_t)prcap);
                    disc = (prune_t *)realloc(disc, sizeof(prune_t) * (size_t)prcap);
                    sel = (int32_t *)realloc(sel, sizeof(int32_t) * (size_t)prcap);
                }
                const int cnt = s->n;
                for (int j = 0; j < cnt; j++) {
                    pr[j].id = s->a[j];
                    pr[j].d = pair_dist(&L->c, nb, s->a[j]);
                    pr[j].ord = j;
                    ev++;
                }
                const int keep = select_heuristic(&L->c, pr, cnt, maxm, 0, sel, NULL, disc, &hev);
                nprune++;
                for (int j = 0; j < cnt; j++) { /* the dropped edges leave BOTH sets (:226-231) */
                    const int32_t x = pr[j].id;
                    int kept = 0;
                    for (int r = 0; r < keep && !kept; r++) kept = sel[r] == x;
                    if (!kept) {
                        lset_del(lit_set(&L->g, nb, lc), x);
                        lset_del(lit_set(&L->g, x, lc), nb);
                    }
                }
            }
        }
        if (level > entry_level) L->g.entry = (int32_t)id; /* :290-292 */
    }
    if (counters) {
        counters[0] = ev;
        counters[1] = hev;
        counters[2] = nprune;
        counters[3] = maxdeg;
        int64_t edges = 0, over = 0;
        for (int64_t i = 0; i < n; i++) {
            edges += lit_set(&L->g, (int32_t)i, 0)->n;
            if (lit_set(&L->g, (int32_t)i, 0)->n > 2 * M) over++;
        }
        counters[4] = edges;
        counters[5] = over; /* nodes whose own layer-0 list was never pruned */
    }
    free(vis.bits);
    free(vis.touched);
    free(cand.a);
    free(near.a);
    free(cur);
    free(lnk);
    free(pr);
    free(disc);
    free(sel);
    return L;
}

/* search-knn, graph.clj:297-320, with ef as a parameter (the reference fixes it at (max k 50)); one query */
static int lit_search_knn(lit_index *L, const float *q, int k, int ef, visited_t *vis, heap_t *cand, heap_t *near, int32_t *cur,
                          double *curd, int32_t *out_ids, double *out_d, int64_t *n_eval) {
    dist_ctx c = L->c;
    ctx_set_query(&c, q);
    int ncur = 1;
    cur[0] = L->g.entry;
    for (int level = L->g.levels[L->g.entry]; level >= 0; level--)
        ncur = lit_search_layer(&L->g, &c, cur, ncur, level > 0 ? 1 : ef, level, vis, cand, near, cur, curd, n_eval);
    srt_t *tmp = (srt_t *)malloc(sizeof(srt_t) * (size_t)(ncur + 1));
    for (int i = 0; i < ncur; i++) {
        tmp[i].d = curd[i];
        tmp[i].id = cur[i];
        tmp[i].ord = i;
    }
    qsort(tmp, (size_t)ncur, sizeof(srt_t), srt_cmp);
    const int m = ncur < k ? ncur : k;
    for (int i = 0; i < k; i++) {
        out_ids[i] = i < m ? tmp[i].id : -1;
        out_d[i] = i < m ? tmp[i].d : INFINITY;
    }
    free(tmp);
    return m;
}
void orc_lit_search(void *handle, const float *Q, int nq, int k, int ef, int32_t *out_ids, double *out_d, int64_t *evals) {
    lit_index *L = (lit_index *)handle;
    const int64_t n = L->g.n;
    visited_t vis;
    vis.bits = (uint64_t *)calloc((size_t)(n + 63) / 64 + 1, sizeof(uint64_t));
    vis.touched = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n + 1));
    heap_t cand, near;
    heap_init(&cand, 1024, 0);
    heap_init(&near, 1024, 1);
    int32_t *cur = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n + 8));
    double *curd = (double *)malloc(sizeof(double) * (size_t)(n + 8));
    int64_t ev = 0;
    for (int i = 0; i < nq; i++)
        lit_search_knn(L, Q + (int64_t)i * L->c.dim, k, ef, &vis, &cand, &near, cur, curd, out_ids + (int64_t)i * k, out_d + (int64_t)i * k, &ev);
    if (evals) *evals = ev;
    free(vis.bits);
    free(vis.touched);
    free(cand.a);
    free(near.a);
    free(cur);
    free(curd);
}
void orc_lit_free(void *handle) {
    lit_index *L = (lit_index *)handle;
    for (int64_t i = 0; i < L->g.n * L->g.maxlev; i++) free(L->g.adj[i].a);
    free(L->g.adj);
    free(L->g.levels);
    free(L->norms);
    free(L);
}

typedef struct {
    dist_ctx c;
    int64_t n;
    const float *Q;
    int nq, k;
    int32_t *out_ids;
    double *out_d;
    volatile int next;
} exact_job;

static void *exact_worker(void *arg) {
    exact_job *job = (exact_job *)arg;
    dist_ctx c = job->c;
    srt_t *top = (srt_t *)malloc(sizeof(srt_t) * (size_t)(job->k + 1));
    for (;;) {
        int qi = __sync_fetch_and_add(&job->next, 1);
        if (qi >= job->nq) break;
        ctx_set_query(&c, job->Q + (int64_t)qi * c.dim);
        int cnt = 0;
        for (int64_t i = 0; i < job->n; i++) {
            double d = ctx_dist(&c, i);
            /* keep the k smallest by (d, index): identical to a stable full sort + take k */
            if (cnt < job->k || d < top[cnt - 1].d) {
                int p = cnt < job->k ? cnt++ : cnt - 1;
                while (p > 0 && top[p - 1].d > d) {
                    top[p] = top[p - 1];
                    p--;
                }
                top[p].d = d;
                top[p].id = (int32_t)i;
                top[p].ord = i;
            }
        }
        for (int i = 0; i < job->k; i++) {
            job->out_ids[(int64_t)qi * job->k + i] = i < cnt ? top[i].id : -1;
            job->out_d[(int64_t)qi * job->k + i] = i < cnt ? top[i].d : INFINITY;
        }
    }
    free(top);
    return NULL;
}

/* src/hnsw/bench.clj:72-84 compute-exact-knn (cosine there; any metric here), over the FULL base
 * (the reference's measure-recall bug of using the test subset is not reproduced, SURVEY B). */
double orc_exact_knn(const float *base, int64_t n, int dim, int metric, int mode, const float *norms,
                     const float *Q, int nq, int k, int nthreads, int32_t *out_ids, double *out_d) {
    exact_job job;
    memset(&job, 0, sizeof(job));
    job.c.metric = metric;
    job.c.mode = mode;
    job.c.dim = dim;
    job.c.base = base;
    job.c.norms = norms;
    job.n = n;
    job.Q = Q;
    job.nq = nq;
    job.k = k;
    job.out_ids = out_ids;
    job.out_d = out_d;
    double t0 = now_ms();
    if (nthreads <= 1) exact_worker(&job);
    else {
        pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)nthreads);
        for (int i = 0; i < nthreads; i++) pthread_create(&th[i], NULL, exact_worker, &job);
        for (int i = 0; i < nthreads; i++) pthread_join(th[i], NULL);
        free(th);
    }
    return now_ms() - t0;
}

/* src/hnsw/bench.clj:86-92 calc-recall: |approx ∩ exact| / |exact| on id sets (ids >= 0). */
double orc_recall(const int32_t *approx, const int32_t *exact, int nq, int k) {
    double tot = 0.0;
    for (int q = 0; q < nq; q++) {
        int ne = 0, hit = 0;
        for (int i = 0; i < k; i++) {
            int32_t e = exact[(int64_t)q * k + i];
            if (e < 0) continue;
            ne++;
            for (int j = 0; j < k; j++)
                if (approx[(int64_t)q * k + j] == e) {
                    hit++;
                    break;
                }
        }
        tot += ne ? (double)hit / ne : 1.0;
    }
    return nq ? tot / nq : 1.0;
}

/* ------------------------------------------------------------------------------------------ */
/* 6. IVF-FLAT (src/hnsw/ann/partition/ivf_flat.clj)                                            */
/* ------------------------------------------------------------------------------------------ */
double orc_dist_f64(int metric, const double *a, const double *b, int n) {
    if (metric == ORC_L2) return orc_euclid(a, b, n);
    if (metric == ORC_DOT) return -orc_dot(a, b, n);
    return orc_cosine_ultra(a, b, n); /* default :distance-fn, ivf_flat.clj:145 */
}
static double dist_fd(int metric, const float *a, const double *b, int n) { /* f32 row vs f64 centroid */
    if (metric == ORC_L2) {
        double s = 0.0;
        for (int i = 0; i < n; i++) {
            double d = (double)a[i] - b[i];
            s = s + d * d;
        }
        return sqrt(s);
    }
    double dot = 0.0, n1 = 0.0, n2 = 0.0;
    for (int i = 0; i < n; i++) {
        double x = (double)a[i];
        dot = dot + x * b[i];
        n1 = n1 + x * x;
        n2 = n2 + b[i] * b[i];
    }
    if (metric == ORC_DOT) return -dot;
    if (n1 > 0.0 && n2 > 0.0) return 1.0 - dot / (sqrt(n1) * sqrt(n2));
    return 1.0;
}

/* ivf_flat.clj:79-90 assign-to-nearest-centroid: strict <, lowest index wins ties. */
void orc_kmeans_assign(const float *base, int64_t n, int dim, int metric, const double *centroids, int nlist,
                       int32_t *assign, double *best_d) {
    for (int64_t i = 0; i < n; i++) {
        double md = 1.7976931348623157e308;
        int bi = 0;
        for (int j = 0; j < nlist; j++) {
            double d = dist_fd(metric, base + i * dim, centroids + (int64_t)j * dim, dim);
            if (d < md) {
                md = d;
                bi = j;
            }
        }
        assign[i] = bi;
        if (best_d) best_d[i] = md;
    }
}

/* Same assignment on float32 centroids (what the engine stores) in any arithmetic mode: used to check
 * hnswgpu_kmeans_assign bit for bit (DEV = scan_kernel order, MFMA = tile_scan_kernel order). */
void orc_kmeans_assign_f32(const float *base, int64_t n, int dim, int metric, int mode, const float *cent,
                           int nlist, int32_t *assign, float *best_d) {
    float *cn = (float *)malloc(sizeof(float) * (size_t)(nlist + 1));
    orc_norms(cent, nlist, dim, mode, cn);
    dist_ctx c;
    c.metric = metric;
    c.mode = mode;
    c.dim = dim;
    c.base = cent;
    c.norms = cn;
    for (int64_t i = 0; i < n; i++) {
        ctx_set_query(&c, base + i * dim);
        double md = 1.7976931348623157e308;
        int bi = 0;
        for (int j = 0; j < nlist; j++) {
            double d = ctx_dist(&c, j);
            if (d < md) {
                md = d;
                bi = j;
            }
        }
        assign[i] = bi;
        if (best_d) best_d[i] = (float)md;
    }
    free(cn);
}

/* ivf_flat.clj:32-60 kmeans-plus-plus-init, restated incrementally (min over chosen centroids is
 * kept per row; identical values to recomputing all of them each round as the reference does
 * :43-49).  Random(42) :36.  Returns chosen row indices.  The pick loop (:54-58) is clamped to
 * n-1 against round-off (the reference would throw). */
void orc_kmeanspp(const float *base, int64_t n, int dim, int metric, int nlist, int64_t seed, int32_t *chosen) {
    jrandom rng;
    jr_init(&rng, seed);
    double *mind = (double *)malloc(sizeof(double) * (size_t)n);
    double *cd = (double *)malloc(sizeof(double) * (size_t)dim);
    for (int64_t i = 0; i < n; i++) mind[i] = 1.7976931348623157e308;
    int32_t cur = jr_next_int_bound(&rng, (int32_t)n);
    chosen[0] = cur;
    for (int c = 1; c < nlist; c++) {
        for (int j = 0; j < dim; j++) cd[j] = (double)base[(int64_t)cur * dim + j];
        double sum = 0.0;
        for (int64_t i = 0; i < n; i++) {
            double d = dist_fd(metric, base + i * dim, cd, dim); /* (distance-fn vec centroid) */
            if (d < mind[i]) mind[i] = d;
            sum = sum + mind[i] * mind[i];
        }
        double r = jr_next_double(&rng) * sum;
        double cum = 0.0;
        int64_t i = 0;
        for (;; i++) {
            double dsq = mind[i] * mind[i];
            if (cum + dsq >= r || i == n - 1) break;
            cum = cum + dsq;
        }
        cur = (int32_t)i;
        chosen[c] = cur;
    }
    free(mind);
    free(cd);
}

/* ivf_flat.clj:92-131 partition-vectors-kmeans: k-means++ init, `iters` Lloyd iterations with no
 * convergence test (:100-117), empty cluster keeps its previous centroid (:112-114), centroid =
 * f64 mean in index order (:66-77), final assignment (:120-124).  Lists keep index order. */
void orc_ivf_build(const float *base, int64_t n, int dim, int metric, int nlist, int iters, int64_t seed,
                   double *centroids, int32_t *assign) {
    int32_t *chosen = (int32_t *)malloc(sizeof(int32_t) * (size_t)nlist);
    orc_kmeanspp(base, n, dim, metric, nlist, seed, chosen);
    for (int c = 0; c < nlist; c++)
        for (int j = 0; j < dim; j++) centroids[(int64_t)c * dim + j] = (double)base[(int64_t)chosen[c] * dim + j];
    free(chosen);
    double *sum = (double *)malloc(sizeof(double) * (size_t)nlist * dim);
    int64_t *cnt = (int64_t *)malloc(sizeof(int64_t) * (size_t)nlist);
    for (int it = 0; it < iters; it++) {
        orc_kmeans_assign(base, n, dim, metric, centroids, nlist, assign, NULL);
        memset(sum, 0, sizeof(double) * (size_t)nlist * dim);
        memset(cnt, 0, sizeof(int64_t) * (size_t)nlist);
        for (int64_t i = 0; i < n; i++) {
            double *s = sum + (int64_t)assign[i] * dim;
            for (int j = 0; j < dim; j++) s[j] = s[j] + (double)base[i * dim + j];
            cnt[assign[i]]++;
        }
        for (int c = 0; c < nlist; c++)
            if (cnt[c] > 0)
                for (int j = 0; j < dim; j++) centroids[(int64_t)c * dim + j] = sum[(int64_t)c * dim + j] / (double)cnt[c];
    }
    orc_kmeans_assign(base, n, dim, metric, centroids, nlist, assign, NULL);
    free(sum);
    free(cnt);
}

/* The same build in the engine's arithmetic (used to check hnswgpu_ivf_build / hnswgpu_kmeanspp bit for bit at
 * sizes where f32-vs-f64 rounding would flip a D^2 sample or an assignment): seeding distances in
 * `seed_mode` (the GEMV scan kernel: DEV), assignment distances in `assign_mode` (MFMA tile kernel for
 * cosine / dot, DEV for L2), centroids = f64 means in index order rounded to f32, D^2 sampling over the f32
 * distances in the reference's sequential f64 order. */
void orc_ivf_build_dev(const float *base, int64_t n, int dim, int metric, int seed_mode, int assign_mode, int nlist,
                       int iters, int64_t seed, int32_t *chosen, float *centroids, int32_t *assign) {
    jrandom rng;
    jr_init(&rng, seed);
    float *mind = (float *)malloc(sizeof(float) * (size_t)n);
    float *bnorm = (float *)malloc(sizeof(float) * (size_t)(n + 1));
    orc_norms(base, n, dim, seed_mode, bnorm);
    for (int64_t i = 0; i < n; i++) mind[i] = 3.402823466e+38f;
    dist_ctx c;
    c.metric = metric;
    c.mode = seed_mode;
    c.dim = dim;
    c.base = base;
    c.norms = bnorm;
    int32_t cur = jr_next_int_bound(&rng, (int32_t)n);
    chosen[0] = cur;
    for (int k = 1; k < nlist; k++) {
        ctx_set_query(&c, base + (int64_t)cur * dim); /* the new centroid is the query of the streaming pass */
        double sum = 0.0;
        for (int64_t i = 0; i < n; i++) {
            float d = (float)ctx_dist(&c, i);
            if (d < mind[i]) mind[i] = d;
            sum = sum + (double)mind[i] * (double)mind[i];
        }
        double r = jr_next_double(&rng) * sum, cum = 0.0;
        int64_t i = 0;
        for (;; i++) {
            double dsq = (double)mind[i] * (double)mind[i];
            if (cum + dsq >= r || i == n - 1) break;
            cum = cum + dsq;
        }
        cur = (int32_t)i;
        chosen[k] = cur;
    }
    for (int k = 0; k < nlist; k++) memcpy(centroids + (int64_t)k * dim, base + (int64_t)chosen[k] * dim, sizeof(float) * (size_t)dim);
    double *acc = (double *)malloc(sizeof(double) * (size_t)nlist * dim);
    int64_t *cnt = (int64_t *)malloc(sizeof(int64_t) * (size_t)nlist);
    for (int it = 0; it <= iters; it++) {
        orc_kmeans_assign_f32(base, n, dim, metric, assign_mode, centroids, nlist, assign, NULL);
        if (it == iters) break;
        memset(acc, 0, sizeof(double) * (size_t)nlist * dim);
        memset(cnt, 0, sizeof(int64_t) * (size_t)nlist);
        for (int64_t i = 0; i < n; i++) {
            double *a = acc + (int64_t)assign[i] * dim;
            for (int j = 0; j < dim; j++) a[j] = a[j] + (double)base[i * dim + j];
            cnt[assign[i]]++;
        }
        for (int k = 0; k < nlist; k++)
            if (cnt[k] > 0)
                for (int j = 0; j < dim; j++) centroids[(int64_t)k * dim + j] = (float)(acc[(int64_t)k * dim + j] / (double)cnt[k]);
    }
    free(mind);
    free(bnorm);
    free(acc);
    free(cnt);
}

/* ivf_flat.clj:236-294 search-ivf-flat with an explicit nprobe (:249-251 path) and centroid
 * routing (:261-269); per list search-partition (:217-234): cosine from dot / (qnorm * vnorm)
 * with precomputed norms (:171-177, 275-278), stable sort, take 2k; merge by stable sort, take k
 * (:291-294).  Zero-norm guard -> 1.0 (reference yields NaN; SURVEY Appendix B "N").
 * Lists: list_off[nlist+1], list_ids in list order (index order inside a list).
 * centroids are float32 here: the values the engine stores (the reference keeps f64 means).
 * mode: F64 (truth), DEV (device-order mimic; needs norms + cnorms from orc_norms(mode DEV)). */
void orc_ivf_search(const float *base, int64_t n, int dim, int metric, int mode, int scan_mode, const float *norms,
                    const float *centroids, const float *cnorms, int nlist, const int64_t *list_off,
                    const int32_t *list_ids, const float *Q, int nq, int k, int nprobe, int32_t *out_ids,
                    double *out_d, int32_t *out_probes) {
    if (nprobe > nlist) nprobe = nlist;
    srt_t *cs = (srt_t *)malloc(sizeof(srt_t) * (size_t)nlist);
    int64_t maxtot = 0;
    {
        /* upper bound on candidates kept: nprobe * 2k */
        maxtot = (int64_t)nprobe * 2 * k + 1;
    }
    srt_t *res = (srt_t *)malloc(sizeof(srt_t) * (size_t)maxtot);
    int64_t maxlen = 0;
    for (int l = 0; l < nlist; l++)
        if (list_off[l + 1] - list_off[l] > maxlen) maxlen = list_off[l + 1] - list_off[l];
    srt_t *part = (srt_t *)malloc(sizeof(srt_t) * (size_t)(maxlen + 1));
    dist_ctx cc, cb;
    cc.metric = metric;
    cc.mode = mode;
    cc.dim = dim;
    cc.base = centroids;
    cc.norms = cnorms;
    cb = cc;
    cb.mode = scan_mode; /* routing and list scan may run on different kernels (GEMV vs MFMA tiles) */
    cb.base = base;
    cb.norms = norms;
    for (int qi = 0; qi < nq; qi++) {
        const float *q = Q + (int64_t)qi * dim;
        ctx_set_query(&cc, q);
        ctx_set_query(&cb, q);
        for (int l = 0; l < nlist; l++) {
            cs[l].d = ctx_dist(&cc, l);
            cs[l].ord = l;
            cs[l].id = l;
        }
        qsort(cs, (size_t)nlist, sizeof(srt_t), srt_cmp);
        int64_t nres = 0, ordc = 0;
        for (int p = 0; p < nprobe; p++) {
            int l = cs[p].id;
            if (out_probes) out_probes[(int64_t)qi * nprobe + p] = l;
            int64_t len = list_off[l + 1] - list_off[l];
            for (int64_t j = 0; j < len; j++) {
                int32_t row = list_ids[list_off[l] + j];
                part[j].d = ctx_dist(&cb, row);
                part[j].id = row;
                part[j].ord = j;
            }
            qsort(part, (size_t)len, sizeof(srt_t), srt_cmp);
            int64_t take = len < 2 * k ? len : 2 * k;
            for (int64_t j = 0; j < take; j++) {
                res[nres] = part[j];
                res[nres].ord = ordc++;
                nres++;
            }
        }
        qsort(res, (size_t)nres, sizeof(srt_t), srt_cmp);
        for (int i = 0; i < k; i++) {
            out_ids[(int64_t)qi * k + i] = i < nres ? res[i].id : -1;
            out_d[(int64_t)qi * k + i] = i < nres ? res[i].d : INFINITY;
        }
    }
    free(cs);
    free(res);
    free(part);
}
