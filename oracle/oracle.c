/*
 * TEST INFRASTRUCTURE ONLY -- CPU restatement (plain C) of the reference hot path.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library.  The product (mira_amd/, libmira_gpu.so) never links, loads or calls it.
 *
 * Restated here (paths relative to /root/reference):
 *   - radix-2 NTT                src/fft.rs:12-27, 51-115, 118-155, 160-174, 178-226
 *   - CommitmentKey::commit      src/commitment.rs:78-87
 *   - best_multiexp              halo2_proofs::arithmetic (THIRD PARTY, source absent:
 *                                Cargo.toml:60-62 pins only the branch joshbeal/dev-mira of
 *                                github.com/joshbeal/halo2, no Cargo.lock).  Restated from
 *                                the published halo2_proofs 0.3 algorithm (arithmetic.rs:
 *                                multiexp_serial / best_multiexp): one contiguous chunk per
 *                                thread, window c = ceil(ln n) (3 if n < 32, 1 if n < 4),
 *                                256/c + 1 segments high->low, buckets of
 *                                {None | Affine | Projective}, summation by parts.
 *   - field / curve arithmetic   halo2curves bn256::{Fr,Fq,G1Affine}, grumpkin::G1Affine
 *                                (THIRD PARTY, source absent, version unknown): 4 x u64
 *                                little-endian limbs in Montgomery form (R = 2^256), affine
 *                                point = x || y, identity = (0, 0).
 *
 * Pinning: see oracle/pyref.py header and tests/test_oracle_pins.py.  NTT: pinned by
 * src/fft.rs:240-257.  BN256 G1 scalar mul: pinned by src/digest.rs:98-113.  MSM: pinned only
 * algebraically (no known-answer vector in the reference).  Grumpkin / ZETA: PARITY UNPINNED.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef uint64_t u64;
typedef unsigned __int128 u128;

typedef struct { u64 l[4]; } fe;

typedef struct {
    fe p;       /* modulus */
    u64 n0;     /* -p^-1 mod 2^64 */
    fe r1;      /* R mod p (Montgomery one) */
    fe r2;      /* R^2 mod p */
} field_t;

typedef struct { fe x, y; } aff;        /* identity = (0,0) */
typedef struct { fe x, y, z; } jac;     /* identity: z == 0 */

typedef struct {
    const field_t *fb;  /* base field (coordinates) */
    const field_t *fs;  /* scalar field */
    fe b;               /* curve constant, Montgomery */
    aff gen;
} curve_t;

static field_t FQ, FR;
static curve_t CURVES[2];
static int g_init = 0;

/* ------------------------------------------------------------------ bigint helpers */
static int fe_is_zero(const fe *a) { return (a->l[0] | a->l[1] | a->l[2] | a->l[3]) == 0; }
static int fe_eq(const fe *a, const fe *b) {
    return a->l[0] == b->l[0] && a->l[1] == b->l[1] && a->l[2] == b->l[2] && a->l[3] == b->l[3];
}
static int fe_geq(const fe *a, const fe *b) {
    for (int i = 3; i >= 0; i--) {
        if (a->l[i] > b->l[i]) return 1;
        if (a->l[i] < b->l[i]) return 0;
    }
    return 1;
}
static u64 raw_add(fe *r, const fe *a, const fe *b) {
    u128 c = 0;
    for (int i = 0; i < 4; i++) { c += (u128)a->l[i] + b->l[i]; r->l[i] = (u64)c; c >>= 64; }
    return (u64)c;
}
static u64 raw_sub(fe *r, const fe *a, const fe *b) {
    u64 br = 0;
    for (int i = 0; i < 4; i++) {
        u128 d = (u128)a->l[i] - b->l[i] - br;
        r->l[i] = (u64)d; br = (u64)(d >> 64) & 1;
    }
    return br;
}

/* ------------------------------------------------------------------ field ops */
static void f_add(fe *r, const fe *a, const fe *b, const field_t *F) {
    u64 c = raw_add(r, a, b);
    if (c || fe_geq(r, &F->p)) raw_sub(r, r, &F->p);
}
static void f_sub(fe *r, const fe *a, const fe *b, const field_t *F) {
    if (raw_sub(r, a, b)) raw_add(r, r, &F->p);
}
static void f_neg(fe *r, const fe *a, const field_t *F) {
    if (fe_is_zero(a)) { *r = *a; return; }
    raw_sub(r, &F->p, a);
}
static void f_dbl(fe *r, const fe *a, const field_t *F) { f_add(r, a, a, F); }

/* Montgomery product a*b*R^-1 mod p (CIOS, 64-bit limbs) */
static void f_mul(fe *r, const fe *a, const fe *b, const field_t *F) {
    u64 t[6] = {0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 4; i++) {
        u128 c = 0;
        for (int j = 0; j < 4; j++) {
            c += (u128)a->l[j] * b->l[i] + t[j];
            t[j] = (u64)c; c >>= 64;
        }
        c += t[4]; t[4] = (u64)c; t[5] = (u64)(c >> 64);
        u64 m = t[0] * F->n0;
        c = (u128)m * F->p.l[0] + t[0]; c >>= 64;
        for (int j = 1; j < 4; j++) {
            c += (u128)m * F->p.l[j] + t[j];
            t[j - 1] = (u64)c; c >>= 64;
        }
        c += t[4]; t[3] = (u64)c; t[4] = t[5] + (u64)(c >> 64);
    }
    fe o = {{t[0], t[1], t[2], t[3]}};
    if (t[4] || fe_geq(&o, &F->p)) raw_sub(&o, &o, &F->p);
    *r = o;
}
static void f_sqr(fe *r, const fe *a, const field_t *F) { f_mul(r, a, a, F); }
static void f_to_mont(fe *r, const fe *a, const field_t *F) { f_mul(r, a, &F->r2, F); }
static void f_from_mont(fe *r, const fe *a, const field_t *F) {
    fe one = {{1, 0, 0, 0}};
    f_mul(r, a, &one, F);
}
/* a^e, e a plain 256-bit integer */
static void f_pow(fe *r, const fe *a, const fe *e, const field_t *F) {
    fe acc = F->r1, base = *a;
    for (int i = 0; i < 256; i++) {
        if ((e->l[i / 64] >> (i % 64)) & 1) f_mul(&acc, &acc, &base, F);
        f_sqr(&base, &base, F);
    }
    *r = acc;
}
static void f_inv(fe *r, const fe *a, const field_t *F) {
    fe e = F->p, two = {{2, 0, 0, 0}};
    raw_sub(&e, &e, &two);
    f_pow(r, a, &e, F);
}

static void field_init(field_t *F, const u64 p[4]) {
    memcpy(F->p.l, p, 32);
    u64 inv = 1;
    for (int i = 0; i < 6; i++) inv *= 2 - p[0] * inv;  /* Newton: p^-1 mod 2^64 */
    F->n0 = (u64)0 - inv;
    fe x = {{1, 0, 0, 0}};
    for (int i = 0; i < 512; i++) {
        u64 c = raw_add(&x, &x, &x);
        if (c || fe_geq(&x, &F->p)) raw_sub(&x, &x, &F->p);
        if (i == 255) F->r1 = x;
    }
    F->r2 = x;
}

/* ------------------------------------------------------------------ curve ops (a = 0) */
static int aff_is_id(const aff *p) { return fe_is_zero(&p->x) && fe_is_zero(&p->y); }
static void jac_set_id(jac *p) { memset(p, 0, sizeof *p); }
static void jac_from_aff(jac *r, const aff *p, const field_t *F) {
    if (aff_is_id(p)) { jac_set_id(r); return; }
    r->x = p->x; r->y = p->y; r->z = F->r1;
}
static void jac_double(jac *r, const jac *p, const field_t *F) {
    if (fe_is_zero(&p->z)) { *r = *p; return; }
    fe a, b, c, d, e, f, t;
    f_sqr(&a, &p->x, F); f_sqr(&b, &p->y, F); f_sqr(&c, &b, F);
    f_add(&t, &p->x, &b, F); f_sqr(&t, &t, F); f_sub(&t, &t, &a, F); f_sub(&t, &t, &c, F);
    f_dbl(&d, &t, F);
    f_dbl(&e, &a, F); f_add(&e, &e, &a, F);
    f_sqr(&f, &e, F);
    fe z3; f_mul(&z3, &p->y, &p->z, F); f_dbl(&z3, &z3, F);
    fe x3; f_dbl(&t, &d, F); f_sub(&x3, &f, &t, F);
    fe y3; f_sub(&t, &d, &x3, F); f_mul(&y3, &e, &t, F);
    f_dbl(&c, &c, F); f_dbl(&c, &c, F); f_dbl(&c, &c, F); f_sub(&y3, &y3, &c, F);
    r->x = x3; r->y = y3; r->z = z3;
}
static void jac_add_mixed(jac *r, const jac *p, const aff *q, const field_t *F) {
    if (aff_is_id(q)) { *r = *p; return; }
    if (fe_is_zero(&p->z)) { jac_from_aff(r, q, F); return; }
    fe z1z1, u2, s2, h, hh, i, j, rr, v, t;
    f_sqr(&z1z1, &p->z, F);
    f_mul(&u2, &q->x, &z1z1, F);
    f_mul(&s2, &q->y, &p->z, F); f_mul(&s2, &s2, &z1z1, F);
    f_sub(&h, &u2, &p->x, F);
    f_sub(&rr, &s2, &p->y, F);
    if (fe_is_zero(&h)) {
        if (fe_is_zero(&rr)) { jac_double(r, p, F); return; }
        jac_set_id(r); return;
    }
    f_dbl(&rr, &rr, F);
    f_sqr(&hh, &h, F);
    f_dbl(&i, &hh, F); f_dbl(&i, &i, F);
    f_mul(&j, &h, &i, F);
    f_mul(&v, &p->x, &i, F);
    fe x3, y3, z3;
    f_sqr(&x3, &rr, F); f_sub(&x3, &x3, &j, F); f_dbl(&t, &v, F); f_sub(&x3, &x3, &t, F);
    f_sub(&t, &v, &x3, F); f_mul(&y3, &rr, &t, F);
    f_mul(&t, &p->y, &j, F); f_dbl(&t, &t, F); f_sub(&y3, &y3, &t, F);
    f_add(&z3, &p->z, &h, F); f_sqr(&z3, &z3, F); f_sub(&z3, &z3, &z1z1, F); f_sub(&z3, &z3, &hh, F);
    r->x = x3; r->y = y3; r->z = z3;
}
static void jac_add(jac *r, const jac *p, const jac *q, const field_t *F) {
    if (fe_is_zero(&q->z)) { *r = *p; return; }
    if (fe_is_zero(&p->z)) { *r = *q; return; }
    fe z1z1, z2z2, u1, u2, s1, s2, h, i, j, rr, v, t;
    f_sqr(&z1z1, &p->z, F); f_sqr(&z2z2, &q->z, F);
    f_mul(&u1, &p->x, &z2z2, F); f_mul(&u2, &q->x, &z1z1, F);
    f_mul(&s1, &p->y, &q->z, F); f_mul(&s1, &s1, &z2z2, F);
    f_mul(&s2, &q->y, &p->z, F); f_mul(&s2, &s2, &z1z1, F);
    f_sub(&h, &u2, &u1, F);
    f_sub(&rr, &s2, &s1, F);
    if (fe_is_zero(&h)) {
        if (fe_is_zero(&rr)) { jac_double(r, p, F); return; }
        jac_set_id(r); return;
    }
    f_dbl(&rr, &rr, F);
    f_dbl(&i, &h, F); f_sqr(&i, &i, F);
    f_mul(&j, &h, &i, F);
    f_mul(&v, &u1, &i, F);
    fe x3, y3, z3;
    f_sqr(&x3, &rr, F); f_sub(&x3, &x3, &j, F); f_dbl(&t, &v, F); f_sub(&x3, &x3, &t, F);
    f_sub(&t, &v, &x3, F); f_mul(&y3, &rr, &t, F);
    f_mul(&t, &s1, &j, F); f_dbl(&t, &t, F); f_sub(&y3, &y3, &t, F);
    f_add(&z3, &p->z, &q->z, F); f_sqr(&z3, &z3, F); f_sub(&z3, &z3, &z1z1, F); f_sub(&z3, &z3, &z2z2, F);
    f_mul(&z3, &z3, &h, F);
    r->x = x3; r->y = y3; r->z = z3;
}
static void jac_to_aff(aff *r, const jac *p, const field_t *F) {
    if (fe_is_zero(&p->z)) { memset(r, 0, sizeof *r); return; }
    fe zi, zi2, zi3;
    f_inv(&zi, &p->z, F); f_sqr(&zi2, &zi, F); f_mul(&zi3, &zi2, &zi, F);
    f_mul(&r->x, &p->x, &zi2, F); f_mul(&r->y, &p->y, &zi3, F);
}
/* k * P, k a plain (non-Montgomery) 256-bit integer */
static void jac_mul_plain(jac *r, const aff *p, const fe *k, const field_t *F) {
    jac acc; jac_set_id(&acc);
    for (int i = 255; i >= 0; i--) {
        jac_double(&acc, &acc, F);
        if ((k->l[i / 64] >> (i % 64)) & 1) jac_add_mixed(&acc, &acc, p, F);
    }
    *r = acc;
}

/* ------------------------------------------------------------------ init */
static void hex_to_mont_small(fe *r, long v, const field_t *F) {
    fe x = {{(u64)(v < 0 ? -v : v), 0, 0, 0}};
    f_to_mont(&x, &x, F);
    if (v < 0) f_neg(&x, &x, F);
    *r = x;
}
void oracle_init(void) {
    if (g_init) return;
    static const u64 Q[4] = {0x3c208c16d87cfd47ULL, 0x97816a916871ca8dULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL};
    static const u64 R[4] = {0x43e1f593f0000001ULL, 0x2833e84879b97091ULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL};
    field_init(&FQ, Q);
    field_init(&FR, R);
    CURVES[0].fb = &FQ; CURVES[0].fs = &FR;
    hex_to_mont_small(&CURVES[0].b, 3, &FQ);
    hex_to_mont_small(&CURVES[0].gen.x, 1, &FQ);
    hex_to_mont_small(&CURVES[0].gen.y, 2, &FQ);
    CURVES[1].fb = &FR; CURVES[1].fs = &FQ;
    hex_to_mont_small(&CURVES[1].b, -17, &FR);
    hex_to_mont_small(&CURVES[1].gen.x, 1, &FR);
    /* sqrt(-16) = 17631683881184975370165255887551781615748388533673675138860
       = 0x2cf135e7506a45d632d270d45f1181294833fc48d823f272c */
    fe gyp = {{0x833fc48d823f272cULL, 0x2d270d45f1181294ULL, 0xcf135e7506a45d63ULL, 0x0000000000000002ULL}};
    f_to_mont(&CURVES[1].gen.y, &gyp, &FR);
    g_init = 1;
}

/* ------------------------------------------------------------------ exported field helpers
 * All element arguments are 4 x u64 little-endian limbs.  field: 0 = Fq (bn256 base),
 * 1 = Fr (bn256 scalar).                                                                      */
static const field_t *fld(int f) { oracle_init(); return f ? &FR : &FQ; }
void oracle_f_mul(int f, const u64 *a, const u64 *b, u64 *r) { f_mul((fe *)r, (const fe *)a, (const fe *)b, fld(f)); }
void oracle_f_add(int f, const u64 *a, const u64 *b, u64 *r) { f_add((fe *)r, (const fe *)a, (const fe *)b, fld(f)); }
void oracle_f_sub(int f, const u64 *a, const u64 *b, u64 *r) { f_sub((fe *)r, (const fe *)a, (const fe *)b, fld(f)); }
void oracle_f_inv(int f, const u64 *a, u64 *r) { f_inv((fe *)r, (const fe *)a, fld(f)); }
void oracle_f_to_mont(int f, const u64 *a, u64 *r, size_t n) {
    const field_t *F = fld(f);
    for (size_t i = 0; i < n; i++) f_to_mont((fe *)(r + 4 * i), (const fe *)(a + 4 * i), F);
}
void oracle_f_from_mont(int f, const u64 *a, u64 *r, size_t n) {
    const field_t *F = fld(f);
    for (size_t i = 0; i < n; i++) f_from_mont((fe *)(r + 4 * i), (const fe *)(a + 4 * i), F);
}
void oracle_generator(int curve, u64 *out8) { oracle_init(); memcpy(out8, &CURVES[curve].gen, 64); }

int oracle_is_on_curve(int curve, const u64 *pt8) {
    oracle_init();
    const curve_t *C = &CURVES[curve]; const field_t *F = C->fb;
    const aff *p = (const aff *)pt8;
    if (aff_is_id(p)) return 1;
    fe l, rr;
    f_sqr(&l, &p->y, F);
    f_sqr(&rr, &p->x, F); f_mul(&rr, &rr, &p->x, F); f_add(&rr, &rr, &C->b, F);
    return fe_eq(&l, &rr);
}

/* out = a + b (affine, Montgomery) */
void oracle_ec_add(int curve, const u64 *a8, const u64 *b8, u64 *out8) {
    oracle_init();
    const field_t *F = CURVES[curve].fb;
    jac j; jac_from_aff(&j, (const aff *)a8, F);
    jac_add_mixed(&j, &j, (const aff *)b8, F);
    jac_to_aff((aff *)out8, &j, F);
}
/* out = k * P, k in Montgomery form of the curve's scalar field (as the reference holds it) */
void oracle_ec_mul(int curve, const u64 *k4_mont, const u64 *p8, u64 *out8) {
    oracle_init();
    const curve_t *C = &CURVES[curve];
    fe k; f_from_mont(&k, (const fe *)k4_mont, C->fs);
    jac j; jac_mul_plain(&j, (const aff *)p8, &k, C->fb);
    jac_to_aff((aff *)out8, &j, C->fb);
}

/* ------------------------------------------------------------------ MSM
 * Ground truth: sum_i k_i * P_i by double-and-add.                                           */
int oracle_msm_naive(int curve, const u64 *scalars_mont, const u64 *bases, size_t n, u64 *out8) {
    oracle_init();
    const curve_t *C = &CURVES[curve]; const field_t *F = C->fb;
    jac acc; jac_set_id(&acc);
#pragma omp parallel
    {
        jac local; jac_set_id(&local);
#pragma omp for schedule(static) nowait
        for (long i = 0; i < (long)n; i++) {
            fe k; f_from_mont(&k, (const fe *)(scalars_mont + 4 * i), C->fs);
            jac t; jac_mul_plain(&t, (const aff *)(bases + 8 * i), &k, F);
            jac_add(&local, &local, &t, F);
        }
#pragma omp critical
        jac_add(&acc, &acc, &local, F);
    }
    jac_to_aff((aff *)out8, &acc, F);
    return 0;
}

/* halo2_proofs 0.3 multiexp_serial, restated (see header). */
typedef struct { int kind; /* 0 none, 1 affine, 2 projective */ aff a; jac j; } bucket_t;

static size_t get_at(size_t segment, size_t c, const unsigned char *bytes) {
    size_t skip_bits = segment * c, skip_bytes = skip_bits / 8;
    if (skip_bytes >= 32) return 0;
    unsigned char v[8] = {0};
    for (size_t i = 0; i < 8 && skip_bytes + i < 32; i++) v[i] = bytes[skip_bytes + i];
    u64 tmp; memcpy(&tmp, v, 8);
    tmp >>= skip_bits - skip_bytes * 8;
    tmp %= ((u64)1 << c);
    return (size_t)tmp;
}
static void multiexp_serial(const curve_t *C, const u64 *scalars_mont, const aff *bases, size_t n, jac *acc) {
    const field_t *F = C->fb;
    fe *repr = (fe *)malloc(n * sizeof(fe));               /* to_repr(): canonical LE bytes */
    for (size_t i = 0; i < n; i++) f_from_mont(&repr[i], (const fe *)(scalars_mont + 4 * i), C->fs);
    size_t c = n < 4 ? 1 : (n < 32 ? 3 : (size_t)ceil(log((double)n)));
    size_t segments = 256 / c + 1;
    size_t nb = ((size_t)1 << c) - 1;
    bucket_t *buckets = (bucket_t *)malloc(nb * sizeof(bucket_t));
    for (size_t seg = segments; seg-- > 0;) {
        for (size_t k = 0; k < c; k++) jac_double(acc, acc, F);
        for (size_t b = 0; b < nb; b++) buckets[b].kind = 0;
        for (size_t i = 0; i < n; i++) {
            size_t d = get_at(seg, c, (const unsigned char *)&repr[i]);
            if (!d) continue;
            bucket_t *bk = &buckets[d - 1];
            if (bk->kind == 0) { bk->kind = 1; bk->a = bases[i]; }
            else if (bk->kind == 1) { jac_from_aff(&bk->j, &bk->a, F); jac_add_mixed(&bk->j, &bk->j, &bases[i], F); bk->kind = 2; }
            else jac_add_mixed(&bk->j, &bk->j, &bases[i], F);
        }
        jac running; jac_set_id(&running);
        for (size_t b = nb; b-- > 0;) {
            if (buckets[b].kind == 1) jac_add_mixed(&running, &running, &buckets[b].a, F);
            else if (buckets[b].kind == 2) jac_add(&running, &running, &buckets[b].j, F);
            jac_add(acc, acc, &running, F);
        }
    }
    free(buckets); free(repr);
}
/* best_multiexp + to_affine.  threads <= 0: all cores (rayon default).  Returns threads used. */
int oracle_msm_pippenger(int curve, const u64 *scalars_mont, const u64 *bases, size_t n, int threads, u64 *out8) {
    oracle_init();
    const curve_t *C = &CURVES[curve]; const field_t *F = C->fb;
#ifdef _OPENMP
    if (threads <= 0) threads = omp_get_max_threads();
#else
    threads = 1;
#endif
    jac total; jac_set_id(&total);
    if (n > (size_t)threads) {
        size_t chunk = n / (size_t)threads;
        size_t nchunks = (n + chunk - 1) / chunk;
        jac *res = (jac *)calloc(nchunks, sizeof(jac));
#pragma omp parallel for num_threads(threads) schedule(dynamic, 1)
        for (long k = 0; k < (long)nchunks; k++) {
            size_t s = (size_t)k * chunk, e = s + chunk > n ? n : s + chunk;
            multiexp_serial(C, scalars_mont + 4 * s, (const aff *)bases + s, e - s, &res[k]);
        }
        for (size_t k = 0; k < nchunks; k++) jac_add(&total, &total, &res[k], F);
        free(res);
    } else {
        multiexp_serial(C, scalars_mont, (const aff *)bases, n, &total);
    }
    jac_to_aff((aff *)out8, &total, F);
    return threads;
}
/* CommitmentKey::commit (src/commitment.rs:78-87).  Returns 0, or 1 = TooLongInput. */
int oracle_commit(int curve, const u64 *ck, size_t ck_len, const u64 *v, size_t v_len, int threads, u64 *out8) {
    if (ck_len >= v_len) { oracle_msm_pippenger(curve, v, ck, v_len, threads, out8); return 0; }
    return 1;
}

/* ------------------------------------------------------------------ witness folding
 * RelaxedPlonkWitness::fold, src/plonk/mod.rs:1097-1134.  field: 0 = Fq, 1 = Fr.             */
void oracle_fold_witness(int f, const u64 *w1, const u64 *w2, const u64 *r4, size_t n, u64 *out) {
    const field_t *F = fld(f);
#pragma omp parallel for
    for (long i = 0; i < (long)n; i++) {                       /* *w1 + *r * *w2, :1107 */
        fe t; f_mul(&t, (const fe *)r4, (const fe *)(w2 + 4 * i), F);
        f_add((fe *)(out + 4 * i), (const fe *)(w1 + 4 * i), &t, F);
    }
}
void oracle_fold_error(int f, u64 *e, const u64 *const *terms, size_t K, const u64 *r4, size_t n) {
    const field_t *F = fld(f);
    fe pw[16], cur = *(const fe *)r4;                          /* r^1, r^2, ...  :1119-1121 */
    for (size_t k = 0; k < K && k < 16; k++) { pw[k] = cur; f_mul(&cur, &cur, (const fe *)r4, F); }
#pragma omp parallel for
    for (long i = 0; i < (long)n; i++) {                       /* fold(*ei, acc + power_of_r * tk[i]) :1126-1130 */
        fe acc = *(fe *)(e + 4 * i);
        for (size_t k = 0; k < K; k++) { fe t; f_mul(&t, &pw[k], (const fe *)(terms[k] + 4 * i), F); f_add(&acc, &acc, &t, F); }
        *(fe *)(e + 4 * i) = acc;
    }
}

/* ------------------------------------------------------------------ cross-term evaluation
 * GraphEvaluator::evaluate (src/polynomial/graph_evaluator.rs:361-390) for every row, rows in
 * parallel as src/nifs/vanilla/mod.rs:104-113 does.  The graph arrives flattened exactly as
 * include/mira_gpu.h describes (the oracle shares no code with the product, only that layout):
 * one intermediate per calculation, as the reference keeps them (:354-359).
 * columns[k]: host pointer; kinds[k] 0 = field elements, 1 = bytes (selector -> ONE / ZERO,
 * src/plonk/eval.rs:57-69).  Returns 0, or -1 on malformed input.                            */
static size_t rot_row(long row, int rot, long nrows) {          /* get_rotation_idx, :51-53 */
    long r = (row + rot) % nrows; if (r < 0) r += nrows; return (size_t)r;
}
int oracle_graph_eval(int f, const uint32_t *code, size_t code_words, uint32_t ncalc, const u64 *constants, uint32_t nconst,
                      const int32_t *rotations, uint32_t nrot, const void *const *columns, const uint32_t *kinds, uint32_t ncols,
                      const u64 *challenges, uint32_t nchal, size_t nrows, u64 *out) {
    const field_t *F = fld(f);
    int bad = 0;
    if (ncalc == 0) { memset(out, 0, nrows * 32); return 0; }  /* Ok(F::ZERO), :386-389 */
#pragma omp parallel
    {
        fe *inter = (fe *)malloc((size_t)ncalc * sizeof(fe));   /* EvaluationData::intermediates */
#pragma omp for
        for (long row = 0; row < (long)nrows; row++) {
            size_t pc = 0;
            for (uint32_t i = 0; i < ncalc && !bad; i++) {
                if (pc >= code_words) { bad = 1; break; }
                uint32_t head = code[pc++], op = head & 0xFF, nparts = head >> 8;
                uint32_t cnt = op <= 2 ? 2 : op == 6 ? 2 + nparts : 1;
                if (op > 7 || pc + cnt > code_words) { bad = 1; break; }
                fe val[2], cur;
                for (uint32_t k = 0; k < cnt; k++) {            /* get_value, :100-134 */
                    uint32_t s = code[pc + k], kind = s >> 29, pl = s & 0x1FFFFFFFu;
                    fe v;
                    if (kind == 0) { if (pl >= nconst) { bad = 1; break; } v = *(const fe *)(constants + 4 * (size_t)pl); }
                    else if (kind == 1) { if (pl >= i) { bad = 1; break; } v = inter[pl]; }
                    else if (kind == 3) { if (pl >= nchal) { bad = 1; break; } v = *(const fe *)(challenges + 4 * (size_t)pl); }
                    else if (kind == 2) {
                        uint32_t col = pl & 0xFFFFF, ri = pl >> 20;
                        if (col >= ncols || ri >= nrot || !columns[col]) { bad = 1; break; }
                        size_t r = rot_row(row, rotations[ri], (long)nrows);
                        if (kinds[col] == 1) v = ((const unsigned char *)columns[col])[r] ? F->r1 : (fe){{0, 0, 0, 0}};
                        else v = ((const fe *)columns[col])[r];
                    } else { bad = 1; break; }
                    if (op == 6 && k >= 2) {                    /* Horner: value = value * factor + part, :148-155 */
                        f_mul(&cur, &cur, &val[1], F); f_add(&cur, &cur, &v, F);
                    } else { val[k] = v; if (op == 6 && k == 0) cur = v; }
                }
                if (bad) break;
                pc += cnt;
                fe r;
                switch (op) {                                   /* :136-158 */
                    case 0: f_add(&r, &val[0], &val[1], F); break;
                    case 1: f_sub(&r, &val[0], &val[1], F); break;
                    case 2: f_mul(&r, &val[0], &val[1], F); break;
                    case 3: f_mul(&r, &val[0], &val[0], F); break;
                    case 4: f_add(&r, &val[0], &val[0], F); break;
                    case 5: { fe z = {{0, 0, 0, 0}}; f_sub(&r, &z, &val[0], F); break; }
                    case 6: r = cur; break;
                    default: r = val[0]; break;
                }
                inter[i] = r;
            }
            if (!bad) memcpy(out + 4 * row, &inter[ncalc - 1], 32);   /* the last calculation's target, :384-385 */
        }
        free(inter);
    }
    return bad ? -1 : 0;
}

/* ------------------------------------------------------------------ ProtoGalaxy tree reduction
 * The tree_reduce of compute_F / compute_G (src/nifs/protogalaxy/poly/mod.rs:131-166, 263-290):
 * adjacent nodes of height h merge as left + right * weights[p][h]; n = 2^levels leaves.
 * leaves: point p's leaves at element p * point_stride (0 = shared).  out: one element per point. */
void oracle_pow_tree(int f, const u64 *leaves, size_t n, size_t point_stride, const u64 *weights, uint32_t levels, uint32_t points, u64 *out) {
    const field_t *F = fld(f);
#pragma omp parallel for
    for (long p = 0; p < (long)points; p++) {
        fe *cur = (fe *)malloc(n * sizeof(fe));
        memcpy(cur, leaves + 4 * (size_t)p * point_stride, n * sizeof(fe));
        size_t cnt = n;
        for (uint32_t h = 0; h < levels; h++, cnt >>= 1)
            for (size_t i = 0; i < cnt / 2; i++) {
                fe t; f_mul(&t, &cur[2 * i + 1], (const fe *)(weights + 4 * ((size_t)p * levels + h)), F);
                f_add(&cur[i], &cur[2 * i], &t, F);
            }
        memcpy(out + 4 * p, &cur[0], 32);
        free(cur);
    }
}

/* ------------------------------------------------------------------ NTT (src/fft.rs) */
static fe fr_pow_u64(const fe *a, u64 e) {
    fe ee = {{e, 0, 0, 0}}, r; f_pow(&r, a, &ee, &FR); return r;
}
/* src/fft.rs:12-23; result in Montgomery form */
void oracle_get_omega_or_inv(uint32_t k, int is_inverse, u64 *out4) {
    oracle_init();
    fe seven = {{7, 0, 0, 0}}; f_to_mont(&seven, &seven, &FR);
    fe e = FR.p, one = {{1, 0, 0, 0}};
    raw_sub(&e, &e, &one);                          /* r - 1 */
    for (int i = 0; i < 28; i++) {                  /* >> 28 */
        for (int j = 0; j < 3; j++) e.l[j] = (e.l[j] >> 1) | (e.l[j + 1] << 63);
        e.l[3] >>= 1;
    }
    fe w; f_pow(&w, &seven, &e, &FR);               /* ROOT_OF_UNITY */
    if (is_inverse) f_inv(&w, &w, &FR);             /* ROOT_OF_UNITY_INV */
    for (uint32_t i = k; i < 28; i++) f_sqr(&w, &w, &FR);
    memcpy(out4, &w, 32);
}
static size_t bitreverse(size_t input, unsigned limit) {
    size_t r = 0;
    for (unsigned i = 0; i < limit; i++) r |= ((input >> i) & 1) << (limit - 1 - i);
    return r;
}
static void butterfly_combine(fe *a, size_t n, size_t twiddle_chunk, const fe *tw) {
    fe *left = a, *right = a + n / 2;
    fe t = right[0];                                        /* twiddle one: src/fft.rs:134-140 */
    right[0] = left[0];
    f_add(&left[0], &left[0], &t, &FR);
    f_sub(&right[0], &right[0], &t, &FR);
    for (size_t i = 1; i < n / 2; i++) {                    /* src/fft.rs:142-152 */
        f_mul(&t, &right[i], &tw[i * twiddle_chunk], &FR);
        right[i] = left[i];
        f_add(&left[i], &left[i], &t, &FR);
        f_sub(&right[i], &right[i], &t, &FR);
    }
}
static void recursive_butterfly(fe *a, size_t n, size_t twiddle_chunk, const fe *tw, int depth) {
    if (n == 2) {                                           /* src/fft.rs:124-128 */
        fe t = a[1]; a[1] = a[0];
        f_add(&a[0], &a[0], &t, &FR); f_sub(&a[1], &a[1], &t, &FR);
        return;
    }
    if (depth > 0) {                                        /* rayon::join, src/fft.rs:130-133 */
#pragma omp task
        recursive_butterfly(a, n / 2, twiddle_chunk * 2, tw, depth - 1);
#pragma omp task
        recursive_butterfly(a + n / 2, n / 2, twiddle_chunk * 2, tw, depth - 1);
#pragma omp taskwait
    } else {
        recursive_butterfly(a, n / 2, twiddle_chunk * 2, tw, 0);
        recursive_butterfly(a + n / 2, n / 2, twiddle_chunk * 2, tw, 0);
    }
    butterfly_combine(a, n, twiddle_chunk, tw);
}
/* best_fft, src/fft.rs:51-115.  a: n x 4 limbs Montgomery, in place.  Returns threads used. */
int oracle_best_fft(u64 *a_, const u64 *omega4, uint32_t log_n, int threads) {
    oracle_init();
    fe *a = (fe *)a_; fe omega; memcpy(&omega, omega4, 32);
#ifdef _OPENMP
    if (threads <= 0) threads = omp_get_max_threads();
#else
    threads = 1;
#endif
    unsigned log_threads = 0; while ((2u << log_threads) <= (unsigned)threads) log_threads++;
    size_t n = (size_t)1 << log_n;
    for (size_t k = 0; k < n; k++) {                        /* src/fft.rs:67-72 */
        size_t rk = bitreverse(k, log_n);
        if (k < rk) { fe t = a[rk]; a[rk] = a[k]; a[k] = t; }
    }
    size_t ntw = n / 2 ? n / 2 : 1;
    fe *tw = (fe *)malloc(ntw * sizeof(fe));                /* src/fft.rs:75-81 */
    fe w = FR.r1;
    for (size_t i = 0; i < n / 2; i++) { tw[i] = w; f_mul(&w, &w, &omega, &FR); }
    if (log_n <= log_threads) {                             /* src/fft.rs:83-111 */
        size_t chunk = 2, twiddle_chunk = n / 2;
        for (uint32_t l = 0; l < log_n; l++) {
            for (size_t base = 0; base < n; base += chunk) butterfly_combine(a + base, chunk, twiddle_chunk, tw);
            chunk *= 2; twiddle_chunk /= 2;
        }
    } else if (n >= 2) {                                    /* src/fft.rs:112-114 */
#pragma omp parallel num_threads(threads)
#pragma omp single
        recursive_butterfly(a, n, 1, tw, (int)log_threads + 2);
    }
    free(tw);
    return threads;
}
int oracle_fft(u64 *a, uint32_t log_n, int threads) {       /* src/fft.rs:160-162 */
    u64 w[4]; oracle_get_omega_or_inv(log_n, 0, w);
    return oracle_best_fft(a, w, log_n, threads);
}
int oracle_ifft(u64 *a_, uint32_t log_n, int threads) {     /* src/fft.rs:165-174 */
    u64 w[4]; oracle_get_omega_or_inv(log_n, 1, w);
    int t = oracle_best_fft(a_, w, log_n, threads);
    fe two = {{2, 0, 0, 0}}; f_to_mont(&two, &two, &FR);
    fe two_inv; f_inv(&two_inv, &two, &FR);
    fe divisor = fr_pow_u64(&two_inv, log_n);               /* src/fft.rs:25-27 */
    fe *a = (fe *)a_; size_t n = (size_t)1 << log_n;
#pragma omp parallel for num_threads(t)
    for (long i = 0; i < (long)n; i++) f_mul(&a[i], &a[i], &divisor, &FR);
    return t;
}
static void distribute_powers_zeta(fe *a, size_t n, int into_coset) {   /* src/fft.rs:205-226 */
    /* Fr::ZETA = 0x30644e72e131a029048b6e193fd84104cc37a73fec2bc5e9b8ca0b2d36636f23 (recalled) */
    fe z = {{0xb8ca0b2d36636f23ULL, 0xcc37a73fec2bc5e9ULL, 0x048b6e193fd84104ULL, 0x30644e72e131a029ULL}};
    f_to_mont(&z, &z, &FR);
    fe zi; f_sqr(&zi, &z, &FR);
    fe pw[2]; if (into_coset) { pw[0] = z; pw[1] = zi; } else { pw[0] = zi; pw[1] = z; }
    for (size_t idx = 0; idx < n; idx++) {
        size_t i = idx % 3;
        if (i) f_mul(&a[idx], &a[idx], &pw[i - 1], &FR);
    }
}
int oracle_coset_fft(u64 *a, uint32_t log_n, int threads) {
    oracle_init();
    distribute_powers_zeta((fe *)a, (size_t)1 << log_n, 1);
    return oracle_fft(a, log_n, threads);
}
int oracle_coset_ifft(u64 *a, uint32_t log_n, int threads) {
    oracle_init();
    int t = oracle_ifft(a, log_n, threads);
    distribute_powers_zeta((fe *)a, (size_t)1 << log_n, 0);
    return t;
}

/* ------------------------------------------------------------------ synthetic inputs
 * Same definition as oracle/pyref.py (synth_*) and the product's mira_synth_* kernels:
 * a splitmix64 stream per index, seeded seed + i * 0xD6E8FEB86659FD93.                        */
static u64 sm_next(u64 *s) {
    *s += 0x9E3779B97F4A7C15ULL;
    u64 z = *s;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
/* kind 0 uniform, 1 witness-like (70 % zero, 20 % < 2^32, 10 % uniform).  Montgomery out. */
void oracle_synth_scalars(int curve, size_t n, u64 seed, int kind, u64 *out) {
    oracle_init();
    const field_t *F = CURVES[curve].fs;
#pragma omp parallel for
    for (long i = 0; i < (long)n; i++) {
        u64 s = seed + (u64)i * 0xD6E8FEB86659FD93ULL;
        fe v; for (int k = 0; k < 4; k++) v.l[k] = sm_next(&s);
        while (fe_geq(&v, &F->p)) raw_sub(&v, &v, &F->p);
        if (kind == 1) {
            u64 sel = sm_next(&s) % 10;
            if (sel < 7) memset(&v, 0, sizeof v);
            else if (sel < 9) { v.l[0] &= 0xFFFFFFFFULL; v.l[1] = v.l[2] = v.l[3] = 0; }
        }
        f_to_mont((fe *)(out + 4 * i), &v, F);
    }
}
/* P_i = k_i * G, k_i = 128-bit odd integer from the stream. */
void oracle_synth_bases(int curve, size_t n, u64 seed, u64 *out) {
    oracle_init();
    const curve_t *C = &CURVES[curve];
#pragma omp parallel for schedule(dynamic, 64)
    for (long i = 0; i < (long)n; i++) {
        u64 s = seed + (u64)i * 0xD6E8FEB86659FD93ULL;
        fe k = {{0, 0, 0, 0}};
        k.l[0] = sm_next(&s) | 1; k.l[1] = sm_next(&s);
        jac j; jac_mul_plain(&j, &C->gen, &k, C->fb);
        jac_to_aff((aff *)(out + 8 * i), &j, C->fb);
    }
}
int oracle_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
/* Size of the default thread pool of every `threads <= 0` entry point (rayon's global pool in the
 * reference; RAYON_NUM_THREADS / available_parallelism there). */
void oracle_set_num_threads(int threads) {
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
#else
    (void)threads;
#endif
}
