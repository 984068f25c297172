/*
 * rt_oracle.c -- scalar CPU restatement of the reference photon random walk.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is the checker for the HIP kernels in
 * radiative-transfer_amd/csrc; the product never links, loads or calls it.
 * It shares only the public struct/enum definitions of include/rt_amd.h so
 * that tests can compare result arrays element by element.
 *
 * Algorithm restated (citations into the reference checkout):
 *   RunPhotons                MC code.py:620-779 (RT_ORDER_MC),
 *                             Numpy_random_walk.py:280-408 (RT_ORDER_GRAY)
 *   TakeStep                  MC code.py:493-508        z += -ln(xi) * mu
 *   CheckForEscape            MC code.py:576-593        z >  tau_atm * Scale
 *   CheckBoundary             MC code.py:522-574        z <= 0
 *   Absorb                    MC code.py:470-491        xi > Albedo
 *   HGScatter / IsoScatter    MC code.py:427-441 / :414-425
 *   RotateIntoLabFrame        MC code.py:595-618
 *   GetZMomentum              MC code.py:360-369
 *   DraineScatter (pdf only)  Numpy_random_walk.py:106-130
 *   AcceptReject (law only)   MC code.py:239-268: P(lambda_i) ~ min(PDF_i, 1)
 * One photon at a time, draws taken in the per-photon order
 * step, absorb, azimuth, cosine, step, ... which is the order each photon sees
 * in the reference's vectorised loop (MC code.py:496, :479, :436-437).
 *
 * Pinned by tests/golden/walk_replay.npz (outputs of the unmodified reference
 * functions run on identical pre-drawn numbers; generator oracle/make_golden.py).
 *
 * Build: gcc -O2 -shared -fPIC -o librt_oracle.so rt_oracle.c -lm
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../include/rt_amd.h"

#define EMIT_BLOCK 0x80000000u
#define TWO_PI_D (2 * 3.141592653589793)

/* ---------------- Philox4x32-10 (Salmon et al., SC'11) ---------------- */
void ora_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3], k0 = key[0], k1 = key[1];
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

static double u52(uint32_t a, uint32_t b) {
    return (double)(((uint64_t)a << 20) | (uint64_t)(b >> 12)) / 4503599627370496.0;
}

/* RT_RNG_PHILOX32X4: one word per uniform, (w + 1/2) / 2^32 */
static double u32mid(uint32_t w) { return ((double)w + 0.5) / 4294967296.0; }
/* RT_RNG_PHILOX40: word r of a block (0 azimuth, 1 cosine, 2 step, 3 absorb) cut as 24 | 40 | 40 | 24 bits, midpoints */
static double u40cut(const uint32_t w[4], uint32_t r) {
    switch (r) {
        case 0: return ((double)(w[0] >> 8) + 0.5) / 16777216.0;
        case 1: return ((double)(((uint64_t)(w[0] & 0xFFu) << 32) | w[1]) + 0.5) / 1099511627776.0;
        case 2: return ((double)(((uint64_t)w[2] << 8) | (w[3] >> 24)) + 0.5) / 1099511627776.0;
        default: return ((double)(w[3] & 0xFFFFFFu) + 0.5) / 16777216.0;
    }
}

static void block_words(uint64_t id, uint32_t block, uint32_t atm, uint64_t seed, uint32_t w[4]) {
    uint32_t ctr[4] = {(uint32_t)id, (uint32_t)(id >> 32), block, atm};
    uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
    ora_philox4x32_10(ctr, key, w);
}

static void block_uniforms(uint64_t id, uint32_t block, uint32_t atm, uint64_t seed, double* ua, double* ub) {
    uint32_t ctr[4] = {(uint32_t)id, (uint32_t)(id >> 32), block, atm};
    uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
    uint32_t w[4];
    ora_philox4x32_10(ctr, key, w);
    *ua = u52(w[0], w[1]);
    *ub = u52(w[2], w[3]);
}

/* ---------------- emission law: Walker/Vose alias table ----------------
 * P(index i) is proportional to w_i = min(max(pdf_i, 0), 1)  (AcceptReject, MC code.py:239-268: a
 * candidate is accepted when xi < PDF[i]).  The table construction order is part of the definition of
 * the photon streams and is restated here independently of the product's host code:
 *   q_i = (w_i * n) / total; two stacks, "small" (q < 1) and "large" (q >= 1), filled in ascending
 *   index order; pop one of each: prob[s] = q_s, alias[s] = l, q_l = (q_l + q_s) - 1, push l back on
 *   the stack its new q belongs to; whatever is left keeps itself with probability 1.
 * A uniform u selects bucket floor(u*n) and, inside it, the bucket's own index if frac(u*n) < prob,
 * else its alias.                                                                                  */
typedef struct { double* prob; int32_t* alias; int n; } alias_t;

static int build_alias(int n, const double* pdf, alias_t* t) {
    double* q = (double*)malloc(sizeof(double) * (size_t)n);
    int32_t* small = (int32_t*)malloc(sizeof(int32_t) * (size_t)n);
    int32_t* large = (int32_t*)malloc(sizeof(int32_t) * (size_t)n);
    t->prob = (double*)malloc(sizeof(double) * (size_t)n);
    t->alias = (int32_t*)malloc(sizeof(int32_t) * (size_t)n);
    t->n = n;
    double total = 0.0;
    for (int i = 0; i < n; ++i) {
        double p = pdf ? pdf[i] : 1.0;
        if (!(p > 0.0)) p = 0.0;
        if (p > 1.0) p = 1.0;
        q[i] = p;
        total += p;
    }
    int ns = 0, nl = 0;
    for (int i = 0; i < n; ++i) {
        q[i] = (q[i] * (double)n) / total;
        if (q[i] < 1.0) small[ns++] = i; else large[nl++] = i;
    }
    while (ns > 0 && nl > 0) {
        int32_t s = small[--ns], l = large[--nl];
        t->prob[s] = q[s];
        t->alias[s] = l;
        q[l] = (q[l] + q[s]) - 1.0;
        if (q[l] < 1.0) small[ns++] = l; else large[nl++] = l;
    }
    while (nl > 0) { int32_t i = large[--nl]; t->prob[i] = 1.0; t->alias[i] = i; }
    while (ns > 0) { int32_t i = small[--ns]; t->prob[i] = 1.0; t->alias[i] = i; }
    free(q); free(small); free(large);
    return total > 0.0;
}

static void free_alias(alias_t* t) { free(t->prob); free(t->alias); }

static int sample_index(const alias_t* t, double u) {
    double x = u * (double)t->n;
    int i = (int)x;
    if (i >= t->n) i = t->n - 1;
    double frac = x - (double)i;
    return frac < t->prob[i] ? i : t->alias[i];
}

int ora_sample_index(int n, const double* pdf, double u) {
    alias_t t;
    build_alias(n, pdf, &t);
    int i = sample_index(&t, u);
    free_alias(&t);
    return i;
}

/* indices for many uniforms at once (tests of the emission law) */
void ora_sample_indices(int n, const double* pdf, int m, const double* u, int32_t* out) {
    alias_t t;
    build_alias(n, pdf, &t);
    for (int k = 0; k < m; ++k) out[k] = sample_index(&t, u[k]);
    free_alias(&t);
}

/* ---------------- one photon ---------------- */
typedef struct {
    /* replay source */
    const double* xi;
    int max_draws;
    /* philox source */
    int use_philox;
    uint64_t id, seed;
    uint32_t atm;
    /* state */
    int n_draws;
    uint32_t n_emit;
    int have_block;
    uint32_t cur_block;
    double ua, ub;
    /* RT_RNG_PHILOX32X4: walk draw j (0 step, 1 absorb, 2 azimuth, 3 cosine, 4 step, ...) is word (j + 2) & 3 of
     * block (j + 2) >> 2 -- words 0, 1 of block 0 belong to the emission */
    int packed; /* 0, RT_RNG_PHILOX32X4 or RT_RNG_PHILOX40 */
    uint32_t words[4];
} stream_t;

static int draw(stream_t* s, double* u) {
    int j = s->n_draws;
    if (s->use_philox && s->packed) {
        uint32_t block = ((uint32_t)j + 2u) >> 2;
        if (!s->have_block || s->cur_block != block) {
            block_words(s->id, block, s->atm, s->seed, s->words);
            s->cur_block = block;
            s->have_block = 1;
        }
        *u = s->packed == RT_RNG_PHILOX40 ? u40cut(s->words, ((uint32_t)j + 2u) & 3u) : u32mid(s->words[((uint32_t)j + 2u) & 3u]);
    } else if (s->use_philox) {
        uint32_t block = (uint32_t)j >> 1;
        if (!s->have_block || s->cur_block != block) {
            block_uniforms(s->id, block, s->atm, s->seed, &s->ua, &s->ub);
            s->cur_block = block;
            s->have_block = 1;
        }
        *u = (j & 1) ? s->ub : s->ua;
    } else {
        if (j >= s->max_draws) return 0;
        *u = s->xi[j];
    }
    s->n_draws = j + 1;
    return 1;
}

typedef struct {
    int n_wl;
    const double *wl, *g, *albedo, *scale;
    alias_t alias;
    int mono_index; /* >= 0: monochromatic */
} tables_t;

typedef struct {
    long double sums[RT_NSUM];
    uint64_t counts[RT_NCNT];
} acc_t;

typedef struct { double lam, g, albedo, scale, mu0; } consts_t;

static void emit_philox(const rt_params* p, const tables_t* t, stream_t* s, consts_t* c) {
    double ua, ub;
    block_uniforms(s->id, EMIT_BLOCK + s->n_emit, s->atm, s->seed, &ua, &ub);
    s->n_emit++;
    int i = t->mono_index >= 0 ? t->mono_index : sample_index(&t->alias, ua);
    c->lam = t->wl[i];
    c->g = t->g[i];
    c->albedo = t->albedo[i];
    c->scale = t->scale[i];
    c->mu0 = p->rand_mu ? ub : 1.0;
}

static double draine_pdf(double g, double alpha, double m) {
    /* Numpy_random_walk.py:124 */
    return 1 / (4 * 3.141592653589793) * (1 - g * g) / (1 + alpha * (1 + 2 * g * g) / 3) * (1 + alpha * m * m) /
           pow(1 + g * g - 2 * g * m, 1.5);
}

static void walk_one(const rt_params* p, double tau, const tables_t* t, stream_t* s, consts_t c, acc_t* a,
                     uint64_t* mu_hist, double* depth_mom, double* depth_en, rt_photon_trace* tr) {
    const int gray = p->loop_order == RT_ORDER_GRAY;
    const int shell = p->geometry == RT_GEOM_SHELL;
    const double h = p->h;
    const int max_ev = p->max_events_per_photon > 0 ? p->max_events_per_photon : (1 << 20);
    double w = gray ? c.lam : 1 / c.lam;
    double zmax = tau * c.scale;
    double z = 0.0, mu = c.mu0;
    double p0 = mu * w * h, e1 = 0.0, b = 0.0;
    double dep = 0.0; /* RT_GEOM_SHELL: radial momentum deposited by this photon */
    int fate = 0, n_scatter = 0, iter = 0;
    a->counts[RT_CNT_PHOTONS]++;
    a->sums[RT_SUM_INITIAL] += p0;
    for (;;) {
        double u;
        if (iter >= max_ev || !draw(s, &u)) { fate = RT_FATE_GUARD; break; }
        iter++;
        a->counts[RT_CNT_FLIGHTS]++;
        if (shell) {
            /* straight flight of optical length `path` in a uniform spherical shell (RT_GEOM_SHELL in rt_amd.h) */
            double path = -log(u);
            double rin = p->shell_inner * zmax, r = rin + z;
            int reemitted = 0;
            if (mu < 0) {
                double sin2 = 1 - mu * mu;
                double d2 = rin * rin - r * r * sin2;
                if (d2 > 0) {
                    double half = sin2 == 0 ? rin : sqrt(d2), d_in = -r * mu - half;
                    if (d_in < 0) d_in = 0;
                    if (path >= d_in) {
                        double mu_c = half / rin;
                        a->counts[RT_CNT_BOUNDARY_HITS]++;
                        if (p->boundary == RT_BOUNDARY_REFLECT) {
                            path -= d_in; r = rin; z = 0; mu = mu_c;
                        } else if (p->boundary == RT_BOUNDARY_ABSORB) {
                            b += -mu_c * w * h;
                            fate = RT_FATE_BOUNDARY;
                            break;
                        } else {
                            b += -mu_c * w * h;
                            a->counts[RT_CNT_REEMITTED]++;
                            if (s->use_philox) emit_philox(p, t, s, &c);
                            w = gray ? c.lam : 1 / c.lam;
                            zmax = tau * c.scale;
                            z = 0.0;
                            mu = c.mu0;
                            a->sums[RT_SUM_REEMITTED] += mu * w * h;
                            reemitted = 1;
                        }
                    }
                }
            }
            if (!reemitted && 1 - mu * mu == 0) { /* radial flight, exact */
                z = z + path * mu;
                if (z < 0) z = 0;
            } else if (!reemitted) {
                double q = path * (path + 2 * r * mu), r2 = sqrt(r * r + q);
                double m2 = (r * mu + path) / r2;
                mu = m2 > 1 ? 1 : (m2 < -1 ? -1 : m2);
                z = z + q / (r2 + r);
                if (z < 0) z = 0;
            }
        } else {
            z += -log(u) * mu;
        }
        if (!shell && !gray && z > zmax) {
            fate = RT_FATE_ESCAPED;
            e1 = mu * w * h;
            break;
        }
        if (!shell && z <= 0) {
            double pb = mu * w * h;
            a->counts[RT_CNT_BOUNDARY_HITS]++;
            if (p->boundary == RT_BOUNDARY_REFLECT) {
                b -= pb;
                z = -z;
                mu = -mu;
            } else if (p->boundary == RT_BOUNDARY_ABSORB) {
                b += pb;
                fate = RT_FATE_BOUNDARY;
                break;
            } else {
                b += pb;
                a->counts[RT_CNT_REEMITTED]++;
                if (s->use_philox) emit_philox(p, t, s, &c);
                w = gray ? c.lam : 1 / c.lam;
                zmax = tau * c.scale;
                z = 0.0;
                mu = c.mu0;
                a->sums[RT_SUM_REEMITTED] += mu * w * h;
            }
        }
        if (z > zmax) {
            if (gray || shell) {
                fate = RT_FATE_ESCAPED;
                e1 = mu * w * h;
            } else {
                fate = RT_FATE_ESCAPED_UNCOUNTED;
                a->sums[RT_SUM_ESCAPED_UNCOUNTED] += mu * w * h;
            }
            break;
        }
        if (!draw(s, &u)) { fate = RT_FATE_GUARD; break; }
        if (p->abs_only ? (u > 0) : (u > c.albedo)) {
            fate = RT_FATE_ABSORBED;
            double pa = mu * w * h;
            a->sums[RT_SUM_DUST_MOMENTUM] += pa;
            a->sums[RT_SUM_ABSORBED_ENERGY] += w * h;
            dep += pa;
            if (p->n_depth_bins > 0) {
                int k = (int)floor(p->n_depth_bins * (z / zmax));
                if (k < 0) k = 0;
                if (k >= p->n_depth_bins) k = p->n_depth_bins - 1;
                depth_mom[k] += pa;
                depth_en[k] += w * h;
            }
            break;
        }
        /* scatter */
        double mu_old = mu;
        int iso = 0, direct = 0;
        if (p->scatter == RT_SCATTER_ISO) iso = gray ? 1 : (iter == 1);
        if (gray && p->scatter == RT_SCATTER_HG && c.g == 0) iso = 1;
        if (p->scatter == RT_SCATTER_DRAINE && c.g == 0 && p->alpha == 0) iso = 1; /* Numpy_random_walk.py:388-391 */
        double u_phi, u_s, mu_prime = 0;
        if (!draw(s, &u_phi)) { fate = RT_FATE_GUARD; break; }
        if ((p->scatter == RT_SCATTER_DRAINE || p->scatter == RT_SCATTER_DRAINE_EXACT) && !iso) {
            /* draws are consumed in aligned pairs: the partner of the azimuth draw is skipped */
            int ok = draw(s, &u_s), tries = 0;
            while (ok) {
                double y, cnd;
                if (tries++ >= 4096) { ok = 0; break; } /* the device's bound on the rejection loop */
                if (!draw(s, &y) || !draw(s, &cnd)) { ok = 0; break; }
                cnd = 2 * cnd - 1;
                if (p->scatter == RT_SCATTER_DRAINE) {
                    if (draine_pdf(c.g, p->alpha, cnd) > y) { mu_prime = cnd; break; }
                } else { /* Henyey-Greenstein candidate, kept with probability (1 + alpha mu^2) / (1 + alpha) */
                    if (c.g != 0) {
                        double g = c.g, q = (1 - g * g) / (1 + g * cnd);
                        cnd = 1 / (2 * g) * (1 + g * g - q * q);
                    }
                    if (1 + p->alpha * cnd * cnd > y * (1 + p->alpha)) { mu_prime = cnd; break; }
                }
            }
            if (!ok) { fate = RT_FATE_GUARD; break; }
        } else {
            if (!draw(s, &u_s)) { fate = RT_FATE_GUARD; break; }
            double sv = 2 * u_s - 1;
            if (iso) { mu_prime = sv; direct = 1; }
            else {
                double g = c.g;
                double q = (1 - g * g) / (1 + g * sv);
                mu_prime = 1 / (2 * g) * (1 + g * g - q * q);
            }
        }
        if (direct) mu = mu_prime;
        else if (mu == 1) mu = mu_prime;
        else if (mu == -1) mu = -mu_prime;
        else {
            double phi = TWO_PI_D * u_phi;
            mu = -sqrt(1 - mu * mu) * sqrt(1 - mu_prime * mu_prime) * cos(phi) + mu * mu_prime;
        }
        n_scatter++;
        a->counts[RT_CNT_SCATTER]++;
        {
            double dp = (mu_old - mu) * w * h;
            a->sums[RT_SUM_DUST_MOMENTUM] += dp;
            dep += dp;
            if (p->n_depth_bins > 0) {
                int k = (int)floor(p->n_depth_bins * (z / zmax));
                if (k < 0) k = 0;
                if (k >= p->n_depth_bins) k = p->n_depth_bins - 1;
                depth_mom[k] += dp;
            }
        }
    }
    switch (fate) {
        case RT_FATE_ESCAPED: a->counts[RT_CNT_ESCAPED]++; a->sums[RT_SUM_ESCAPED] += e1; break;
        case RT_FATE_ESCAPED_UNCOUNTED: a->counts[RT_CNT_ESCAPED_UNCOUNTED]++; break;
        case RT_FATE_ABSORBED: a->counts[RT_CNT_ABSORBED]++; break;
        case RT_FATE_BOUNDARY: a->counts[RT_CNT_BOUNDARY_ABSORBED]++; break;
        default: a->counts[RT_CNT_GUARD]++; break;
    }
    if (mu_hist && ((fate == RT_FATE_ESCAPED && (gray || shell)) || fate == RT_FATE_ESCAPED_UNCOUNTED)) {
        int k = (int)floor(p->n_mu_bins * mu);
        if (k < 0) k = 0;
        if (k > p->n_mu_bins) k = p->n_mu_bins;
        mu_hist[k]++;
    }
    a->sums[RT_SUM_BOUNDARY] += b;
    {
        double ti = shell ? dep : p0 - e1 + b;
        a->sums[RT_SUM_TRANSFER_SQ] += ti * ti;
        a->sums[RT_SUM_TRANSFER_P0] += ti * p0;
        a->sums[RT_SUM_P0_SQ] += p0 * p0;
    }
    if (tr) {
        tr->final_z = z;
        tr->final_mu = mu;
        tr->fate = fate;
        tr->n_scatter = n_scatter;
        tr->n_draws = s->n_draws;
        tr->pad_ = 0;
    }
}

static void store(const rt_params* p, int atm, const acc_t* a, rt_tallies* out) {
    if (out->sums)
        for (int k = 0; k < RT_NSUM; ++k) out->sums[(size_t)atm * RT_NSUM + k] = (double)a->sums[k];
    if (out->counts)
        for (int k = 0; k < RT_NCNT; ++k) out->counts[(size_t)atm * RT_NCNT + k] = a->counts[k];
    (void)p;
}

/* Replay: per-photon constants and pre-drawn uniforms (see rt_replay in rt_amd.h). */
int ora_replay(const rt_params* p, int64_t n, int32_t max_draws, const double* xi, const double* lambda,
               const double* g, const double* albedo, const double* scale, const double* mu0, rt_tallies* out,
               rt_photon_trace* trace) {
    acc_t a;
    memset(&a, 0, sizeof a);
    int nd = p->n_depth_bins > 0 ? p->n_depth_bins : 0;
    double* dm = (double*)calloc((size_t)(nd > 0 ? nd : 1), sizeof(double));
    double* de = (double*)calloc((size_t)(nd > 0 ? nd : 1), sizeof(double));
    uint64_t* hist = (uint64_t*)calloc((size_t)p->n_mu_bins + 1, sizeof(uint64_t));
    for (int64_t i = 0; i < n; ++i) {
        stream_t s;
        memset(&s, 0, sizeof s);
        s.xi = xi + (size_t)i * (size_t)max_draws;
        s.max_draws = max_draws;
        consts_t c = {lambda[i], g[i], albedo[i], scale[i], mu0[i]};
        walk_one(p, p->tau_atm[0], NULL, &s, c, &a, hist, dm, de, trace ? trace + i : NULL);
    }
    store(p, 0, &a, out);
    if (out->mu_hist) memcpy(out->mu_hist, hist, sizeof(uint64_t) * ((size_t)p->n_mu_bins + 1));
    if (out->depth_momentum && nd) memcpy(out->depth_momentum, dm, sizeof(double) * (size_t)nd);
    if (out->depth_energy && nd) memcpy(out->depth_energy, de, sizeof(double) * (size_t)nd);
    free(dm); free(de); free(hist);
    return 0;
}

/* Production layout: Philox streams, table emission, slabs of this rank only
 * (see rt_run in rt_amd.h).  `trace`, if given, must hold n_photons[0] records and is
 * filled for atmosphere 0 only. */
int ora_run(const rt_params* p, int32_t n_tables, int32_t n_wl, const double* wl, const double* g_all,
            const double* albedo_all, const double* scale_all, const double* pdf_all, rt_tallies* out,
            rt_photon_trace* trace) {
    /* g_all, albedo_all, scale_all, pdf_all: [n_tables][n_wl]; atmosphere a walks table p->table_of_atm[a] */
    tables_t* tabs = (tables_t*)calloc((size_t)n_tables, sizeof(tables_t));
    int mono = -1;
    if (p->monochromatic_wl != 0) {
        double want = round(p->monochromatic_wl * 1e4) / 1e4;
        for (int i = 0; i < n_wl; ++i)
            if (wl[i] == want || wl[i] == p->monochromatic_wl) { mono = i; break; }
        if (mono < 0) { free(tabs); return RT_ERR_INVALID; }
    }
    for (int k = 0; k < n_tables; ++k) {
        tables_t* tk = &tabs[k];
        tk->n_wl = n_wl; tk->wl = wl; tk->g = g_all + (size_t)k * n_wl; tk->albedo = albedo_all + (size_t)k * n_wl;
        tk->scale = scale_all + (size_t)k * n_wl; tk->mono_index = mono;
        if (!build_alias(n_wl, pdf_all ? pdf_all + (size_t)k * n_wl : NULL, &tk->alias)) return RT_ERR_INVALID;
    }
    const int n_slabs = p->n_slabs > 0 ? p->n_slabs : 8;
    const int world = p->world > 0 ? p->world : 1;
    const int nd = p->n_depth_bins > 0 ? p->n_depth_bins : 0;
    for (int atm = 0; atm < p->n_atm; ++atm) {
        acc_t a;
        memset(&a, 0, sizeof a);
        const int which = p->table_of_atm ? p->table_of_atm[atm] : 0;
        if (which < 0 || which >= n_tables) return RT_ERR_INVALID;
        const tables_t t = tabs[which];
        const double *g = t.g, *albedo = t.albedo, *scale = t.scale;
        uint64_t N = p->n_photons[atm];
        uint64_t L = (N + (uint64_t)n_slabs - 1) / (uint64_t)n_slabs;
        uint64_t* hist = out->mu_hist ? out->mu_hist + (size_t)atm * ((size_t)p->n_mu_bins + 1) : NULL;
        double* dm = (out->depth_momentum && nd) ? out->depth_momentum + (size_t)atm * nd : NULL;
        double* de = (out->depth_energy && nd) ? out->depth_energy + (size_t)atm * nd : NULL;
        double* dm_l = (double*)calloc((size_t)(nd > 0 ? nd : 1), sizeof(double));
        double* de_l = (double*)calloc((size_t)(nd > 0 ? nd : 1), sizeof(double));
        if (hist) memset(hist, 0, sizeof(uint64_t) * ((size_t)p->n_mu_bins + 1));
        for (int sl = 0; sl < n_slabs; ++sl) {
            acc_t as;
            memset(&as, 0, sizeof as);
            if (sl % world == p->rank) {
                uint64_t lo = (uint64_t)sl * L, hi = lo + L < N ? lo + L : N;
                for (uint64_t i = lo; i < hi; ++i) {
                    stream_t s;
                    memset(&s, 0, sizeof s);
                    s.use_philox = 1;
                    s.id = i + p->photon_id_base;
                    s.seed = p->seed;
                    s.atm = (uint32_t)atm;
                    consts_t c;
                    if (p->rng_mode != RT_RNG_PHILOX52) {
                        /* wavelength from words 0, 1 of walk block 0 (52 bits); the direction draw, if any, is the
                         * second uniform of emission block 0, as in the default mode */
                        uint32_t w[4];
                        double ua, ub;
                        s.packed = p->rng_mode;
                        block_words(s.id, 0u, s.atm, s.seed, w);
                        block_uniforms(s.id, EMIT_BLOCK, s.atm, s.seed, &ua, &ub);
                        s.n_emit = 1;
                        int k = t.mono_index >= 0 ? t.mono_index : sample_index(&t.alias, u52(w[0], w[1]));
                        c.lam = wl[k]; c.g = g[k]; c.albedo = albedo[k]; c.scale = scale[k];
                        c.mu0 = p->rand_mu ? ub : 1.0;
                    } else {
                        emit_philox(p, &t, &s, &c);
                    }
                    walk_one(p, p->tau_atm[atm], &t, &s, c, &as, hist, dm_l, de_l,
                             (trace && atm == 0) ? trace + i : NULL);
                }
            }
            for (int k = 0; k < RT_NSUM; ++k) {
                a.sums[k] += as.sums[k];
                if (out->slab_sums)
                    out->slab_sums[((size_t)atm * n_slabs + sl) * RT_NSUM + k] = (double)as.sums[k];
            }
            for (int k = 0; k < RT_NCNT; ++k) a.counts[k] += as.counts[k];
        }
        store(p, atm, &a, out);
        if (dm) memcpy(dm, dm_l, sizeof(double) * (size_t)nd);
        if (de) memcpy(de, de_l, sizeof(double) * (size_t)nd);
        free(dm_l); free(de_l);
    }
    for (int k = 0; k < n_tables; ++k) free_alias(&tabs[k].alias);
    free(tabs);
    return 0;
}
