/*
 * mfsgd_oracle.c -- CPU restatement of the matrix-factorisation SGD hot path.
 *
 * TEST INFRASTRUCTURE ONLY (see mfsgd_oracle.h).  PARITY UNPINNED: the
 * reference (/root/reference/README.md:1-2) contains no code to follow, so
 * each function cites the SURVEY.md section 8a row it restates instead.
 *
 * Build: gcc -O2 -ffp-contract=off -fPIC -shared -pthread (oracle/Makefile).
 * -ffp-contract=off matters: every rounding below is part of the contract,
 * the compiler must not fuse a*b+c on its own.  fmaf() is an explicit,
 * single-rounding fused multiply-add (C99 7.12.13.1); the hot functions are
 * cloned for FMA3 hardware so that it is one instruction where available.
 */
#include "mfsgd_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

#if defined(__x86_64__) && defined(__GNUC__) && !defined(__clang__)
#define MFO_CLONES __attribute__((target_clones("arch=haswell", "default")))
#else
#define MFO_CLONES
#endif

/* ---------------------------------------------------------------------------
 * java.util.Random, as specified in the JDK API documentation (class Random:
 * "The class uses a 48-bit seed, which is modified using a linear congruential
 * formula", next(int bits), nextInt(), nextFloat(), nextDouble()).
 * ------------------------------------------------------------------------- */
#define JR_MULT 0x5DEECE66DULL
#define JR_ADD 0xBULL
#define JR_MASK ((1ULL << 48) - 1)

void mfo_jrandom_init(mfo_jrandom* g, int64_t seed) {
    g->seed = ((uint64_t)seed ^ JR_MULT) & JR_MASK;
}

int32_t mfo_jrandom_next(mfo_jrandom* g, int bits) {
    g->seed = (g->seed * JR_MULT + JR_ADD) & JR_MASK;
    /* (int)(seed >>> (48 - bits)): truncate to 32 bits, reinterpret signed */
    return (int32_t)(uint32_t)(g->seed >> (48 - bits));
}

int32_t mfo_jrandom_next_int(mfo_jrandom* g) { return mfo_jrandom_next(g, 32); }

float mfo_jrandom_next_float(mfo_jrandom* g) {
    return (float)mfo_jrandom_next(g, 24) / (float)(1 << 24);
}

double mfo_jrandom_next_double(mfo_jrandom* g) {
    int64_t hi = (int64_t)mfo_jrandom_next(g, 26);
    int64_t lo = (int64_t)mfo_jrandom_next(g, 27);
    return (double)((hi << 27) + lo) * 0x1.0p-53;
}

/* SURVEY.md 8a row a4 / "open choices": init U(0,1)/sqrt(k), P first then Q. */
void mfo_init_factors(float* P, float* Q, int32_t U, int32_t I, int32_t k, int64_t seed) {
    mfo_jrandom g;
    mfo_jrandom_init(&g, seed);
    const float scale = (float)(1.0 / sqrt((double)k));
    for (int64_t j = 0; j < (int64_t)U * k; ++j) P[j] = mfo_jrandom_next_float(&g) * scale;
    for (int64_t j = 0; j < (int64_t)I * k; ++j) Q[j] = mfo_jrandom_next_float(&g) * scale;
}

/* ---------------------------------------------------------------------------
 * Canonical arithmetic.  Maximum supported k is 256*MFO_MAX_R (chunks per
 * tree leaf = 1 for k <= 256).
 * ------------------------------------------------------------------------- */
#define MFO_MAX_CHUNKS 64

static inline int chunks_for_k(int32_t k) {
    int need = (k + 3) / 4, L = 1;
    while (L < need) L <<= 1;
    return L;
}

static inline float elem(const float* row, int32_t k, int idx) { return idx < k ? row[idx] : 0.0f; }

/* SURVEY.md 8a row a1 (dot), arithmetic per DESIGN.md section 3. */
static inline __attribute__((always_inline)) float dot_impl(const float* p, const float* q,
                                                            int32_t k) {
    float s[MFO_MAX_CHUNKS];
    const int L = chunks_for_k(k);
    for (int c = 0; c < L; ++c) {
        const int b = 4 * c;
        float t0, t1;
        if (b + 4 <= k) {
            t0 = p[b] * q[b];
            t1 = p[b + 1] * q[b + 1];
            t0 = fmaf(p[b + 2], q[b + 2], t0);
            t1 = fmaf(p[b + 3], q[b + 3], t1);
        } else {
            t0 = elem(p, k, b) * elem(q, k, b);
            t1 = elem(p, k, b + 1) * elem(q, k, b + 1);
            t0 = fmaf(elem(p, k, b + 2), elem(q, k, b + 2), t0);
            t1 = fmaf(elem(p, k, b + 3), elem(q, k, b + 3), t1);
        }
        s[c] = t0 + t1;
    }
    for (int m = 1; m < L; m <<= 1) {
        /* every lane of the butterfly ends with the same value; computing the
         * pairs (c, c^m) once and mirroring is the same arithmetic */
        for (int c = 0; c < L; ++c) {
            if ((c & m) == 0) {
                const float v = s[c] + s[c | m];
                s[c] = v;
                s[c | m] = v;
            }
        }
    }
    return s[0];
}

MFO_CLONES
float mfo_dot(const float* p, const float* q, int32_t k) { return dot_impl(p, q, k); }

/* dot, error, L2-regularised rank-1 update. */
static inline __attribute__((always_inline)) float update_impl(float* p, float* q, int32_t k,
                                                               float r, float lr, float lambda) {
    const float dot = dot_impl(p, q, k);
    const float e = r - dot;
    /* s = lr*(r - dot) evaluated as one fused operation on the rounded product lr*r:
     * a single dependent operation between the dot product and the new rows */
    const float s = fmaf(-lr, dot, lr * r);
    const float c = 1.0f - lr * lambda;
    for (int f = 0; f < k; ++f) {
        const float pf = p[f], qf = q[f];
        const float cp = c * pf;
        const float cq = c * qf;
        p[f] = fmaf(s, qf, cp);
        q[f] = fmaf(s, pf, cq);
    }
    return e;
}

MFO_CLONES
float mfo_sgd_update(float* p, float* q, int32_t k, float r, float lr, float lambda) {
    return update_impl(p, q, k, r, lr, lambda);
}

MFO_CLONES
void mfo_sgd_pass(float* P, float* Q, int32_t k, const int32_t* u, const int32_t* i,
                  const float* r, int64_t n, float lr, float lambda) {
    for (int64_t j = 0; j < n; ++j)
        update_impl(P + (int64_t)u[j] * k, Q + (int64_t)i[j] * k, k, r[j], lr, lambda);
}

MFO_CLONES
void mfo_sgd_pass_ordered(float* P, float* Q, int32_t k, const int32_t* u, const int32_t* i,
                          const float* r, const int64_t* order, int64_t n, float lr,
                          float lambda) {
    for (int64_t j = 0; j < n; ++j) {
        const int64_t x = order[j];
        update_impl(P + (int64_t)u[x] * k, Q + (int64_t)i[x] * k, k, r[x], lr, lambda);
    }
}

/* ---------------------------------------------------------------------------
 * The "textbook" loop: what a Java maintainer's per-rating loop looks like --
 * a plain left-to-right fp32 dot, e = r - dot, then p += lr*(e*q - lambda*p),
 * q += lr*(e*p_old - lambda*q), every operation rounded on its own (SURVEY.md 8a
 * rows a1-a3 read literally, no regrouping, no tree).  It is NOT the contract
 * the GPU is bit-exact against (that is update_impl above); it exists so that
 * bench.py can report a CPU baseline that is not slowed down by the lane-tree
 * dot, next to the RMSE gap between the two arithmetics.
 * ------------------------------------------------------------------------- */
static inline __attribute__((always_inline)) void textbook_update(float* p, float* q, int32_t k,
                                                                  float r, float lr, float lambda) {
    float dot = 0.0f;
    for (int f = 0; f < k; ++f) dot += p[f] * q[f];
    const float e = r - dot;
    for (int f = 0; f < k; ++f) {
        const float pf = p[f], qf = q[f];
        p[f] = pf + lr * (e * qf - lambda * pf);
        q[f] = qf + lr * (e * pf - lambda * qf);
    }
}

MFO_CLONES
void mfo_textbook_pass_ordered(float* P, float* Q, int32_t k, const int32_t* u, const int32_t* i,
                               const float* r, const int64_t* order, int64_t n, float lr,
                               float lambda) {
    for (int64_t j = 0; j < n; ++j) {
        const int64_t x = order[j];
        textbook_update(P + (int64_t)u[x] * k, Q + (int64_t)i[x] * k, k, r[x], lr, lambda);
    }
}

/* ---------------------------------------------------------------------------
 * Multithreaded block-schedule epoch (the CPU baseline; SURVEY.md 8d "CPU
 * baseline timing").  Static cyclic distribution of a round's cells.
 * ------------------------------------------------------------------------- */
typedef void (*mfo_pass_fn)(float*, float*, int32_t, const int32_t*, const int32_t*, const float*,
                            const int64_t*, int64_t, float, float);

typedef struct {
    float *P, *Q;
    int32_t k;
    const int32_t *u, *i;
    const float* r;
    const int64_t *order, *cell_ptr;
    int32_t n_rounds, n_cells, n_threads;
    float lr, lambda;
    mfo_pass_fn pass;
    pthread_barrier_t* bar;
    /* start gate: 0 = wait, 1 = run, 2 = abort (a thread failed to start) */
    pthread_mutex_t mu;
    pthread_cond_t cv;
    int go;
} mt_shared;

typedef struct {
    mt_shared* sh;
    int32_t tid;
} mt_arg;

static void* mt_worker(void* vp) {
    mt_arg* a = (mt_arg*)vp;
    mt_shared* s = a->sh;
    pthread_mutex_lock(&s->mu);
    while (s->go == 0) pthread_cond_wait(&s->cv, &s->mu);
    const int go = s->go;
    pthread_mutex_unlock(&s->mu);
    if (go != 1) return NULL;
    for (int32_t rd = 0; rd < s->n_rounds; ++rd) {
        for (int32_t b = a->tid; b < s->n_cells; b += s->n_threads) {
            const int64_t c = (int64_t)rd * s->n_cells + b;
            const int64_t lo = s->cell_ptr[c], hi = s->cell_ptr[c + 1];
            s->pass(s->P, s->Q, s->k, s->u, s->i, s->r, s->order + lo, hi - lo, s->lr, s->lambda);
        }
        pthread_barrier_wait(s->bar);
    }
    return NULL;
}

static int epoch_mt(mfo_pass_fn pass, float* P, float* Q, int32_t k, const int32_t* u,
                    const int32_t* i, const float* r, const int64_t* order,
                    const int64_t* cell_ptr, int32_t n_rounds, int32_t n_cells, float lr,
                    float lambda, int32_t n_threads) {
    if (n_threads < 1) n_threads = 1;
    pthread_barrier_t bar;
    if (pthread_barrier_init(&bar, NULL, (unsigned)n_threads) != 0) return -1;
    mt_shared sh;
    memset(&sh, 0, sizeof sh);
    sh.P = P; sh.Q = Q; sh.k = k; sh.u = u; sh.i = i; sh.r = r;
    sh.order = order; sh.cell_ptr = cell_ptr;
    sh.n_rounds = n_rounds; sh.n_cells = n_cells; sh.n_threads = n_threads;
    sh.lr = lr; sh.lambda = lambda; sh.pass = pass; sh.bar = &bar; sh.go = 0;
    pthread_mutex_init(&sh.mu, NULL);
    pthread_cond_init(&sh.cv, NULL);
    pthread_t* th = (pthread_t*)malloc(sizeof(pthread_t) * (size_t)n_threads);
    mt_arg* args = (mt_arg*)malloc(sizeof(mt_arg) * (size_t)n_threads);
    int started = 0, rc = 0;
    if (!th || !args) rc = -1;
    for (int32_t t = 1; rc == 0 && t < n_threads; ++t) {
        args[t].sh = &sh;
        args[t].tid = t;
        if (pthread_create(&th[t], NULL, mt_worker, &args[t]) != 0) rc = -1;
        else ++started;
    }
    pthread_mutex_lock(&sh.mu);
    sh.go = rc == 0 ? 1 : 2;
    pthread_cond_broadcast(&sh.cv);
    pthread_mutex_unlock(&sh.mu);
    if (rc == 0) {
        args[0].sh = &sh;
        args[0].tid = 0;
        mt_worker(&args[0]);
    }
    for (int32_t t = 1; t <= started; ++t) pthread_join(th[t], NULL);
    free(th);
    free(args);
    pthread_cond_destroy(&sh.cv);
    pthread_mutex_destroy(&sh.mu);
    pthread_barrier_destroy(&bar);
    return rc;
}

int mfo_sgd_epoch_mt(float* P, float* Q, int32_t k, const int32_t* u, const int32_t* i,
                     const float* r, const int64_t* order, const int64_t* cell_ptr,
                     int32_t n_rounds, int32_t n_cells, float lr, float lambda,
                     int32_t n_threads) {
    return epoch_mt(mfo_sgd_pass_ordered, P, Q, k, u, i, r, order, cell_ptr, n_rounds, n_cells, lr,
                    lambda, n_threads);
}

int mfo_textbook_epoch_mt(float* P, float* Q, int32_t k, const int32_t* u, const int32_t* i,
                          const float* r, const int64_t* order, const int64_t* cell_ptr,
                          int32_t n_rounds, int32_t n_cells, float lr, float lambda,
                          int32_t n_threads) {
    return epoch_mt(mfo_textbook_pass_ordered, P, Q, k, u, i, r, order, cell_ptr, n_rounds,
                    n_cells, lr, lambda, n_threads);
}

/* SURVEY.md 8a row a6. */
MFO_CLONES
double mfo_sse(const float* P, const float* Q, int32_t k, const int32_t* u, const int32_t* i,
               const float* r, int64_t n) {
    double acc = 0.0;
    for (int64_t j = 0; j < n; ++j) {
        const float e = r[j] - dot_impl(P + (int64_t)u[j] * k, Q + (int64_t)i[j] * k, k);
        acc += (double)e * (double)e;
    }
    return acc;
}

double mfo_rmse(const float* P, const float* Q, int32_t k, const int32_t* u, const int32_t* i,
                const float* r, int64_t n) {
    if (n <= 0) return 0.0;
    return sqrt(mfo_sse(P, Q, k, u, i, r, n) / (double)n);
}

MFO_CLONES
void mfo_predict(const float* P, const float* Q, int32_t k, const int32_t* u, const int32_t* i,
                 float* out, int64_t n) {
    for (int64_t j = 0; j < n; ++j)
        out[j] = dot_impl(P + (int64_t)u[j] * k, Q + (int64_t)i[j] * k, k);
}

int mfo_check_block_schedule(const int32_t* u, const int32_t* i, int64_t n, int32_t U, int32_t I,
                             const int64_t* order, const int64_t* cell_ptr, int32_t n_rounds,
                             int32_t n_cells) {
    unsigned char* seen = (unsigned char*)calloc((size_t)(n > 0 ? n : 1), 1);
    /* owner cell of each user / item inside the current round, stamped by round */
    int64_t* uown = (int64_t*)malloc(sizeof(int64_t) * (size_t)(U > 0 ? U : 1));
    int64_t* iown = (int64_t*)malloc(sizeof(int64_t) * (size_t)(I > 0 ? I : 1));
    int rc = 0;
    if (!seen || !uown || !iown) {
        rc = -1;
        goto done;
    }
    for (int32_t x = 0; x < U; ++x) uown[x] = -1;
    for (int32_t x = 0; x < I; ++x) iown[x] = -1;
    if (cell_ptr[0] != 0 || cell_ptr[(int64_t)n_rounds * n_cells] != n) {
        rc = 1;
        goto done;
    }
    for (int32_t rd = 0; rd < n_rounds && rc == 0; ++rd) {
        for (int32_t b = 0; b < n_cells && rc == 0; ++b) {
            const int64_t c = (int64_t)rd * n_cells + b;
            if (cell_ptr[c + 1] < cell_ptr[c]) {
                rc = 1;
                break;
            }
            for (int64_t j = cell_ptr[c]; j < cell_ptr[c + 1]; ++j) {
                const int64_t x = order[j];
                if (x < 0 || x >= n || seen[x]) {
                    rc = 1;
                    break;
                }
                seen[x] = 1;
                if (u[x] < 0 || u[x] >= U || i[x] < 0 || i[x] >= I) {
                    rc = 1;
                    break;
                }
                if (uown[u[x]] >= (int64_t)rd * n_cells && uown[u[x]] != c) {
                    rc = 2;
                    break;
                }
                if (iown[i[x]] >= (int64_t)rd * n_cells && iown[i[x]] != c) {
                    rc = 2;
                    break;
                }
                uown[u[x]] = c;
                iown[i[x]] = c;
            }
        }
    }
    if (rc == 0)
        for (int64_t j = 0; j < n; ++j)
            if (!seen[j]) {
                rc = 1;
                break;
            }
done:
    free(seen);
    free(uown);
    free(iown);
    return rc;
}
