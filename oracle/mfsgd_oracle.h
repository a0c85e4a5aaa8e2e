/*
 * mfsgd_oracle.h -- CPU restatement of the matrix-factorisation SGD hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under matrixfactorizationsgd.java_amd/ (the
 * product) may include, link, import or execute this.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, and only as
 * the checker / the reported CPU baseline.
 *
 * PARITY UNPINNED.  The reference mount (/root/reference) holds a two-line
 * README.md and no source, test, fixture or golden vector
 * (/root/reference/README.md:1-2 is the whole repository).
 * There is therefore no reference file:line for any function below to follow.
 * Each function instead restates a row of SURVEY.md section 8a (a1..a6), which
 * quotes BASELINE.json's north_star, and the arithmetic contract written in
 * DESIGN.md section 3 ("canonical arithmetic").  The only externally pinned
 * pieces are (i) the java.util.Random LCG (JDK specification; KAT
 * new Random(42).nextInt() == -1170105035) and (ii) the hand-computed 2x2 KAT
 * of SURVEY.md section 8c.
 */
#ifndef MFSGD_ORACLE_H
#define MFSGD_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- java.util.Random (JDK spec: 48-bit LCG) ------------------------------ */
typedef struct { uint64_t seed; } mfo_jrandom;
void    mfo_jrandom_init(mfo_jrandom* g, int64_t seed);
int32_t mfo_jrandom_next(mfo_jrandom* g, int bits);
int32_t mfo_jrandom_next_int(mfo_jrandom* g);
float   mfo_jrandom_next_float(mfo_jrandom* g);
double  mfo_jrandom_next_double(mfo_jrandom* g);

/* ---- a4: factor initialisation -------------------------------------------
 * P then Q, row-major, value = nextFloat() * (float)(1.0/sqrt((double)k)),
 * from new java.util.Random(seed).  P is U x k, Q is I x k, dense (stride k). */
void mfo_init_factors(float* P, float* Q, int32_t U, int32_t I, int32_t k, int64_t seed);

/* ---- a1: dot -- canonical arithmetic ----------------
 * Rows are viewed as kp = 4*L floats (L = smallest power of two >= ceil(k/4)),
 * zero padded.  Chunk c holds elements 4c..4c+3:
 *     t0 = a0*b0; t1 = a1*b1; t0 = fma(a2,b2,t0); t1 = fma(a3,b3,t1); s_c = t0+t1
 * then a balanced binary tree over the L chunk sums: for m = 1,2,4,..,L/2:
 *     s_c = s_c + s_(c xor m).                                              */
float mfo_dot(const float* p, const float* q, int32_t k);

/* ---- a1+a2+a3: one rating update in place ---------------------------------
 *   e = r - dot(p,q);  s = fma(-lr, dot, lr*r);  c = 1 - lr*lambda  (all fp32)
 *   p'[f] = fma(s, q[f], c*p[f]);  q'[f] = fma(s, p[f], c*q[f])   (old p,q)
 * Returns e (the error before the update).                                  */
float mfo_sgd_update(float* p, float* q, int32_t k, float r, float lr, float lambda);

/* ---- the per-rating loop, sequential, in the order given ------------------ */
void mfo_sgd_pass(float* P, float* Q, int32_t k,
                  const int32_t* u, const int32_t* i, const float* r,
                  int64_t n, float lr, float lambda);

/* Same loop visiting ratings through a permutation: rating j is order[j]. */
void mfo_sgd_pass_ordered(float* P, float* Q, int32_t k,
                          const int32_t* u, const int32_t* i, const float* r,
                          const int64_t* order, int64_t n, float lr, float lambda);

/* ---- multithreaded CPU path ("port" of the multithread trainer) ------------
 * Executes a block schedule: n_rounds rounds of n_cells cells; cell (rd, b)
 * owns order[cell_ptr[rd*n_cells+b] .. cell_ptr[rd*n_cells+b+1]).  Cells of a
 * round are distributed over n_threads pthreads, with a barrier between
 * rounds.  With a conflict-free schedule the result is bit-identical to
 * mfo_sgd_pass_ordered.  Returns 0, or -1 if threads could not be started.   */
int mfo_sgd_epoch_mt(float* P, float* Q, int32_t k,
                     const int32_t* u, const int32_t* i, const float* r,
                     const int64_t* order, const int64_t* cell_ptr,
                     int32_t n_rounds, int32_t n_cells,
                     float lr, float lambda, int32_t n_threads);

/* ---- the textbook loop (second CPU baseline; NOT the bit-exact contract) ------
 * Plain left-to-right fp32: dot = sum p[f]*q[f]; e = r - dot;
 * p[f] += lr*(e*q[f] - lambda*p[f]); q[f] += lr*(e*p_old[f] - lambda*q[f]).
 * Same visiting order / block schedule / threads as the functions above, so the
 * only difference to the contract is the rounding (RMSE gap ~1e-7).          */
void mfo_textbook_pass_ordered(float* P, float* Q, int32_t k,
                               const int32_t* u, const int32_t* i, const float* r,
                               const int64_t* order, int64_t n, float lr, float lambda);
int mfo_textbook_epoch_mt(float* P, float* Q, int32_t k,
                          const int32_t* u, const int32_t* i, const float* r,
                          const int64_t* order, const int64_t* cell_ptr,
                          int32_t n_rounds, int32_t n_cells,
                          float lr, float lambda, int32_t n_threads);

/* ---- a6: RMSE = sqrt(sum (r - dot)^2 / n), fp32 dot, fp64 accumulation ---- */
double mfo_sse(const float* P, const float* Q, int32_t k,
               const int32_t* u, const int32_t* i, const float* r, int64_t n);
double mfo_rmse(const float* P, const float* Q, int32_t k,
                const int32_t* u, const int32_t* i, const float* r, int64_t n);

/* ---- predict: out[j] = dot(P[u[j]], Q[i[j]]) ------------------------------ */
void mfo_predict(const float* P, const float* Q, int32_t k,
                 const int32_t* u, const int32_t* i, float* out, int64_t n);

/* ---- schedule checker -----------------------------------------------------
 * Verifies what makes "parallel == sequential" true for a block schedule:
 *  (1) order is a permutation of 0..n-1;
 *  (2) within every round, no user and no item appears in two different cells.
 * Returns 0 if both hold, 1 if (1) fails, 2 if (2) fails, -1 on allocation
 * failure.                                                                   */
int mfo_check_block_schedule(const int32_t* u, const int32_t* i, int64_t n,
                             int32_t U, int32_t I,
                             const int64_t* order, const int64_t* cell_ptr,
                             int32_t n_rounds, int32_t n_cells);

#ifdef __cplusplus
}
#endif
#endif
