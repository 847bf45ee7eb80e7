/*
 * dqp_oracle.c -- CPU restatement of the reference's batched differentiable QP path.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is the parity oracle and the "port" CPU baseline.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it; the
 * product (diff-qp-mpc_amd/) never links, imports or falls back to anything in oracle/.
 *
 * It restates, in plain C with batch-synchronous control flow (so the reference's
 * batch-coupled termination and step rules are reproduced exactly), these reference
 * functions (paths relative to the reference checkout):
 *
 *   qpth/solvers/pdipm/batch.py:377-428  pre_factor_kkt   -> pre_factor()
 *   qpth/solvers/pdipm/batch.py:434-469  factor_kkt       -> factor_T()
 *   qpth/solvers/pdipm/batch.py:351-374  solve_kkt        -> solve_kkt()
 *   qpth/solvers/pdipm/batch.py:46-208   forward          -> dqp_oracle_qp_forward()
 *   qpth/solvers/pdipm/batch.py:211-214  get_step         -> step_ratio()/step_from()
 *   qpth/qp.py:128-183                   QPFunctionFn.backward -> dqp_oracle_qp_backward()
 *   qpth/qp.py:195-217                   DenseQPFunction.preprocess -> dense_build_K()
 *   qpth/solvers/pdipm/batch_LU.py:29-201  forward        -> dqp_oracle_dense_forward()
 *   qpth/solvers/pdipm/batch_LU.py:204-210 get_step       -> (dv==0 -> 1 variant)
 *   qpth/solvers/pdipm/batch_LU.py:212-244 solve_kkt      -> dense_solve_kkt()
 *   qpth/qp.py:239-270                   Solver.backward  -> dqp_oracle_dense_backward()
 *
 * torch.linalg.lu_factor / lu_solve (third-party, PyTorch 2.0.1 pinned by the reference's
 * env_khai.yml:146; 2.10 in this image) are restated as textbook row-pivoted LU.
 *
 * Pinned: tests/test_oracle_golden.py checks every function here against the fixtures in
 * tests/golden/ (.npz files), which were produced by importing the reference itself
 * (tests/golden/make_golden.py).
 *
 * Layout: everything batch-major, row-major, contiguous, fp64.
 *   Q (B,nz,nz) p (B,nz) G (B,nineq,nz) h (B,nineq) A (B,neq,nz) b (B,neq)
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define API __attribute__((visibility("default")))

/* ------------------------------------------------------------------ small dense LA --- */

/* Row-pivoted LU in place (row-major).  Returns 0, or k+1 for the first exactly-zero
 * pivot (torch.linalg.lu_factor raises in that case). */
static int lu_factor(int n, double *a, int *piv)
{
    int info = 0;
    for (int k = 0; k < n; ++k) {
        int p = k;
        double mv = fabs(a[k * n + k]);
        for (int i = k + 1; i < n; ++i) {
            double v = fabs(a[i * n + k]);
            if (v > mv) { mv = v; p = i; }
        }
        piv[k] = p;
        if (p != k)
            for (int j = 0; j < n; ++j) {
                double t = a[k * n + j]; a[k * n + j] = a[p * n + j]; a[p * n + j] = t;
            }
        double akk = a[k * n + k];
        if (akk == 0.0) { if (!info) info = k + 1; continue; }
        for (int i = k + 1; i < n; ++i) {
            double l = a[i * n + k] / akk;
            a[i * n + k] = l;
            if (l != 0.0)
                for (int j = k + 1; j < n; ++j) a[i * n + j] -= l * a[k * n + j];
        }
    }
    return info;
}

static void lu_solve(int n, const double *a, const int *piv, double *b)
{
    for (int k = 0; k < n; ++k)
        if (piv[k] != k) { double t = b[k]; b[k] = b[piv[k]]; b[piv[k]] = t; }
    for (int i = 1; i < n; ++i) {
        double s = b[i];
        for (int j = 0; j < i; ++j) s -= a[i * n + j] * b[j];
        b[i] = s;
    }
    for (int i = n - 1; i >= 0; --i) {
        double s = b[i];
        for (int j = i + 1; j < n; ++j) s -= a[i * n + j] * b[j];
        b[i] = s / a[i * n + i];
    }
}

/* y = M x (M is m x n row-major) */
static void mv(int m, int n, const double *M, const double *x, double *y)
{
    for (int i = 0; i < m; ++i) {
        double s = 0.0;
        for (int j = 0; j < n; ++j) s += M[i * n + j] * x[j];
        y[i] = s;
    }
}
/* y = M^T x (M is m x n row-major, x has m entries, y has n) */
static void mtv(int m, int n, const double *M, const double *x, double *y)
{
    for (int j = 0; j < n; ++j) y[j] = 0.0;
    for (int i = 0; i < m; ++i) {
        double xi = x[i];
        for (int j = 0; j < n; ++j) y[j] += M[i * n + j] * xi;
    }
}
static double nrm2(int n, const double *x)
{
    double s = 0.0;
    for (int i = 0; i < n; ++i) s += x[i] * x[i];
    return sqrt(s);
}
/* torch.min / torch.max propagate NaN */
static double tmin(double a, double b) { return (isnan(a) || isnan(b)) ? NAN : (a < b ? a : b); }
static double tmax(double a, double b) { return (isnan(a) || isnan(b)) ? NAN : (a > b ? a : b); }

/* ------------------------------------------------------- block-LU KKT (batch.py) ----- */

typedef struct {
    int nz, nineq, neq;
    const double *Q, *p, *G, *h, *A, *b;
    double *Qlu; int *Qpiv;         /* LU(Q)                        batch.py:381 */
    double *R;                      /* Schur complement (constant)  batch.py:399,420 */
    double *S11lu; int *S11piv;     /* LU(A Q^-1 A^T)               batch.py:407 */
    double *GiQAt;                  /* G Q^-1 A^T  (nineq x neq)    batch.py:405 */
    double *Tlu; int *Tpiv;         /* LU(R + diag(1/d))            batch.py:444-446 */
    double *d;
    double *x, *s, *z, *y;
    double *bx, *bs, *bz, *by; double bres; int has_best;
    double *rx, *rs, *rz, *ry;
    double *dxa, *dsa, *dza, *dya, *dxc, *dsc, *dzc, *dyc;
    double *t1, *t2, *t3;           /* scratch, each max(nz, nineq+neq) */
    double mu, resid, amax_z, amax_s, alpha;
    int info;
} qpws;

static size_t qpws_doubles(int nz, int nineq, int neq)
{
    int m = nz > nineq + neq ? nz : nineq + neq;
    return (size_t)nz * nz + (size_t)nineq * nineq * 2 + (size_t)neq * neq + (size_t)nineq * neq
         + (size_t)nineq                                   /* d */
         + 2 * ((size_t)nz + 2 * nineq + neq)              /* x s z y + best */
         + ((size_t)nz + 2 * nineq + neq)                  /* r* */
         + 2 * ((size_t)nz + 2 * nineq + neq)              /* d*a d*c */
         + 3 * (size_t)m + 64;
}

static void qpws_carve(qpws *w, double *buf, int *ibuf, int nz, int nineq, int neq)
{
    int m = nz > nineq + neq ? nz : nineq + neq;
    w->nz = nz; w->nineq = nineq; w->neq = neq;
    double *q = buf;
#define TAKE(ptr, n) do { ptr = q; q += (n); } while (0)
    TAKE(w->Qlu, (size_t)nz * nz); TAKE(w->R, (size_t)nineq * nineq);
    TAKE(w->Tlu, (size_t)nineq * nineq); TAKE(w->S11lu, (size_t)neq * neq);
    TAKE(w->GiQAt, (size_t)nineq * neq); TAKE(w->d, nineq);
    TAKE(w->x, nz); TAKE(w->s, nineq); TAKE(w->z, nineq); TAKE(w->y, neq);
    TAKE(w->bx, nz); TAKE(w->bs, nineq); TAKE(w->bz, nineq); TAKE(w->by, neq);
    TAKE(w->rx, nz); TAKE(w->rs, nineq); TAKE(w->rz, nineq); TAKE(w->ry, neq);
    TAKE(w->dxa, nz); TAKE(w->dsa, nineq); TAKE(w->dza, nineq); TAKE(w->dya, neq);
    TAKE(w->dxc, nz); TAKE(w->dsc, nineq); TAKE(w->dzc, nineq); TAKE(w->dyc, neq);
    TAKE(w->t1, m); TAKE(w->t2, m); TAKE(w->t3, m);
#undef TAKE
    w->Qpiv = ibuf; w->S11piv = ibuf + nz; w->Tpiv = ibuf + nz + neq;
    w->has_best = 0; w->bres = 0.0; w->info = 0;
}

/* batch.py:377-428.  Returns nonzero if LU(Q) hits a zero pivot. */
static int pre_factor(qpws *w)
{
    const int nz = w->nz, nineq = w->nineq, neq = w->neq;
    memcpy(w->Qlu, w->Q, sizeof(double) * nz * nz);
    if (lu_factor(nz, w->Qlu, w->Qpiv)) return 1;
    double *col = w->t1, *tmp = w->t2;
    /* R = G Q^-1 G^T */
    for (int j = 0; j < nineq; ++j) {
        for (int k = 0; k < nz; ++k) col[k] = w->G[j * nz + k];      /* column j of G^T */
        lu_solve(nz, w->Qlu, w->Qpiv, col);
        for (int i = 0; i < nineq; ++i) {
            double s = 0.0;
            for (int k = 0; k < nz; ++k) s += w->G[i * nz + k] * col[k];
            w->R[i * nineq + j] = s;
        }
    }
    if (neq > 0) {
        /* invQ_AT column by column -> A_invQ_AT (S11) and G_invQ_AT */
        for (int j = 0; j < neq; ++j) {
            for (int k = 0; k < nz; ++k) col[k] = w->A[j * nz + k];
            lu_solve(nz, w->Qlu, w->Qpiv, col);
            for (int i = 0; i < neq; ++i) {
                double s = 0.0;
                for (int k = 0; k < nz; ++k) s += w->A[i * nz + k] * col[k];
                w->S11lu[i * neq + j] = s;
            }
            for (int i = 0; i < nineq; ++i) {
                double s = 0.0;
                for (int k = 0; k < nz; ++k) s += w->G[i * nz + k] * col[k];
                w->GiQAt[i * neq + j] = s;
            }
        }
        if (lu_factor(neq, w->S11lu, w->S11piv)) return 2;
        /* T = S11^-1 (G_invQ_AT)^T ; R -= G_invQ_AT T          batch.py:414,420 */
        for (int j = 0; j < nineq; ++j) {
            for (int k = 0; k < neq; ++k) tmp[k] = w->GiQAt[j * neq + k];
            lu_solve(neq, w->S11lu, w->S11piv, tmp);
            for (int i = 0; i < nineq; ++i) {
                double s = 0.0;
                for (int k = 0; k < neq; ++k) s += w->GiQAt[i * neq + k] * tmp[k];
                w->R[i * nineq + j] -= s;
            }
        }
    }
    return 0;
}

/* batch.py:434-469: LU(R + diag(1/d)).  Pivot bookkeeping of S_LU (re-permuting S21) is an
 * implementation detail of storing one packed LU; the linear map applied is S^-1. */
static int factor_T(qpws *w)
{
    const int n = w->nineq;
    memcpy(w->Tlu, w->R, sizeof(double) * n * n);
    for (int i = 0; i < n; ++i) w->Tlu[i * n + i] += 1.0 / w->d[i];
    return lu_factor(n, w->Tlu, w->Tpiv);
}

/* batch.py:351-374 */
static void solve_kkt(qpws *w, const double *rx, const double *rs, const double *rz,
                      const double *ry, double *dx, double *ds, double *dz, double *dy)
{
    const int nz = w->nz, nineq = w->nineq, neq = w->neq;
    double *invQ_rx = w->t1, *hh = w->t2, *g1 = w->t3;
    double *hy = hh, *hz = hh + neq;
    memcpy(invQ_rx, rx, sizeof(double) * nz);
    lu_solve(nz, w->Qlu, w->Qpiv, invQ_rx);
    if (neq > 0) {
        mv(neq, nz, w->A, invQ_rx, hy);
        for (int i = 0; i < neq; ++i) hy[i] -= ry[i];
    }
    mv(nineq, nz, w->G, invQ_rx, hz);
    for (int i = 0; i < nineq; ++i) hz[i] += rs[i] / w->d[i] - rz[i];
    /* w = -S^-1 h via the block factorisation [S11 S12; S21 S22+D^-1] */
    double *wy = dy, *wz = dz;
    if (neq > 0) {
        for (int i = 0; i < neq; ++i) wy[i] = hy[i];
        lu_solve(neq, w->S11lu, w->S11piv, wy);                 /* a = S11^-1 hy */
        for (int i = 0; i < nineq; ++i) {
            double s = hz[i];
            for (int k = 0; k < neq; ++k) s -= w->GiQAt[i * neq + k] * wy[k];
            wz[i] = s;
        }
    } else {
        for (int i = 0; i < nineq; ++i) wz[i] = hz[i];
    }
    lu_solve(nineq, w->Tlu, w->Tpiv, wz);
    if (neq > 0) {
        for (int k = 0; k < neq; ++k) {
            double s = hy[k];
            for (int i = 0; i < nineq; ++i) s -= w->GiQAt[i * neq + k] * wz[i];
            wy[k] = s;
        }
        lu_solve(neq, w->S11lu, w->S11piv, wy);
        for (int k = 0; k < neq; ++k) wy[k] = -wy[k];
    }
    for (int i = 0; i < nineq; ++i) wz[i] = -wz[i];
    /* g1 = -rx - G^T wz - A^T wy ; dx = Q^-1 g1 ; ds = (-rs - wz)/d */
    mtv(nineq, nz, w->G, wz, g1);
    for (int k = 0; k < nz; ++k) g1[k] = -rx[k] - g1[k];
    if (neq > 0) {
        mtv(neq, nz, w->A, wy, invQ_rx);
        for (int k = 0; k < nz; ++k) g1[k] -= invQ_rx[k];
    }
    lu_solve(nz, w->Qlu, w->Qpiv, g1);
    memcpy(dx, g1, sizeof(double) * nz);
    for (int i = 0; i < nineq; ++i) ds[i] = (-rs[i] - wz[i]) / w->d[i];
}

/* get_step, first half: a = -v/dv (all entries); returns torch-style max over them.
 * dv0_is_one selects batch_LU.py's variant (a[dv == 0] = 1). */
static double step_ratio(int n, const double *v, const double *dv, double *a, int dv0_is_one)
{
    double mx = -INFINITY;
    for (int i = 0; i < n; ++i) {
        a[i] = -v[i] / dv[i];
        if (dv0_is_one && dv[i] == 0.0) a[i] = 1.0;
        mx = tmax(mx, a[i]);
    }
    return mx;
}
/* get_step, second half: a[dv>0] = max(1.0, a.max()) with a.max() taken over the WHOLE
 * batch tensor (python's max(1.0, nan) == 1.0); return a.min(1). */
static double step_from(int n, const double *a, const double *dv, double amax_global)
{
    double M = (amax_global > 1.0) ? amax_global : 1.0;
    double mn = INFINITY;
    for (int i = 0; i < n; ++i) mn = tmin(mn, dv[i] > 0.0 ? M : a[i]);
    return mn;
}

static void residuals(qpws *w)
{
    const int nz = w->nz, nineq = w->nineq, neq = w->neq;
    /* rx = A^T y + G^T z + Q x + p      (cost_grad(x) = Qx + p)   batch.py:93-96 */
    mv(nz, nz, w->Q, w->x, w->rx);
    for (int k = 0; k < nz; ++k) w->rx[k] += w->p[k];
    mtv(nineq, nz, w->G, w->z, w->t1);
    for (int k = 0; k < nz; ++k) w->rx[k] += w->t1[k];
    if (neq > 0) {
        mtv(neq, nz, w->A, w->y, w->t1);
        for (int k = 0; k < nz; ++k) w->rx[k] += w->t1[k];
        mv(neq, nz, w->A, w->x, w->ry);                      /* dyn_res(x) = Ax - b */
        for (int i = 0; i < neq; ++i) w->ry[i] -= w->b[i];
    }
    mv(nineq, nz, w->G, w->x, w->rz);
    double sz = 0.0;
    for (int i = 0; i < nineq; ++i) {
        w->rz[i] += w->s[i] - w->h[i];
        sz += w->s[i] * w->z[i];
    }
    w->mu = fabs(sz / nineq);
    double pri = nrm2(nineq, w->rz) + (neq > 0 ? nrm2(neq, w->ry) : 0.0);
    w->resid = pri + nrm2(nz, w->rx) + nineq * w->mu;
}

static void shift_ge_one(int n, double *v)   /* batch.py:76-86 */
{
    double m = INFINITY;
    for (int i = 0; i < n; ++i) m = tmin(m, v[i]);
    if (m < 0) for (int i = 0; i < n; ++i) v[i] -= m - 1.0;
}

static void save_best(qpws *w)
{
    memcpy(w->bx, w->x, sizeof(double) * w->nz);
    memcpy(w->bs, w->s, sizeof(double) * w->nineq);
    memcpy(w->bz, w->z, sizeof(double) * w->nineq);
    if (w->neq) memcpy(w->by, w->y, sizeof(double) * w->neq);
    w->bres = w->resid;
}

static int set_threads(int nthreads)
{
#ifdef _OPENMP
    if (nthreads <= 0) nthreads = omp_get_max_threads();
    return nthreads;
#else
    (void)nthreads; return 1;
#endif
}

API int dqp_oracle_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/*
 * QPFunction forward (qp.py:24-126 with solver == PDIPM_BATCHED, batch.py:46-208).
 * Outputs: zhat (B,nz), lam (B,nineq), nu (B,neq), slack (B,nineq); iters_out[0] = number
 * of PDIPM iterations the batch executed; best_resid (B) may be NULL.
 * Returns 0; 1 if LU(Q) failed (reference raises RuntimeError, batch.py:381-388).
 */
static int qp_forward_impl(int B, int nz, int nineq, int neq,
                           const double *Q, const double *p, const double *G,
                           const double *h, const double *A, const double *b,
                           double eps, int notImprovedLim, int maxIter,
                           double *zhat, double *lam, double *nu, double *slack,
                           int *iters_out, double *best_resid, double *resid_hist,
                           int nthreads, int dv0_guard)
{
    /* resid_hist (B, maxIter) or NULL: diagnostic trace of `resids` (batch.py:108) per
     * iteration, NaN where the batch had already stopped. */
    nthreads = set_threads(nthreads);
    if (resid_hist) for (size_t k = 0; k < (size_t)B * maxIter; ++k) resid_hist[k] = NAN;
    const size_t nd = qpws_doubles(nz, nineq, neq);
    const size_t ni = (size_t)nz + neq + nineq + 8;
    double *buf = (double *)malloc(sizeof(double) * nd * B);
    int *ibuf = (int *)malloc(sizeof(int) * ni * B);
    qpws *W = (qpws *)malloc(sizeof(qpws) * B);
    if (!buf || !ibuf || !W) { free(buf); free(ibuf); free(W); return -1; }
    int fail = 0;

#pragma omp parallel for num_threads(nthreads) schedule(static) reduction(| : fail)
    for (int i = 0; i < B; ++i) {
        qpws *w = &W[i];
        qpws_carve(w, buf + nd * i, ibuf + ni * i, nz, nineq, neq);
        w->Q = Q + (size_t)i * nz * nz; w->p = p + (size_t)i * nz;
        w->G = G + (size_t)i * nineq * nz; w->h = h + (size_t)i * nineq;
        w->A = neq ? A + (size_t)i * neq * nz : NULL; w->b = neq ? b + (size_t)i * neq : NULL;
        if (pre_factor(w)) { fail |= 1; continue; }
        /* initial point: d = 1, solve_kkt(p, 0, -h, -b)        batch.py:60-74 */
        for (int k = 0; k < nineq; ++k) w->d[k] = 1.0;
        factor_T(w);
        for (int k = 0; k < nineq; ++k) { w->rs[k] = 0.0; w->rz[k] = -w->h[k]; }
        for (int k = 0; k < neq; ++k) w->ry[k] = -w->b[k];
        solve_kkt(w, w->p, w->rs, w->rz, w->ry, w->x, w->s, w->z, w->y);
        shift_ge_one(nineq, w->s);
        shift_ge_one(nineq, w->z);
    }
    int iters = 0;
    if (!fail) {
        int nNotImproved = 0;
        for (int it = 0; it < maxIter; ++it) {
            int any_info = 0;
#pragma omp parallel for num_threads(nthreads) schedule(static) reduction(| : any_info)
            for (int i = 0; i < B; ++i) {
                qpws *w = &W[i];
                residuals(w);
                memcpy(w->rs, w->z, sizeof(double) * nineq);            /* rs = z */
                for (int k = 0; k < nineq; ++k) w->d[k] = w->z[k] / w->s[k];
                w->info = factor_T(w);
                any_info |= (w->info != 0);
                if (resid_hist) resid_hist[(size_t)i * maxIter + it] = w->resid;
            }
            if (any_info) break;      /* try/except around factor_kkt: return best */
            iters = it + 1;
            /* best-iterate tracking                                  batch.py:119-140 */
            int improved = 0, first = !W[0].has_best;
            double best_max = -INFINITY, mu_min = INFINITY;
            for (int i = 0; i < B; ++i) {
                qpws *w = &W[i];
                if (!w->has_best) { save_best(w); w->has_best = 1; }
                else if (w->resid < w->bres) { save_best(w); improved = 1; }
                best_max = tmax(best_max, w->bres);
                mu_min = tmin(mu_min, w->mu);
            }
            if (first) nNotImproved = 0;
            else if (improved) nNotImproved = 0;
            else nNotImproved += 1;
            if (nNotImproved == notImprovedLim || best_max < eps || mu_min > 1e32) break;

            /* affine direction                                      batch.py:151 */
            double amz = -INFINITY, ams = -INFINITY;
#pragma omp parallel for num_threads(nthreads) schedule(static)
            for (int i = 0; i < B; ++i) {
                qpws *w = &W[i];
                solve_kkt(w, w->rx, w->rs, w->rz, w->ry, w->dxa, w->dsa, w->dza, w->dya);
                w->amax_z = step_ratio(nineq, w->z, w->dza, w->dzc, dv0_guard);  /* dzc/dsc as scratch */
                w->amax_s = step_ratio(nineq, w->s, w->dsa, w->dsc, dv0_guard);
            }
            for (int i = 0; i < B; ++i) { amz = tmax(amz, W[i].amax_z); ams = tmax(ams, W[i].amax_s); }
#pragma omp parallel for num_threads(nthreads) schedule(static)
            for (int i = 0; i < B; ++i) {
                qpws *w = &W[i];
                double alpha = tmin(tmin(step_from(nineq, w->dzc, w->dza, amz),
                                         step_from(nineq, w->dsc, w->dsa, ams)), 1.0);
                double t3 = 0.0, t4 = 0.0;
                for (int k = 0; k < nineq; ++k) {
                    t3 += (w->s[k] + alpha * w->dsa[k]) * (w->z[k] + alpha * w->dza[k]);
                    t4 += w->s[k] * w->z[k];
                }
                double sig = t3 / t4; sig = sig * sig * sig;
                /* corrector rhs: rx = 0, rs = (-mu sig + ds_aff dz_aff)/s, rz = ry = 0 */
                for (int k = 0; k < nz; ++k) w->rx[k] = 0.0;   /* rx..ry reused as rhs storage */
                for (int k = 0; k < nineq; ++k) {
                    w->rs[k] = (-w->mu * sig + w->dsa[k] * w->dza[k]) / w->s[k];
                    w->rz[k] = 0.0;
                }
                for (int k = 0; k < neq; ++k) w->ry[k] = 0.0;
                solve_kkt(w, w->rx, w->rs, w->rz, w->ry, w->dxc, w->dsc, w->dzc, w->dyc);
                for (int k = 0; k < nz; ++k) w->dxa[k] += w->dxc[k];
                for (int k = 0; k < nineq; ++k) { w->dsa[k] += w->dsc[k]; w->dza[k] += w->dzc[k]; }
                for (int k = 0; k < neq; ++k) w->dya[k] += w->dyc[k];
                w->amax_z = step_ratio(nineq, w->z, w->dza, w->dzc, dv0_guard);
                w->amax_s = step_ratio(nineq, w->s, w->dsa, w->dsc, dv0_guard);
            }
            amz = -INFINITY; ams = -INFINITY;
            for (int i = 0; i < B; ++i) { amz = tmax(amz, W[i].amax_z); ams = tmax(ams, W[i].amax_s); }
#pragma omp parallel for num_threads(nthreads) schedule(static)
            for (int i = 0; i < B; ++i) {
                qpws *w = &W[i];
                double alpha = tmin(0.999 * tmin(step_from(nineq, w->dzc, w->dza, amz),
                                                 step_from(nineq, w->dsc, w->dsa, ams)), 1.0);
                for (int k = 0; k < nz; ++k) w->x[k] += alpha * w->dxa[k];
                for (int k = 0; k < nineq; ++k) { w->s[k] += alpha * w->dsa[k]; w->z[k] += alpha * w->dza[k]; }
                for (int k = 0; k < neq; ++k) w->y[k] += alpha * w->dya[k];
            }
        }
        for (int i = 0; i < B; ++i) {
            qpws *w = &W[i];
            if (!w->has_best) { residuals(w); save_best(w); }
            memcpy(zhat + (size_t)i * nz, w->bx, sizeof(double) * nz);
            memcpy(lam + (size_t)i * nineq, w->bz, sizeof(double) * nineq);
            memcpy(slack + (size_t)i * nineq, w->bs, sizeof(double) * nineq);
            if (neq) memcpy(nu + (size_t)i * neq, w->by, sizeof(double) * neq);
            if (best_resid) best_resid[i] = w->bres;
        }
    }
    if (iters_out) iters_out[0] = iters;
    free(buf); free(ibuf); free(W);
    return fail ? 1 : 0;
}

API int dqp_oracle_qp_forward(int B, int nz, int nineq, int neq,
                              const double *Q, const double *p, const double *G,
                              const double *h, const double *A, const double *b,
                              double eps, int notImprovedLim, int maxIter,
                              double *zhat, double *lam, double *nu, double *slack,
                              int *iters_out, double *best_resid, double *resid_hist,
                              int nthreads)
{
    return qp_forward_impl(B, nz, nineq, neq, Q, p, G, h, A, b, eps, notImprovedLim, maxIter, zhat,
                           lam, nu, slack, iters_out, best_resid, resid_hist, nthreads, 0);
}

/*
 * The same forward with get_step taken from the reference's OTHER PDIPM module
 * (batch_LU.py:204-210: `a[dv == 0] = 1.0` before the min) instead of batch.py:211-214.
 * Why it exists: batch.py's get_step computes -v/dv with no guard, so a step component that
 * rounds to exactly 0.0 (typical for the slack of an inactive box bound late in the solve:
 * dz_aff == -z bit for bit) gives -inf, alpha = -inf, 0 * inf = NaN, and that sample's iterate
 * is NaN for the rest of the batch's iterations -- its returned "best" is frozen one or two
 * iterations before convergence (mu ~ 1e-14 instead of ~ 1e-17).  Which samples hit an exact
 * zero is a floating-point accident of the reference's arithmetic order, not a property of the
 * QP; the fused kernels (different coordinates, reciprocal-based step rule) sail past it.
 * Parity for those samples is therefore checked against this variant, which is pinned by
 * tests/golden/Mz_guard_b8.npz (the reference itself run with pdipm_b.get_step replaced by
 * batch_LU.get_step, see tests/golden/make_golden_guard.py).
 */
API int dqp_oracle_qp_forward_guarded(int B, int nz, int nineq, int neq,
                                      const double *Q, const double *p, const double *G,
                                      const double *h, const double *A, const double *b,
                                      double eps, int notImprovedLim, int maxIter,
                                      double *zhat, double *lam, double *nu, double *slack,
                                      int *iters_out, double *best_resid, double *resid_hist,
                                      int nthreads)
{
    return qp_forward_impl(B, nz, nineq, neq, Q, p, G, h, A, b, eps, notImprovedLim, maxIter, zhat,
                           lam, nu, slack, iters_out, best_resid, resid_hist, nthreads, 1);
}

/* dQ = 1/2 (dx z^T + z dx^T), dp = dx, dG = dlam z^T + lam dx^T, dh = -dlam,
 * dA = dnu z^T + nu dx^T, db = -dnu                     qp.py:158-181 / qp.py:257-268 */
static void kkt_grads(int nz, int nineq, int neq, const double *zhat, const double *lam,
                      const double *nu, const double *dx, const double *dlam, const double *dnu,
                      double *dQ, double *dp, double *dG, double *dh, double *dA, double *db)
{
    for (int i = 0; i < nz; ++i) {
        dp[i] = dx[i];
        for (int j = 0; j < nz; ++j) dQ[i * nz + j] = 0.5 * (dx[i] * zhat[j] + zhat[i] * dx[j]);
    }
    for (int i = 0; i < nineq; ++i) {
        dh[i] = -dlam[i];
        for (int j = 0; j < nz; ++j) dG[i * nz + j] = dlam[i] * zhat[j] + lam[i] * dx[j];
    }
    for (int i = 0; i < neq; ++i) {
        db[i] = -dnu[i];
        for (int j = 0; j < nz; ++j) dA[i * nz + j] = dnu[i] * zhat[j] + nu[i] * dx[j];
    }
}

/* QPFunctionFn.backward (qp.py:128-183), per-sample gradients (no .mean(0): the caller
 * applies it for parameters that were broadcast). */
API int dqp_oracle_qp_backward(int B, int nz, int nineq, int neq,
                               const double *Q, const double *G, const double *A,
                               const double *zhat, const double *lam, const double *nu,
                               const double *slack, const double *dl_dzhat,
                               double *dQ, double *dp, double *dG, double *dh,
                               double *dA, double *db, int nthreads)
{
    nthreads = set_threads(nthreads);
    const size_t nd = qpws_doubles(nz, nineq, neq);
    const size_t ni = (size_t)nz + neq + nineq + 8;
    double *buf = (double *)malloc(sizeof(double) * nd * B);
    int *ibuf = (int *)malloc(sizeof(int) * ni * B);
    if (!buf || !ibuf) { free(buf); free(ibuf); return -1; }
    int fail = 0;
#pragma omp parallel for num_threads(nthreads) schedule(static) reduction(| : fail)
    for (int i = 0; i < B; ++i) {
        qpws ws, *w = &ws;
        qpws_carve(w, buf + nd * i, ibuf + ni * i, nz, nineq, neq);
        w->Q = Q + (size_t)i * nz * nz; w->G = G + (size_t)i * nineq * nz;
        w->A = neq ? A + (size_t)i * neq * nz : NULL;
        if (pre_factor(w)) { fail |= 1; continue; }
        const double *l = lam + (size_t)i * nineq, *sl = slack + (size_t)i * nineq;
        for (int k = 0; k < nineq; ++k) {
            double lc = l[k] < 1e-8 ? 1e-8 : l[k], sc = sl[k] < 1e-8 ? 1e-8 : sl[k];
            w->d[k] = lc / sc;                                   /* qp.py:149 */
            w->rs[k] = 0.0; w->rz[k] = 0.0;
        }
        for (int k = 0; k < neq; ++k) w->ry[k] = 0.0;
        factor_T(w);
        solve_kkt(w, dl_dzhat + (size_t)i * nz, w->rs, w->rz, w->ry, w->dxa, w->dsa, w->dza, w->dya);
        kkt_grads(nz, nineq, neq, zhat + (size_t)i * nz, l, neq ? nu + (size_t)i * neq : NULL,
                  w->dxa, w->dza, w->dya,
                  dQ + (size_t)i * nz * nz, dp + (size_t)i * nz, dG + (size_t)i * nineq * nz,
                  dh + (size_t)i * nineq, neq ? dA + (size_t)i * neq * nz : NULL,
                  neq ? db + (size_t)i * neq : NULL);
    }
    free(buf); free(ibuf);
    return fail ? 1 : 0;
}

/* --------------------------------------------- full-KKT PDIPM (batch_LU.py / Dense) --- */

typedef struct {
    int nz, nineq, neq, N;
    const double *Q, *p, *G, *h, *A, *b;
    double *K, *Kt, *Klu, *bK; int *piv;
    double *x, *s, *z, *y, *bx, *bs, *bz, *by; double bres; int has_best;
    double *r, *l, *res, *l2;           /* N-vectors */
    double *rx, *rs, *rz, *ry;
    double *da, *dc, *az, *as;          /* da/dc: N-vectors [dx ds dz dy] */
    double mu, resid, amax_z, amax_s;
} dnws;

static size_t dnws_doubles(int nz, int nineq, int neq)
{
    size_t N = (size_t)nz + 2 * nineq + neq;
    return 4 * N * N + 12 * N + 64;
}

/* qp.py:195-217 */
static void dense_build_K(dnws *w)
{
    const int nz = w->nz, m = w->nineq, q = w->neq, N = w->N;
    memset(w->K, 0, sizeof(double) * N * N);
    for (int i = 0; i < nz; ++i)
        for (int j = 0; j < nz; ++j) w->K[i * N + j] = w->Q[i * nz + j];
    for (int i = 0; i < m; ++i)
        for (int j = 0; j < nz; ++j) {
            w->K[j * N + nz + m + i] = w->G[i * nz + j];       /* G^T block */
            w->K[(nz + m + i) * N + j] = w->G[i * nz + j];     /* G block   */
        }
    for (int i = 0; i < q; ++i)
        for (int j = 0; j < nz; ++j) {
            w->K[j * N + nz + 2 * m + i] = w->A[i * nz + j];
            w->K[(nz + 2 * m + i) * N + j] = w->A[i * nz + j];
        }
    for (int i = 0; i < m; ++i) {
        w->K[(nz + i) * N + nz + i] = 1.0;          /* Z diag  (Zidx) */
        w->K[(nz + i) * N + nz + m + i] = 1.0;      /* S diag  (Sidx) */
        w->K[(nz + m + i) * N + nz + i] = 1.0;      /* I in the G row */
    }
}

/* batch_LU.py:212-244: LU(Ktilde), solve, one refinement step against K. */
static int dense_solve_kkt(dnws *w, const double *K, const double *Kt,
                           const double *rx, const double *rs, const double *rz,
                           const double *ry, double *out)
{
    const int nz = w->nz, m = w->nineq, q = w->neq, N = w->N;
    double *r = w->r, *l = w->l, *res = w->res;
    for (int i = 0; i < nz; ++i) r[i] = -rx[i];
    for (int i = 0; i < m; ++i) { r[nz + i] = -rs[i]; r[nz + m + i] = -rz[i]; }
    for (int i = 0; i < q; ++i) r[nz + 2 * m + i] = -ry[i];
    memcpy(w->Klu, Kt, sizeof(double) * N * N);
    int info = lu_factor(N, w->Klu, w->piv);
    if (info) return info;
    memcpy(l, r, sizeof(double) * N);
    lu_solve(N, w->Klu, w->piv, l);
    mv(N, N, K, l, res);
    for (int i = 0; i < N; ++i) res[i] = r[i] - res[i];
    lu_solve(N, w->Klu, w->piv, res);
    for (int i = 0; i < N; ++i) out[i] = l[i] + res[i];
    return 0;
}

static void dense_set_diag(dnws *w, double *K, const double *zv, const double *sv, double zadd)
{
    const int nz = w->nz, m = w->nineq, N = w->N;
    for (int i = 0; i < m; ++i) {
        K[(nz + i) * N + nz + i] = zv[i] + zadd;
        K[(nz + i) * N + nz + m + i] = sv[i];
    }
}

static void dense_residuals(dnws *w)
{
    const int nz = w->nz, m = w->nineq, q = w->neq;
    mv(nz, nz, w->Q, w->x, w->rx);
    for (int k = 0; k < nz; ++k) w->rx[k] += w->p[k];
    mtv(m, nz, w->G, w->z, w->l2);
    for (int k = 0; k < nz; ++k) w->rx[k] += w->l2[k];
    if (q > 0) {
        mtv(q, nz, w->A, w->y, w->l2);
        for (int k = 0; k < nz; ++k) w->rx[k] += w->l2[k];
        mv(q, nz, w->A, w->x, w->ry);
        for (int i = 0; i < q; ++i) w->ry[i] -= w->b[i];
    }
    mv(m, nz, w->G, w->x, w->rz);
    double sz = 0.0;
    for (int i = 0; i < m; ++i) {
        w->rz[i] += w->s[i] - w->h[i];
        w->rs[i] = w->s[i] * w->z[i];                          /* batch_LU.py:95 */
        sz += w->s[i] * w->z[i];
    }
    w->mu = fabs(sz / m);
    double pri = nrm2(m, w->rz) + (q > 0 ? nrm2(q, w->ry) : 0.0);
    w->resid = pri + nrm2(nz, w->rx) + m * w->mu;
}

static void dense_save_best(dnws *w)
{
    memcpy(w->bx, w->x, sizeof(double) * w->nz);
    memcpy(w->bs, w->s, sizeof(double) * w->nineq);
    memcpy(w->bz, w->z, sizeof(double) * w->nineq);
    if (w->neq) memcpy(w->by, w->y, sizeof(double) * w->neq);
    memcpy(w->bK, w->K, sizeof(double) * w->N * w->N);
    w->bres = w->resid;
}

/*
 * DenseQPFunction forward (qp.py:219-237 + batch_LU.py:29-201).  Kbest (B,N,N), N =
 * nz+2 nineq+neq, receives the best iterate's K (what Solver.forward saves for backward).
 */
API int dqp_oracle_dense_forward(int B, int nz, int nineq, int neq,
                                 const double *Q, const double *p, const double *G,
                                 const double *h, const double *A, const double *b,
                                 double eps, int notImprovedLim, int maxIter,
                                 double *zhat, double *lam, double *nu, double *slack,
                                 double *Kbest, int *iters_out, int nthreads)
{
    nthreads = set_threads(nthreads);
    const double KKTeps = 1e-7;                                 /* batch_LU.py:42 */
    const int N = nz + 2 * nineq + neq;
    const size_t nd = dnws_doubles(nz, nineq, neq);
    double *buf = (double *)malloc(sizeof(double) * nd * B);
    int *ibuf = (int *)malloc(sizeof(int) * (size_t)(N + 8) * B);
    dnws *W = (dnws *)malloc(sizeof(dnws) * B);
    if (!buf || !ibuf || !W) { free(buf); free(ibuf); free(W); return -1; }
    int fail = 0;
#pragma omp parallel for num_threads(nthreads) schedule(static) reduction(| : fail)
    for (int i = 0; i < B; ++i) {
        dnws *w = &W[i];
        double *q = buf + nd * i;
        w->nz = nz; w->nineq = nineq; w->neq = neq; w->N = N;
        w->K = q; q += (size_t)N * N; w->Kt = q; q += (size_t)N * N;
        w->Klu = q; q += (size_t)N * N; w->bK = q; q += (size_t)N * N;
        w->x = q; q += nz; w->s = q; q += nineq; w->z = q; q += nineq; w->y = q; q += neq;
        w->bx = q; q += nz; w->bs = q; q += nineq; w->bz = q; q += nineq; w->by = q; q += neq;
        w->r = q; q += N; w->l = q; q += N; w->res = q; q += N; w->l2 = q; q += N;
        w->rx = q; q += nz; w->rs = q; q += nineq; w->rz = q; q += nineq; w->ry = q; q += neq;
        w->da = q; q += N; w->dc = q; q += N; w->az = q; q += nineq; w->as = q; q += nineq;
        w->piv = ibuf + (size_t)(N + 8) * i;
        w->has_best = 0;
        w->Q = Q + (size_t)i * nz * nz; w->p = p + (size_t)i * nz;
        w->G = G + (size_t)i * nineq * nz; w->h = h + (size_t)i * nineq;
        w->A = neq ? A + (size_t)i * neq * nz : NULL; w->b = neq ? b + (size_t)i * neq : NULL;
        dense_build_K(w);                                       /* Z = S = 1 */
        memcpy(w->Kt, w->K, sizeof(double) * N * N);
        for (int k = 0; k < N; ++k) w->Kt[k * N + k] += (k < nz + nineq) ? KKTeps : -KKTeps;
        for (int k = 0; k < nineq; ++k) { w->rs[k] = 0.0; w->rz[k] = -w->h[k]; }
        for (int k = 0; k < neq; ++k) w->ry[k] = -w->b[k];
        if (dense_solve_kkt(w, w->K, w->Kt, w->p, w->rs, w->rz, w->ry, w->da)) { fail |= 1; continue; }
        memcpy(w->x, w->da, sizeof(double) * nz);
        memcpy(w->s, w->da + nz, sizeof(double) * nineq);
        memcpy(w->z, w->da + nz + nineq, sizeof(double) * nineq);
        memcpy(w->y, w->da + nz + 2 * nineq, sizeof(double) * neq);
        shift_ge_one(nineq, w->s);
        shift_ge_one(nineq, w->z);
    }
    int iters = 0;
    if (!fail) {
        int nNotImproved = 0;
        for (int it = 0; it < maxIter; ++it) {
            int improved = 0, first = !W[0].has_best;
            double best_max = -INFINITY, mu_min = INFINITY;
#pragma omp parallel for num_threads(nthreads) schedule(static)
            for (int i = 0; i < B; ++i) {
                dnws *w = &W[i];
                dense_residuals(w);
                dense_set_diag(w, w->K, w->z, w->s, 0.0);       /* batch_LU.py:110-113 */
                dense_set_diag(w, w->Kt, w->z, w->s, KKTeps);
            }
            iters = it + 1;
            for (int i = 0; i < B; ++i) {
                dnws *w = &W[i];
                if (!w->has_best) { dense_save_best(w); w->has_best = 1; }
                else if (w->resid < w->bres) { dense_save_best(w); improved = 1; }
                best_max = tmax(best_max, w->bres);
                mu_min = tmin(mu_min, w->mu);
            }
            if (first || improved) nNotImproved = 0; else nNotImproved += 1;
            if (nNotImproved == notImprovedLim || best_max < eps || mu_min > 1e32) break;

            int bad = 0;
#pragma omp parallel for num_threads(nthreads) schedule(static) reduction(| : bad)
            for (int i = 0; i < B; ++i) {
                dnws *w = &W[i];
                bad |= dense_solve_kkt(w, w->K, w->Kt, w->rx, w->rs, w->rz, w->ry, w->da) != 0;
                w->amax_z = step_ratio(nineq, w->z, w->da + nz + nineq, w->az, 1);
                w->amax_s = step_ratio(nineq, w->s, w->da + nz, w->as, 1);
            }
            if (bad) break;
            double amz = -INFINITY, ams = -INFINITY;
            for (int i = 0; i < B; ++i) { amz = tmax(amz, W[i].amax_z); ams = tmax(ams, W[i].amax_s); }
#pragma omp parallel for num_threads(nthreads) schedule(static) reduction(| : bad)
            for (int i = 0; i < B; ++i) {
                dnws *w = &W[i];
                const double *dsa = w->da + nz, *dza = w->da + nz + nineq;
                double alpha = tmin(tmin(step_from(nineq, w->az, dza, amz),
                                         step_from(nineq, w->as, dsa, ams)), 1.0);
                double t3 = 0.0, t4 = 0.0;
                for (int k = 0; k < nineq; ++k) {
                    t3 += (w->s[k] + alpha * dsa[k]) * (w->z[k] + alpha * dza[k]);
                    t4 += w->s[k] * w->z[k];
                }
                double sig = t3 / t4; sig = sig * sig * sig;
                for (int k = 0; k < nz; ++k) w->rx[k] = 0.0;
                for (int k = 0; k < nineq; ++k) {
                    w->rs[k] = -w->mu * sig + dsa[k] * dza[k];     /* batch_LU.py:170 (no /s) */
                    w->rz[k] = 0.0;
                }
                for (int k = 0; k < neq; ++k) w->ry[k] = 0.0;
                bad |= dense_solve_kkt(w, w->K, w->Kt, w->rx, w->rs, w->rz, w->ry, w->dc) != 0;
                for (int k = 0; k < N; ++k) w->da[k] += w->dc[k];
                w->amax_z = step_ratio(nineq, w->z, w->da + nz + nineq, w->az, 1);
                w->amax_s = step_ratio(nineq, w->s, w->da + nz, w->as, 1);
            }
            if (bad) break;
            amz = -INFINITY; ams = -INFINITY;
            for (int i = 0; i < B; ++i) { amz = tmax(amz, W[i].amax_z); ams = tmax(ams, W[i].amax_s); }
#pragma omp parallel for num_threads(nthreads) schedule(static)
            for (int i = 0; i < B; ++i) {
                dnws *w = &W[i];
                const double *dsa = w->da + nz, *dza = w->da + nz + nineq;
                double alpha = tmin(0.999 * tmin(step_from(nineq, w->az, dza, amz),
                                                 step_from(nineq, w->as, dsa, ams)), 1.0);
                for (int k = 0; k < nz; ++k) w->x[k] += alpha * w->da[k];
                for (int k = 0; k < nineq; ++k) { w->s[k] += alpha * dsa[k]; w->z[k] += alpha * dza[k]; }
                for (int k = 0; k < neq; ++k) w->y[k] += alpha * w->da[nz + 2 * nineq + k];
            }
        }
        for (int i = 0; i < B; ++i) {
            dnws *w = &W[i];
            if (!w->has_best) { dense_residuals(w); dense_save_best(w); }
            memcpy(zhat + (size_t)i * nz, w->bx, sizeof(double) * nz);
            memcpy(lam + (size_t)i * nineq, w->bz, sizeof(double) * nineq);
            memcpy(slack + (size_t)i * nineq, w->bs, sizeof(double) * nineq);
            if (neq) memcpy(nu + (size_t)i * neq, w->by, sizeof(double) * neq);
            if (Kbest) memcpy(Kbest + (size_t)i * N * N, w->bK, sizeof(double) * N * N);
        }
    }
    if (iters_out) iters_out[0] = iters;
    free(buf); free(ibuf); free(W);
    return fail ? 1 : 0;
}

/* DenseQPFunction backward (qp.py:239-270): solve_kkt(K, K, dl_dzhat, 0, 0, 0) with the
 * saved best K (unregularised LU + one refinement), then the gradient formulas. */
API int dqp_oracle_dense_backward(int B, int nz, int nineq, int neq, const double *Kbest,
                                  const double *zhat, const double *lam, const double *nu,
                                  const double *dl_dzhat,
                                  double *dQ, double *dp, double *dG, double *dh,
                                  double *dA, double *db, int nthreads)
{
    nthreads = set_threads(nthreads);
    const int N = nz + 2 * nineq + neq;
    const size_t nd = (size_t)N * N + 8 * (size_t)N + 64;
    double *buf = (double *)malloc(sizeof(double) * nd * B);
    int *ibuf = (int *)malloc(sizeof(int) * (size_t)(N + 8) * B);
    if (!buf || !ibuf) { free(buf); free(ibuf); return -1; }
    int fail = 0;
#pragma omp parallel for num_threads(nthreads) schedule(static) reduction(| : fail)
    for (int i = 0; i < B; ++i) {
        dnws ws, *w = &ws;
        double *q = buf + nd * i;
        w->nz = nz; w->nineq = nineq; w->neq = neq; w->N = N;
        w->Klu = q; q += (size_t)N * N;
        w->r = q; q += N; w->l = q; q += N; w->res = q; q += N; w->da = q; q += N;
        w->rs = q; q += nineq; w->rz = q; q += nineq; w->ry = q; q += neq + 1;
        w->piv = ibuf + (size_t)(N + 8) * i;
        for (int k = 0; k < nineq; ++k) { w->rs[k] = 0.0; w->rz[k] = 0.0; }
        for (int k = 0; k < neq; ++k) w->ry[k] = 0.0;
        const double *K = Kbest + (size_t)i * N * N;
        if (dense_solve_kkt(w, K, K, dl_dzhat + (size_t)i * nz, w->rs, w->rz, w->ry, w->da)) {
            fail |= 1; continue;
        }
        kkt_grads(nz, nineq, neq, zhat + (size_t)i * nz, lam + (size_t)i * nineq,
                  neq ? nu + (size_t)i * neq : NULL,
                  w->da, w->da + nz + nineq, w->da + nz + 2 * nineq,
                  dQ + (size_t)i * nz * nz, dp + (size_t)i * nz, dG + (size_t)i * nineq * nz,
                  dh + (size_t)i * nineq, neq ? dA + (size_t)i * neq * nz : NULL,
                  neq ? db + (size_t)i * neq : NULL);
    }
    free(buf); free(ibuf);
    return fail ? 1 : 0;
}
