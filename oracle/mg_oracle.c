/*
 * mg_oracle.c -- TEST INFRASTRUCTURE ONLY (CPU oracle).
 *
 * Plain-C restatement of the 2D Poisson V-cycle of RoyJames/MultiGrid, used as
 * the checker for the HIP path.  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may load this file's shared object; the product
 * library never links or calls it.
 *
 * Every function cites the reference lines it follows (paths relative to the
 * reference checkout).  Arrays are column-major like DTDoubleArray
 * (DTSource/DTDoubleArray.h:38-49): A(i,j) = a[i + j*m], i fastest.
 *
 * Build WITHOUT fast-math and WITHOUT FMA contraction
 * (-O2 -ffp-contract=off): the left-to-right expression trees below are the
 * definition of a correct result, bit for bit.
 *
 * Parity pin: the reference's own solver functions (main.cpp:147-357, all of
 * main.cpp except the boost include at :21-23 and main() at :360-459) are
 * compiled from the reference checkout into oracle/_ref/librefops.so
 * (oracle/Makefile, oracle/ref_main_harness.cpp; -O2 -DNDEBUG
 * -ffp-contract=off).  tests/test_oracle.py compares every function below with
 * it bit for bit on seeded inputs, and against that library's committed
 * outputs (tests/golden/reference_vectors.npz, reference_known_answers.json)
 * where the checkout is absent.  Maximum/Minimum/+= are additionally checked
 * against the reference's DTSource sources (oracle/_ref/libdtref.so).
 * Deviation: a hierarchy with a level of fewer than 3 points (coarsest = 1
 * ends in a 2x2 grid with no unknowns, which the reference walks through) is
 * refused here, as it is by the product.
 *
 * The same file builds a second library, libmg_oracle_omp.so (-fopenmp): the
 * loops over columns carry OpenMP pragmas.  Every grid point is computed by
 * the same expression from the same inputs whatever thread runs it (Jacobi
 * reads only the previous sweep), so the threaded build returns the same bits;
 * it exists to time the algorithm on all host cores, the reference itself is
 * single-threaded.
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/time.h>

#define IDX(i, j, m) ((size_t)(i) + (size_t)(j) * (size_t)(m))

#ifdef _OPENMP
#include <omp.h>
int mgo_threads(void) { return omp_get_max_threads(); }
#else
int mgo_threads(void) { return 1; }
#endif

/* memcpy / memset of a whole grid, split over the threads of the OpenMP build */
static void grid_copy(double *dst, const double *src, size_t len)
{
    const size_t blk = (size_t)1 << 16;
    const long nblk = (long)((len + blk - 1) / blk);
#pragma omp parallel for schedule(static)
    for (long b = 0; b < nblk; b++) {
        size_t k0 = (size_t)b * blk, n = len - k0 < blk ? len - k0 : blk;
        memcpy(dst + k0, src + k0, n * sizeof(double));
    }
}

static void grid_zero(double *dst, size_t len)
{
    const size_t blk = (size_t)1 << 16;
    const long nblk = (long)((len + blk - 1) / blk);
#pragma omp parallel for schedule(static)
    for (long b = 0; b < nblk; b++) {
        size_t k0 = (size_t)b * blk, n = len - k0 < blk ? len - k0 : blk;
        memset(dst + k0, 0, n * sizeof(double));
    }
}

/* DTSource/DTDoubleArray.cpp:883-899 (Maximum) and :923-939 (Minimum):
 * start at -inf / +inf, `maxV < v ? v : maxV`, so NaNs never win.
 * main.cpp:147-150: calcNorm = max(Maximum(r), -Minimum(r)). */
double mgo_maximum(const double *a, size_t len)
{
    double maxv = -INFINITY;
#ifdef _OPENMP
    /* per-thread partial maxima over contiguous blocks, merged in block order with the
     * same comparison: the value is that of the flat loop (max is exact and associative) */
    enum { MAXT = 1024 };
    double part[MAXT];
    int nt = omp_get_max_threads();
    if (nt > MAXT) nt = MAXT;
    for (int t = 0; t < nt; t++) part[t] = -INFINITY;
#pragma omp parallel num_threads(nt)
    {
        int t = omp_get_thread_num(), n = omp_get_num_threads();
        size_t k0 = len * (size_t)t / (size_t)n, k1 = len * (size_t)(t + 1) / (size_t)n;
        double mv = -INFINITY;
        for (size_t k = k0; k < k1; k++) {
            double v = a[k];
            mv = (mv < v ? v : mv);
        }
        part[t] = mv;
    }
    for (int t = 0; t < nt; t++) maxv = (maxv < part[t] ? part[t] : maxv);
#else
    for (size_t k = 0; k < len; k++) {
        double v = a[k];
        maxv = (maxv < v ? v : maxv);
    }
#endif
    return maxv;
}

double mgo_minimum(const double *a, size_t len)
{
    double minv = INFINITY;
#ifdef _OPENMP
    enum { MAXT = 1024 };
    double part[MAXT];
    int nt = omp_get_max_threads();
    if (nt > MAXT) nt = MAXT;
    for (int t = 0; t < nt; t++) part[t] = INFINITY;
#pragma omp parallel num_threads(nt)
    {
        int t = omp_get_thread_num(), n = omp_get_num_threads();
        size_t k0 = len * (size_t)t / (size_t)n, k1 = len * (size_t)(t + 1) / (size_t)n;
        double mv = INFINITY;
        for (size_t k = k0; k < k1; k++) {
            double v = a[k];
            mv = (mv > v ? v : mv);
        }
        part[t] = mv;
    }
    for (int t = 0; t < nt; t++) minv = (minv > part[t] ? part[t] : minv);
#else
    for (size_t k = 0; k < len; k++) {
        double v = a[k];
        minv = (minv > v ? v : minv);
    }
#endif
    return minv;
}

double mgo_calc_norm(const double *r, size_t len)
{
    double mx = mgo_maximum(r, len);
    double mn = -mgo_minimum(r, len);
    /* std::max<double>(a,b) = (a < b) ? b : a */
    return (mx < mn) ? mn : mx;
}

/* main.cpp:171-199 relax(): Niter damped-Jacobi sweeps, each reading a full
 * pre-sweep copy (:186) and writing the interior in place (:194-195). */
void mgo_relax(double *u, const double *f, int m, double dx, int niter, double omega)
{
    double nomega = 1 - omega;
    double h2 = dx * dx;
    double factor = 0.25;
    size_t len = (size_t)m * (size_t)m;
    double *old = (double *)malloc(len * sizeof(double));
    for (int it = 0; it < niter; it++) {
        grid_copy(old, u, len);
#pragma omp parallel for schedule(static)
        for (int j = 1; j < m - 1; j++) {
            for (int i = 1; i < m - 1; i++) {
                double s = old[IDX(i - 1, j, m)] + old[IDX(i + 1, j, m)];
                s = s + old[IDX(i, j - 1, m)];
                s = s + old[IDX(i, j + 1, m)];
                double fh = f[IDX(i, j, m)] * h2;
                s = s - fh;
                s = s * factor;
                s = s * omega;
                double a = old[IDX(i, j, m)] * nomega;
                u[IDX(i, j, m)] = a + s;
            }
        }
    }
    free(old);
}

/* main.cpp:255-281 residual(): r = f - (uW+uE+uS+uN - 4 u) * (1/h2) on the
 * interior, 0 elsewhere (:267). */
void mgo_residual(const double *u, const double *f, double *r, int m, double dx)
{
    double h2 = dx * dx;
    double invh2 = 1.0 / h2;
    grid_zero(r, (size_t)m * (size_t)m);
#pragma omp parallel for schedule(static)
    for (int j = 1; j < m - 1; j++) {
        for (int i = 1; i < m - 1; i++) {
            double s = u[IDX(i - 1, j, m)] + u[IDX(i + 1, j, m)];
            s = s + u[IDX(i, j - 1, m)];
            s = s + u[IDX(i, j + 1, m)];
            double c4 = u[IDX(i, j, m)] * 4.0;
            s = s - c4;
            s = s * invh2;
            r[IDX(i, j, m)] = f[IDX(i, j, m)] - s;
        }
    }
}

/* main.cpp:201-221 coarsen(): 9-point weighted restriction.  The corner term
 * at :218 adds fine(2i+1,2j-1) twice and never fine(2i-1,2j+1); reproduced
 * verbatim.  coarse is zeroed first (:211) so its boundary is 0. */
void mgo_coarsen(const double *fine, int mf, double *coarse, int mc)
{
    double selfw = 1.0 / 4.0;
    double neighborw = 1.0 / 8.0;
    double cornerw = 1.0 / 16.0;
    grid_zero(coarse, (size_t)mc * (size_t)mc);
#pragma omp parallel for schedule(static)
    for (int i = 1; i < mc - 1; i++) {
        for (int j = 1; j < mc - 1; j++) {
            double a = fine[IDX(i * 2, j * 2, mf)] * selfw;
            double nb = fine[IDX(i * 2 - 1, j * 2, mf)] + fine[IDX(i * 2 + 1, j * 2, mf)];
            nb = nb + fine[IDX(i * 2, j * 2 - 1, mf)];
            nb = nb + fine[IDX(i * 2, j * 2 + 1, mf)];
            nb = nb * neighborw;
            double cn = fine[IDX(i * 2 - 1, j * 2 - 1, mf)] + fine[IDX(i * 2 + 1, j * 2 - 1, mf)];
            cn = cn + fine[IDX(i * 2 + 1, j * 2 - 1, mf)];
            cn = cn + fine[IDX(i * 2 + 1, j * 2 + 1, mf)];
            cn = cn * cornerw;
            coarse[IDX(i, j, mc)] = (a + nb) + cn;
        }
    }
}

/* main.cpp:223-253 refine(): bilinear interpolation into a zeroed array. */
void mgo_refine(const double *coarse, int mc, double *fine, int mf)
{
    grid_zero(fine, (size_t)mf * (size_t)mf);
#pragma omp parallel for schedule(static)
    for (int j = 1; j < mf - 1; j++) {
        for (int i = 1; i < mf - 1; i++) {
            int I = i / 2, J = j / 2;
            double v;
            if (i % 2 == 0 && j % 2 == 0) {
                v = coarse[IDX(I, J, mc)];
            } else if (i % 2 == 0 && j % 2 == 1) {
                v = 0.5 * (coarse[IDX(I, J, mc)] + coarse[IDX(I, J + 1, mc)]);
            } else if (i % 2 == 1 && j % 2 == 0) {
                v = 0.5 * (coarse[IDX(I, J, mc)] + coarse[IDX(I + 1, J, mc)]);
            } else {
                double s = coarse[IDX(I, J, mc)] + coarse[IDX(I + 1, J, mc)];
                s = s + coarse[IDX(I, J + 1, mc)];
                s = s + coarse[IDX(I + 1, J + 1, mc)];
                v = 0.25 * s;
            }
            fine[IDX(i, j, mf)] = v;
        }
    }
}

/* DTSource/DTArrayTemplates.h:157-167 (operator+= on two arrays): flat loop
 * over every entry, used at main.cpp:341. */
void mgo_plus_equals(double *a, const double *b, size_t len)
{
#pragma omp parallel for schedule(static)
    for (long k = 0; k < (long)len; k++) a[k] = a[k] + b[k];
}

/* main.cpp:153-169 direct_solve(): closed form for the single unknown of a
 * 3x3 grid; any other size prints the reference's message and changes nothing.
 * Returns 0 if solved, 1 if the size is unsupported. */
int mgo_direct_solve(double *u, const double *f, int m, double dx, int quiet)
{
    double h2 = dx * dx;
    double factor = 0.25;
    if (m == 3) {
        double s = u[IDX(0, 1, 3)] + u[IDX(2, 1, 3)];
        s = s + u[IDX(1, 0, 3)];
        s = s + u[IDX(1, 2, 3)];
        s = s - f[IDX(1, 1, 3)] * h2;
        u[IDX(1, 1, 3)] = s * factor;
        return 0;
    }
    if (!quiet) printf("Error: Direct solver not defined for %dx%d matrices!\n", m, m);
    return 1;
}

/* EXTENSION (not reference behaviour): exact solve of the coarsest grid, any size.
 * The system is the one the reference's unused getSparseSol() assembles
 * (main.cpp:79-110): 4 u(i,j) - sum of the four neighbours = -h^2 f(i,j), boundary
 * values of u moved to the right-hand side; unknowns ordered i fastest.  Solved by
 * a banded Cholesky factorisation (half bandwidth n = m-2), the textbook
 * algorithm; pinned against Eigen::SimplicialCholesky, which the reference calls
 * at main.cpp:119-120, through oracle/_ref/libeigenref.so.  Returns 0, or 1 if
 * the matrix is not positive definite (cannot happen for this matrix). */
int mgo_coarse_exact(double *u, const double *f, int m, double dx)
{
    const int n = m - 2;
    if (n < 1) return 0;
    const size_t N = (size_t)n * (size_t)n;
    const size_t bw = (size_t)n + 1; /* L(r, c) for r-c = 0..n sits at band[r*bw + (r-c)] */
    double *band = (double *)calloc(N * bw, sizeof(double));
    double *x = (double *)malloc(N * sizeof(double));
    const double h2 = dx * dx;
    for (int j = 1; j <= n; j++) {
        for (int i = 1; i <= n; i++) {
            size_t row = (size_t)(i - 1) + (size_t)(j - 1) * (size_t)n;
            double b = -h2 * f[IDX(i, j, m)];
            band[row * bw + 0] = 4.0;
            if (i > 1) band[row * bw + 1] = -1.0; else b += u[IDX(i - 1, j, m)];
            if (i == n) b += u[IDX(i + 1, j, m)];
            if (j > 1) band[row * bw + (size_t)n] += -1.0; else b += u[IDX(i, j - 1, m)];
            if (j == n) b += u[IDX(i, j + 1, m)];
            x[row] = b;
        }
    }
    /* for n == 1 the two sub-diagonals coincide at distance 1 and neither exists */
    int rc = 0;
    for (size_t c = 0; c < N && !rc; c++) {
        size_t k0 = c > (size_t)n ? c - (size_t)n : 0;
        double d = band[c * bw + 0];
        for (size_t k = k0; k < c; k++) d -= band[c * bw + (c - k)] * band[c * bw + (c - k)];
        if (!(d > 0.0)) { rc = 1; break; }
        d = sqrt(d);
        band[c * bw + 0] = d;
        size_t r1 = c + (size_t)n < N - 1 ? c + (size_t)n : N - 1;
        for (size_t r = c + 1; r <= r1; r++) {
            size_t kk0 = r > (size_t)n ? r - (size_t)n : 0;
            if (kk0 < k0) kk0 = k0;
            double t = band[r * bw + (r - c)];
            for (size_t k = kk0; k < c; k++) t -= band[r * bw + (r - k)] * band[c * bw + (c - k)];
            band[r * bw + (r - c)] = t / d;
        }
    }
    if (!rc) {
        for (size_t r = 0; r < N; r++) { /* L y = b */
            size_t k0 = r > (size_t)n ? r - (size_t)n : 0;
            double t = x[r];
            for (size_t k = k0; k < r; k++) t -= band[r * bw + (r - k)] * x[k];
            x[r] = t / band[r * bw + 0];
        }
        for (size_t rr = N; rr-- > 0;) { /* L^T x = y */
            size_t r1 = rr + (size_t)n < N - 1 ? rr + (size_t)n : N - 1;
            double t = x[rr];
            for (size_t k = rr + 1; k <= r1; k++) t -= band[k * bw + (k - rr)] * x[k];
            x[rr] = t / band[rr * bw + 0];
        }
        for (int j = 1; j <= n; j++)
            for (int i = 1; i <= n; i++) u[IDX(i, j, m)] = x[(size_t)(i - 1) + (size_t)(j - 1) * (size_t)n];
    }
    free(band);
    free(x);
    return rc;
}

/* main.cpp:286: depth = int(log2(1.0f*(M-1)/coarsest) + 0.5f), float math. */
int mgo_depth(int m, int coarsest)
{
    float ratio = 1.0f * (float)(m - 1) / (float)coarsest;
    return (int)(log2f(ratio) + 0.5f);
}

static double now_s(void)
{
    /* DTSource/DTTimer.h:13-19 uses gettimeofday wall time. */
    struct timeval tv;
    gettimeofday(&tv, NULL);
    return (double)tv.tv_sec + 1e-6 * (double)tv.tv_usec;
}

/* main.cpp:283-357 MultiGrid(): level set-up (:286-300), Nv V(Ndown,Nup)
 * cycles (:309-354), resnorm(0..Nv) and cumulative smoother-only times(0..Nv).
 * f, v are the fine-level arrays (v in/out, boundary values already set).
 * The unused `before`/`lowest` residual passes (:311,:332) have no effect on
 * any output and are skipped.  Returns 0, or <0 for a hierarchy this
 * restatement refuses (depth<0 or an even-sized level above the coarsest, where
 * the reference indexes out of range in refine(); the coarsest level itself may be
 * even).  coarse_exact != 0 is the EXTENSION: a coarsest grid that is not 3x3 is
 * solved exactly (mgo_coarse_exact) instead of being left as it is (:167). */
int mgo_multigrid_opt(const double *f, double *v, int m, double dx, int Nv, int Ndown, int Nup,
                      double omega, int coarsest, int pure_jacobi, int coarse_exact, double *resnorm,
                      double *times);

int mgo_multigrid(const double *f, double *v, int m, double dx, int Nv, int Ndown, int Nup,
                  double omega, int coarsest, int pure_jacobi, double *resnorm, double *times)
{
    return mgo_multigrid_opt(f, v, m, dx, Nv, Ndown, Nup, omega, coarsest, pure_jacobi, 0, resnorm, times);
}

static int multigrid_core(const double *f, double *v, int m, double dx, int Nv, int Ndown, int Nup, double omega,
                          int depth, int pure_jacobi, int coarse_exact, double *resnorm, double *times);

int mgo_multigrid_opt(const double *f, double *v, int m, double dx, int Nv, int Ndown, int Nup,
                      double omega, int coarsest, int pure_jacobi, int coarse_exact, double *resnorm,
                      double *times)
{
    if (m < 3 || (m % 2) == 0 || coarsest < 1) return -1;
    int depth = mgo_depth(m, coarsest);
    if (depth < 0) return -2;
    return multigrid_core(f, v, m, dx, Nv, Ndown, Nup, omega, depth, pure_jacobi, coarse_exact, resnorm, times);
}

/* The body of MultiGrid() for a given number of coarsenings (main.cpp:288-357). */
static int multigrid_core(const double *f, double *v, int m, double dx, int Nv, int Ndown, int Nup, double omega,
                          int depth, int pure_jacobi, int coarse_exact, double *resnorm, double *times)
{
    int *dims = (int *)malloc((size_t)(depth + 1) * sizeof(int));
    double *hs = (double *)malloc((size_t)(depth + 1) * sizeof(double));
    dims[0] = m;
    hs[0] = dx;
    for (int d = 1; d <= depth; d++) {
        dims[d] = (dims[d - 1] - 1) / 2 + 1; /* :293 */
        hs[d] = hs[d - 1] * 2.0;             /* :297 */
    }
    for (int d = 0; d <= depth; d++) {
        if (dims[d] < 3 || ((dims[d] % 2) == 0 && d < depth)) {
            if (d > 0 || depth > 0) { free(dims); free(hs); return -3; }
        }
    }
    double **F = (double **)calloc((size_t)(depth + 1), sizeof(double *));
    double **V = (double **)calloc((size_t)(depth + 1), sizeof(double *));
    size_t n0 = (size_t)m * (size_t)m;
    F[0] = (double *)f; /* never written */
    V[0] = v;           /* shared storage, main.cpp:288 */
    for (int d = 1; d <= depth; d++) {
        size_t nd = (size_t)dims[d] * (size_t)dims[d];
        F[d] = (double *)calloc(nd, sizeof(double));
        V[d] = (double *)calloc(nd, sizeof(double));
    }
    double *res = (double *)malloc(n0 * sizeof(double));
    double *refined = (double *)malloc(n0 * sizeof(double));

    mgo_residual(V[0], F[0], res, m, dx);
    resnorm[0] = mgo_calc_norm(res, n0); /* :307 */
    times[0] = 0;                        /* :308 */
    for (int iter = 0; iter < Nv; iter++) {
        double t_single = 0;
        if (!pure_jacobi) {
            for (int d = 0; d < depth; d++) { /* :317-328 */
                double t0 = now_s();
                mgo_relax(V[d], F[d], dims[d], hs[d], Ndown, omega);
                t_single += now_s() - t0;
                mgo_residual(V[d], F[d], res, dims[d], hs[d]);
                mgo_coarsen(res, dims[d], F[d + 1], dims[d + 1]);
            }
            if (coarse_exact && dims[depth] != 3)
                mgo_coarse_exact(V[depth], F[depth], dims[depth], hs[depth]);
            else
                mgo_direct_solve(V[depth], F[depth], dims[depth], hs[depth], 0); /* :331 */
            for (int d = depth - 1; d >= 0; d--) { /* :335-346 */
                size_t nd = (size_t)dims[d] * (size_t)dims[d];
                mgo_refine(V[d + 1], dims[d + 1], refined, dims[d]);
                mgo_plus_equals(V[d], refined, nd);
                double t0 = now_s();
                mgo_relax(V[d], F[d], dims[d], hs[d], Nup, omega);
                t_single += now_s() - t0;
                grid_zero(V[d + 1], (size_t)dims[d + 1] * (size_t)dims[d + 1]); /* :345 */
            }
        } else {
            mgo_relax(V[0], F[0], m, dx, 1, omega); /* :348 */
        }
        times[iter + 1] = times[iter] + t_single; /* :350 */
        mgo_residual(V[0], F[0], res, m, dx);
        resnorm[iter + 1] = mgo_calc_norm(res, n0); /* :351-353 */
    }
    for (int d = 1; d <= depth; d++) { free(F[d]); free(V[d]); }
    free(F); free(V); free(res); free(refined); free(dims); free(hs);
    return 0;
}

/* EXTENSION (not in the reference): full multigrid (nested iteration) composed of the
 * operators above.  The right-hand side is restricted to every level with coarsen(), the
 * Dirichlet values of v are injected (coarse boundary point k <- fine boundary point 2k),
 * the coarsest grid is solved (direct_solve, or mgo_coarse_exact with coarse_exact), and
 * on the way up each level starts from refine() of the level below and runs ncyc
 * V(Ndown,Nup) cycles of MultiGrid() on its own sub-hierarchy.  The interior of v on
 * entry is ignored.  resnorm[0..ncyc] is the residual history of the finest level's
 * cycles.  Same refusals as mgo_multigrid. */
int mgo_fmg(const double *f, double *v, int m, double dx, int ncyc, int Ndown, int Nup, double omega,
            int coarsest, int coarse_exact, double *resnorm)
{
    if (m < 3 || (m % 2) == 0 || coarsest < 1) return -1;
    int depth = mgo_depth(m, coarsest);
    if (depth < 0) return -2;
    int *dims = (int *)malloc((size_t)(depth + 1) * sizeof(int));
    double *hs = (double *)malloc((size_t)(depth + 1) * sizeof(double));
    dims[0] = m;
    hs[0] = dx;
    for (int d = 1; d <= depth; d++) {
        dims[d] = (dims[d - 1] - 1) / 2 + 1;
        hs[d] = hs[d - 1] * 2.0;
    }
    for (int d = 1; d <= depth; d++)
        if (dims[d] < 3 || ((dims[d] % 2) == 0 && d < depth)) { free(dims); free(hs); return -3; }
    double **F = (double **)calloc((size_t)(depth + 1), sizeof(double *));
    double **V = (double **)calloc((size_t)(depth + 1), sizeof(double *));
    F[0] = (double *)f;
    V[0] = v;
    for (int d = 1; d <= depth; d++) {
        int mc = dims[d], mf = dims[d - 1];
        F[d] = (double *)calloc((size_t)mc * mc, sizeof(double));
        V[d] = (double *)calloc((size_t)mc * mc, sizeof(double));
        mgo_coarsen(F[d - 1], mf, F[d], mc);
        for (int k = 0; k < mc; k++) {
            V[d][IDX(0, k, mc)] = V[d - 1][IDX(0, 2 * k, mf)];
            V[d][IDX(mc - 1, k, mc)] = V[d - 1][IDX(mf - 1, 2 * k, mf)];
            V[d][IDX(k, 0, mc)] = V[d - 1][IDX(2 * k, 0, mf)];
            V[d][IDX(k, mc - 1, mc)] = V[d - 1][IDX(2 * k, mf - 1, mf)];
        }
    }
    double *rn = (double *)malloc((size_t)(ncyc + 1) * sizeof(double));
    double *tm = (double *)malloc((size_t)(ncyc + 1) * sizeof(double));
    double *refined = (double *)malloc((size_t)m * (size_t)m * sizeof(double));
    if (depth == 0) {
        /* a single level: MultiGrid() on it is the coarsest solve, ncyc times */
        multigrid_core(F[0], V[0], m, dx, ncyc, Ndown, Nup, omega, 0, 0, coarse_exact, rn, tm);
    } else {
        if (coarse_exact && dims[depth] != 3)
            mgo_coarse_exact(V[depth], F[depth], dims[depth], hs[depth]);
        else
            mgo_direct_solve(V[depth], F[depth], dims[depth], hs[depth], 1);
        for (int d = depth - 1; d >= 0; d--) {
            int mf = dims[d], mc = dims[d + 1];
            mgo_refine(V[d + 1], mc, refined, mf);
            for (int j = 1; j < mf - 1; j++)
                for (int i = 1; i < mf - 1; i++) V[d][IDX(i, j, mf)] = refined[IDX(i, j, mf)];
            multigrid_core(F[d], V[d], mf, hs[d], ncyc, Ndown, Nup, omega, depth - d, 0, coarse_exact, rn, tm);
        }
    }
    for (int k = 0; k <= ncyc; k++) resnorm[k] = rn[k];
    for (int d = 1; d <= depth; d++) { free(F[d]); free(V[d]); }
    free(F); free(V); free(rn); free(tm); free(refined); free(dims); free(hs);
    return 0;
}
