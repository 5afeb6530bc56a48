/*
 * solvers_oracle.c — CPU restatement of the RANSAC entry points of the sfm.py hot path:
 *
 *   cv2.findEssentialMat(pts0, pts1, K, RANSAC, 0.999, 0.4)      sfm.py:307
 *   cv2.recoverPose(E, pts0, pts1, K)                            sfm.py:311
 *   cv2.solvePnPRansac(X, p, K, d, <stray arg>)                  sfm.py:67
 *
 * TEST INFRASTRUCTURE ONLY (see sfm_oracle.h).  PARITY UNPINNED: the arithmetic restated here lives
 * in OpenCV (modules calib3d / core; un-vendored and version-unpinned in the reference; cv2 is not
 * installable in the build container), so every function follows OpenCV's published algorithm as
 * the author knows it — file and routine named at each function — and is pinned by known-answer
 * tests with planted ground truth (tests/test_oracle_solvers.py), not by OpenCV output.
 *
 * Everything here is plain, SEQUENTIAL C: one RANSAC iteration at a time, one model at a time,
 * one point at a time, exactly in the order RANSACPointSetRegistrator::run visits them.  It
 * shares no code with the product (sfm_mvs_amd/), which generates hypotheses in chunks and
 * scores them in batches on the device; the GPU tests hold the two to identical masks.
 *
 * Where OpenCV releases / builds differ, the choice made is stated at the spot.
 */
#include "sfm_oracle.h"
#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

void orc_jacobi_core(double* At, int m, int n, double* W, double* Vt); /* sfm_oracle.c */

/* ------------------------------------------------------------------------------------------
 * cv::RNG (core/operations.hpp): multiply-with-carry; RANSACPointSetRegistrator::run seeds it
 * with (uint64)-1, cv::SVD's null-space completion with 0x12345678.
 * ---------------------------------------------------------------------------------------- */
uint32_t orc_rng_next(uint64_t* state) {
    *state = (uint64_t)(uint32_t)*state * 4164903690u + (uint32_t)(*state >> 32);
    return (uint32_t)*state;
}

int orc_rng_uniform(uint64_t* state, int a, int b) { return a == b ? a : (int)(orc_rng_next(state) % (uint32_t)(b - a) + (uint32_t)a); }

/* RANSACUpdateNumIters (calib3d/ptsetreg.cpp). */
int orc_ransac_update_num_iters(double p, double ep, int model_points, int max_iters) {
    p = p < 0. ? 0. : p > 1. ? 1. : p;
    ep = ep < 0. ? 0. : ep > 1. ? 1. : ep;
    double num = 1. - p > DBL_MIN ? 1. - p : DBL_MIN;
    double denom = 1. - pow(1. - ep, model_points);
    if (denom < DBL_MIN) return 0;
    num = log(num);
    denom = log(denom);
    return denom >= 0 || -num >= max_iters * (-denom) ? max_iters : (int)lrint(num / denom); /* cvRound: nearest-even */
}

/* ------------------------------------------------------------------------------------------
 * cv::SVD::compute(A, w, u, vt[, FULL_UV]) (core/lapack.cpp: _SVDcompute + JacobiSVDImpl_).
 * A is m x n row-major.  The Jacobi sweeps run on the rows of `a` = A^T (m >= n) or A itself
 * (m < n, the roles of U and V swap afterwards).  With FULL_UV, or for singular values
 * <= DBL_MIN, the missing left vectors are completed the way OpenCV does it: a +-1/m vector
 * drawn from RNG(0x12345678) (bit 8 of each draw), two Gram-Schmidt passes against all the
 * previous rows with an L1 renormalisation after every projection, then an L2 normalisation.
 * Outputs: w[min(m,n)] descending; U is m x uc, Vt is vr x n with uc = vr = min(m,n), or
 * uc = m, vr = n when full_uv.  U / Vt may be NULL.
 * ---------------------------------------------------------------------------------------- */
#define SVD_MAX 12
void orc_svd(const double* A, int m, int n, int full_uv, double* w, double* U, double* Vt) {
    const int at = m < n;
    const int mm = at ? n : m, nn = at ? m : n; /* mm >= nn */
    if (mm > 2 * SVD_MAX || nn > SVD_MAX) return;
    const int urows = full_uv ? mm : nn;
    double a[2 * SVD_MAX * 2 * SVD_MAX], v[SVD_MAX * SVD_MAX], W[SVD_MAX];
    memset(a, 0, sizeof(a));
    for (int i = 0; i < nn; ++i)
        for (int k = 0; k < mm; ++k) a[i * mm + k] = at ? A[i * n + k] : A[k * n + i];
    orc_jacobi_core(a, mm, nn, W, v);
    /* left vectors: rows of `a` divided by their norm; zero / missing ones completed */
    {
        const double minval = DBL_MIN, eps = DBL_EPSILON * 10;
        uint64_t rng = 0x12345678u;
        for (int i = 0; i < urows; ++i) {
            double sd = i < nn ? W[i] : 0;
            for (int ii = 0; ii < 100 && sd <= minval; ++ii) {
                const double val0 = 1. / mm;
                for (int k = 0; k < mm; ++k) a[i * mm + k] = (orc_rng_next(&rng) & 256) != 0 ? val0 : -val0;
                for (int iter = 0; iter < 2; ++iter)
                    for (int j = 0; j < i; ++j) {
                        sd = 0;
                        for (int k = 0; k < mm; ++k) sd += a[i * mm + k] * a[j * mm + k];
                        double asum = 0;
                        for (int k = 0; k < mm; ++k) {
                            const double t = a[i * mm + k] - sd * a[j * mm + k];
                            a[i * mm + k] = t;
                            asum += fabs(t);
                        }
                        asum = asum > eps * 100 ? 1 / asum : 0;
                        for (int k = 0; k < mm; ++k) a[i * mm + k] *= asum;
                    }
                sd = 0;
                for (int k = 0; k < mm; ++k) sd += a[i * mm + k] * a[i * mm + k];
                sd = sqrt(sd);
            }
            const double s = sd > minval ? 1 / sd : 0.;
            for (int k = 0; k < mm; ++k) a[i * mm + k] *= s;
        }
    }
    for (int i = 0; i < nn; ++i) w[i] = W[i];
    if (!at) { /* u = (rows of a)^T : m x urows;  vt = v : n x n */
        if (U)
            for (int k = 0; k < mm; ++k)
                for (int i = 0; i < urows; ++i) U[k * urows + i] = a[i * mm + k];
        if (Vt) memcpy(Vt, v, sizeof(double) * (size_t)nn * nn);
    } else { /* u = v^T : m x m;  vt = rows of a : urows x n */
        if (U)
            for (int k = 0; k < nn; ++k)
                for (int i = 0; i < nn; ++i) U[k * nn + i] = v[i * nn + k];
        if (Vt) memcpy(Vt, a, sizeof(double) * (size_t)urows * mm);
    }
}

/* x = V diag(1/w) U^T b with singular values <= 2*DBL_EPSILON*sum(w) dropped
 * (core/lapack.cpp SVBkSbImpl_, nb = 1) — what cv::solve(..., DECOMP_SVD) runs after JacobiSVD. */
static void svd_solve(const double* A, int m, int n, const double* b, double* x) {
    double w[SVD_MAX], U[2 * SVD_MAX * SVD_MAX], Vt[SVD_MAX * SVD_MAX];
    orc_svd(A, m, n, 0, w, U, Vt);
    double threshold = 0;
    for (int i = 0; i < n; ++i) threshold += w[i];
    threshold *= DBL_EPSILON * 2;
    for (int j = 0; j < n; ++j) x[j] = 0;
    for (int i = 0; i < n; ++i) {
        double wi = w[i];
        if (fabs(wi) <= threshold) continue;
        wi = 1 / wi;
        double s = 0;
        for (int j = 0; j < m; ++j) s += U[j * n + i] * b[j];
        s *= wi;
        for (int j = 0; j < n; ++j) x[j] = x[j] + s * Vt[i * n + j];
    }
}

/* cv::invert(A, DECOMP_SVD) of an n x n matrix: sum_i (1/w_i) v_i u_i^T in singular-value order. */
static void svd_invert(const double* A, int n, double* inv) {
    double w[SVD_MAX], U[SVD_MAX * SVD_MAX], Vt[SVD_MAX * SVD_MAX], buf[SVD_MAX];
    orc_svd(A, n, n, 0, w, U, Vt);
    double threshold = 0;
    for (int i = 0; i < n; ++i) threshold += w[i];
    threshold *= DBL_EPSILON * 2;
    for (int j = 0; j < n * n; ++j) inv[j] = 0;
    for (int i = 0; i < n; ++i) {
        double wi = w[i];
        if (fabs(wi) <= threshold) continue;
        wi = 1 / wi;
        for (int j = 0; j < n; ++j) buf[j] = U[j * n + i] * wi;
        for (int r = 0; r < n; ++r) {
            const double s = Vt[i * n + r];
            for (int j = 0; j < n; ++j) inv[r * n + j] += s * buf[j];
        }
    }
}

static double det3(const double* M) {
    return M[0] * (M[4] * M[8] - M[5] * M[7]) - M[1] * (M[3] * M[8] - M[5] * M[6]) + M[2] * (M[3] * M[7] - M[4] * M[6]);
}

/* ------------------------------------------------------------------------------------------
 * cv::solvePoly (core/mathfuncs.cpp): Durand-Kerner on complex roots started at powers of
 * (1 + i), at most 300 sweeps, stopping only when no root moved at all.  coeffs[k] multiplies
 * x^k.  Returns the degree actually solved (leading coefficients <= DBL_EPSILON are dropped).
 * (OpenCV >= 4.2 adds a multiple-root correction; the simple iteration of 3.x is restated.)
 * ---------------------------------------------------------------------------------------- */
typedef struct { double re, im; } cplx;
static cplx cmul(cplx a, cplx b) { cplx r = {a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re}; return r; }
static cplx csub(cplx a, cplx b) { cplx r = {a.re - b.re, a.im - b.im}; return r; }
static cplx cadd(cplx a, cplx b) { cplx r = {a.re + b.re, a.im + b.im}; return r; }
static cplx cdiv(cplx a, cplx b) {
    const double t = 1. / (b.re * b.re + b.im * b.im);
    cplx r = {(a.re * b.re + a.im * b.im) * t, (-a.re * b.im + a.im * b.re) * t};
    return r;
}

int orc_solve_poly(const double* coeffs0, int deg, double* roots_re, double* roots_im, int max_iters) {
    cplx coeffs[32], roots[32];
    int n = deg;
    if (deg >= 32) return -1;
    for (int i = 0; i <= n; ++i) { coeffs[i].re = coeffs0[i]; coeffs[i].im = 0; }
    for (; n > 1; --n)
        if (fabs(coeffs[n].re) + fabs(coeffs[n].im) > DBL_EPSILON) break;
    cplx p = {1, 0}, r = {1, 1};
    for (int i = 0; i < n; ++i) {
        roots[i] = p;
        p = cmul(p, r);
    }
    max_iters = max_iters <= 0 ? 1000 : max_iters;
    for (int iter = 0; iter < max_iters; ++iter) {
        double max_diff = 0;
        for (int i = 0; i < n; ++i) {
            p = roots[i];
            cplx num = coeffs[n], denom = coeffs[n];
            for (int j = 0; j < n; ++j) {
                num = cadd(cmul(num, p), coeffs[n - j - 1]);
                if (j != i) denom = cmul(denom, csub(p, roots[j]));
            }
            num = cdiv(num, denom);
            roots[i] = csub(p, num);
            const double a = sqrt(num.re * num.re + num.im * num.im);
            if (a > max_diff) max_diff = a;
        }
        if (max_diff <= 0) break;
    }
    for (int i = 0; i < n; ++i) {
        roots_re[i] = roots[i].re;
        roots_im[i] = fabs(roots[i].im) < 1e-100 ? 0 : roots[i].im;
    }
    return n;
}

/* ------------------------------------------------------------------------------------------
 * EMEstimatorCallback::runKernel (calib3d/five-point.cpp) — Nister's five-point solver in the
 * Li-Hartley hidden-variable form:
 *   Q (5 x 9) rows [x1 x2, y1 x2, x2, x1 y2, y1 y2, y2, x1, y1, 1]  (x1 = first image)
 *   full SVD; rows 5..8 of Vt span the null space: E = x E0 + y E1 + z E2 + E3
 *   the ten cubic constraints det E = 0, 2 E E^T E - tr(E E^T) E = 0 as a 10 x 20 matrix over the
 *   monomials  x^3 y^3 x^2y xy^2 x^2z x^2 y^2z y^2 xyz xy | xz^2 xz x yz^2 yz y z^3 z^2 z 1
 *   A <- A[:, :10]^-1 A[:, 10:]; rows 4..9 pair up into a 3 x 3 polynomial matrix B(z);
 *   det B(z) has degree 10 -> solvePoly; real roots (|Im| <= 1e-10) give (x, y) from the null
 *   vector of B(z) (SVD::solveZ) and E, normalised to unit Frobenius norm.
 * OpenCV spells the coefficient matrix and the determinant out as expanded expressions; here
 * they are built by polynomial arithmetic — same algebra, rounding differs in the last bits.
 * Models come out in solvePoly's root order (it decides ties: RANSAC keeps the FIRST best).
 * ---------------------------------------------------------------------------------------- */
typedef struct { double c[4][4][4]; } poly3; /* c[i][j][k] multiplies x^i y^j z^k, total degree <= 3 */

static void p_lin(poly3* p, double cx, double cy, double cz, double c1) {
    memset(p, 0, sizeof(*p));
    p->c[1][0][0] = cx; p->c[0][1][0] = cy; p->c[0][0][1] = cz; p->c[0][0][0] = c1;
}
static void p_mul(poly3* out, const poly3* a, const poly3* b) {
    poly3 r;
    memset(&r, 0, sizeof(r));
    for (int i = 0; i < 4; ++i)
        for (int j = 0; i + j < 4; ++j)
            for (int k = 0; i + j + k < 4; ++k) {
                const double av = a->c[i][j][k];
                if (av == 0) continue;
                for (int l = 0; i + l < 4; ++l)
                    for (int m = 0; j + m < 4; ++m)
                        for (int o = 0; k + o < 4; ++o)
                            if (i + j + k + l + m + o < 4) r.c[i + l][j + m][k + o] += av * b->c[l][m][o];
            }
    *out = r;
}
static void p_axpy(poly3* y, double a, const poly3* x) {
    for (int i = 0; i < 64; ++i) (&y->c[0][0][0])[i] += a * (&x->c[0][0][0])[i];
}

static const int kMono[20][3] = {{3, 0, 0}, {0, 3, 0}, {2, 1, 0}, {1, 2, 0}, {2, 0, 1}, {2, 0, 0}, {0, 2, 1},
                                 {0, 2, 0}, {1, 1, 1}, {1, 1, 0}, {1, 0, 2}, {1, 0, 1}, {1, 0, 0}, {0, 1, 2},
                                 {0, 1, 1}, {0, 1, 0}, {0, 0, 3}, {0, 0, 2}, {0, 0, 1}, {0, 0, 0}};

/* a(z) * b(z): coefficient arrays in ASCENDING powers, degrees da, db */
static void up_mul(const double* a, int da, const double* b, int db, double* out) {
    for (int i = 0; i <= da + db; ++i) out[i] = 0;
    for (int i = 0; i <= da; ++i)
        for (int j = 0; j <= db; ++j) out[i + j] += a[i] * b[j];
}

int orc_five_point(const double* x1, const double* x2, double* E_out) {
    double Q[5 * 9], w[5], Vt[9 * 9];
    for (int i = 0; i < 5; ++i) {
        const double a1 = x1[2 * i], b1 = x1[2 * i + 1], a2 = x2[2 * i], b2 = x2[2 * i + 1];
        double* q = Q + 9 * i;
        q[0] = a1 * a2; q[1] = b1 * a2; q[2] = a2;
        q[3] = a1 * b2; q[4] = b1 * b2; q[5] = b2;
        q[6] = a1;      q[7] = b1;      q[8] = 1.0;
    }
    orc_svd(Q, 5, 9, 1, w, NULL, Vt);
    const double* EE[4] = {Vt + 9 * 5, Vt + 9 * 6, Vt + 9 * 7, Vt + 9 * 8};

    poly3 e[3][3];
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) p_lin(&e[r][c], EE[0][3 * r + c], EE[1][3 * r + c], EE[2][3 * r + c], EE[3][3 * r + c]);
    poly3 rows[10], t0, t1;
    /* det(E) by the first row */
    memset(&rows[0], 0, sizeof(poly3));
    {
        const int cof[3][2] = {{1, 2}, {0, 2}, {0, 1}};
        for (int c = 0; c < 3; ++c) {
            poly3 minor;
            p_mul(&t0, &e[1][cof[c][0]], &e[2][cof[c][1]]);
            p_mul(&t1, &e[1][cof[c][1]], &e[2][cof[c][0]]);
            minor = t0;
            p_axpy(&minor, -1.0, &t1);
            p_mul(&t0, &e[0][c], &minor);
            p_axpy(&rows[0], c == 1 ? -1.0 : 1.0, &t0);
        }
    }
    /* 2 E E^T E - tr(E E^T) E */
    poly3 eet[3][3], tr;
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) {
            memset(&eet[r][c], 0, sizeof(poly3));
            for (int k = 0; k < 3; ++k) {
                p_mul(&t0, &e[r][k], &e[c][k]);
                p_axpy(&eet[r][c], 1.0, &t0);
            }
        }
    tr = eet[0][0];
    p_axpy(&tr, 1.0, &eet[1][1]);
    p_axpy(&tr, 1.0, &eet[2][2]);
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) {
            poly3* out = &rows[1 + 3 * r + c];
            memset(out, 0, sizeof(poly3));
            for (int k = 0; k < 3; ++k) {
                p_mul(&t0, &eet[r][k], &e[k][c]);
                p_axpy(out, 2.0, &t0);
            }
            p_mul(&t0, &tr, &e[r][c]);
            p_axpy(out, -1.0, &t0);
        }
    double A[10][20];
    for (int r = 0; r < 10; ++r)
        for (int m = 0; m < 20; ++m) A[r][m] = rows[r].c[kMono[m][0]][kMono[m][1]][kMono[m][2]];

    /* A[:, :10]^-1 * A[:, 10:]  — cv::Mat::inv() (LU with partial pivoting) on the augmented system */
    for (int col = 0; col < 10; ++col) {
        int piv = col;
        for (int r = col + 1; r < 10; ++r)
            if (fabs(A[r][col]) > fabs(A[piv][col])) piv = r;
        if (fabs(A[piv][col]) < DBL_EPSILON) return 0;
        if (piv != col)
            for (int m = 0; m < 20; ++m) { const double t = A[col][m]; A[col][m] = A[piv][m]; A[piv][m] = t; }
        const double d = 1 / A[col][col];
        for (int r = col + 1; r < 10; ++r) {
            const double f = A[r][col] * d;
            for (int m = col; m < 20; ++m) A[r][m] -= f * A[col][m];
        }
    }
    for (int col = 9; col >= 0; --col) {
        const double d = 1 / A[col][col];
        for (int m = 10; m < 20; ++m) {
            double s = A[col][m];
            for (int k = col + 1; k < 10; ++k) s -= A[col][k] * A[k][m];
            A[col][m] = s * d;
        }
    }
    /* B(z): rows (4,5), (6,7), (8,9):  row_a - z row_b = x p1(z) + y p2(z) + p3(z);  b[13] as in OpenCV:
     * [0..3] x (z^3 z^2 z 1), [4..7] y, [8..12] 1 (z^4 .. 1) */
    double b[3][13];
    for (int i = 0; i < 3; ++i) {
        const double* ra = &A[2 * i + 4][10];
        const double* rb = &A[2 * i + 5][10];
        double r1[13] = {0}, r2[13] = {0};
        for (int k = 0; k < 3; ++k) { r1[1 + k] = ra[k]; r1[5 + k] = ra[3 + k]; }
        for (int k = 0; k < 4; ++k) r1[9 + k] = ra[6 + k];
        for (int k = 0; k < 3; ++k) { r2[k] = rb[k]; r2[4 + k] = rb[3 + k]; }
        for (int k = 0; k < 4; ++k) r2[8 + k] = rb[6 + k];
        for (int k = 0; k < 13; ++k) b[i][k] = r1[k] - r2[k];
    }
    /* det B(z), ascending coefficient arrays */
    double pz[3][3][5];
    int dg[3] = {3, 3, 4};
    for (int i = 0; i < 3; ++i) {
        for (int k = 0; k < 4; ++k) { pz[i][0][k] = b[i][3 - k]; pz[i][1][k] = b[i][7 - k]; }
        for (int k = 0; k < 5; ++k) pz[i][2][k] = b[i][12 - k];
    }
    double c[11];
    for (int k = 0; k < 11; ++k) c[k] = 0;
    {
        static const int perm[6][4] = {{0, 1, 2, 1}, {1, 2, 0, 1}, {2, 0, 1, 1}, {2, 1, 0, -1}, {1, 0, 2, -1}, {0, 2, 1, -1}};
        for (int p = 0; p < 6; ++p) {
            double t[11], u[11];
            const int c0 = perm[p][0], c1 = perm[p][1], c2 = perm[p][2];
            up_mul(pz[0][c0], dg[c0], pz[1][c1], dg[c1], t);
            up_mul(t, dg[c0] + dg[c1], pz[2][c2], dg[c2], u);
            for (int k = 0; k <= 10; ++k) c[k] += perm[p][3] * u[k];
        }
    }
    for (int k = 0; k < 11; ++k)
        if (!isfinite(c[k])) return 0;
    double rre[10], rim[10];
    const int nroots = orc_solve_poly(c, 10, rre, rim, 300);
    int count = 0;
    for (int i = 0; i < nroots; ++i) {
        if (fabs(rim[i]) > 1e-10) continue;
        const double z1 = rre[i], z2 = z1 * z1, z3 = z2 * z1, z4 = z3 * z1;
        double bz[9], ww[3], vt[9];
        for (int j = 0; j < 3; ++j) {
            const double* br = b[j];
            bz[3 * j + 0] = br[0] * z3 + br[1] * z2 + br[2] * z1 + br[3];
            bz[3 * j + 1] = br[4] * z3 + br[5] * z2 + br[6] * z1 + br[7];
            bz[3 * j + 2] = br[8] * z4 + br[9] * z3 + br[10] * z2 + br[11] * z1 + br[12];
        }
        orc_svd(bz, 3, 3, 0, ww, NULL, vt); /* SVD::solveZ: last row of vt */
        const double* xy1 = vt + 6;
        if (fabs(xy1[2]) < 1e-10) continue;
        const double x = xy1[0] / xy1[2], y = xy1[1] / xy1[2];
        double Ev[9], nrm = 0;
        for (int k = 0; k < 9; ++k) {
            Ev[k] = EE[0][k] * x + EE[1][k] * y + EE[2][k] * z1 + EE[3][k];
            nrm += Ev[k] * Ev[k];
        }
        nrm = sqrt(nrm);
        for (int k = 0; k < 9; ++k) E_out[9 * count + k] = Ev[k] / nrm;
        ++count;
    }
    return count;
}

/* ------------------------------------------------------------------------------------------
 * K-normalisation of both RANSAC entry points (calib3d/five-point.cpp):
 *   points.convertTo(CV_64F);  points.col(0) = (points.col(0) - cx) / fx;  …
 * The matrix expression folds into ONE scaled conversion dst = src * (1/fx) + (-cx * (1/fx))
 * (MatOp_AddEx: alpha *= 1/s, beta *= 1/s).  Restated with separately rounded mul and add
 * (SSE2 baseline; an FMA3-dispatched build fuses them — unpinned).
 * ---------------------------------------------------------------------------------------- */
static void k_normalise(const float* pts, int64_t n, const double* K, double* out) {
    const double ifx = 1. / K[0], ify = 1. / K[4];
    const double bx = -K[2] * ifx, by = -K[5] * ify;
    for (int64_t i = 0; i < n; ++i) {
        out[2 * i] = (double)pts[2 * i] * ifx + bx;
        out[2 * i + 1] = (double)pts[2 * i + 1] * ify + by;
    }
}

void orc_k_normalise(const float* pts, int64_t n, const double* K, double* out) { k_normalise(pts, n, K, out); }

/* getSubset (ptsetreg.cpp): modelPoints distinct indices, a duplicate is redrawn in place. */
static void get_subset(uint64_t* rng, int count, int model_points, int* idx) {
    for (int i = 0; i < model_points; ++i) {
        int v;
        for (;;) {
            v = orc_rng_uniform(rng, 0, count);
            int dup = 0;
            for (int j = 0; j < i; ++j) dup |= idx[j] == v;
            if (!dup) break;
        }
        idx[i] = v;
    }
}

/* ------------------------------------------------------------------------------------------
 * cv2.findEssentialMat(points1, points2, K, RANSAC, prob, threshold)              sfm.py:307
 * RANSACPointSetRegistrator::run (ptsetreg.cpp) over EMEstimatorCallback, maxIters = 1000:
 * every model of an iteration is scored over ALL points (Sampson distance as float <=
 * (float)(thr^2), thr = threshold / ((fx+fy)/2)); the best is replaced on a STRICTLY larger count
 * (> max(best, 4)) and `niters` re-estimated then; `niters` is looked at only between
 * iterations.  Returns the number of 3x3 models written to E (1; k <= 10 when n == 5: OpenCV
 * then returns them stacked; 0 on failure).  mask is {0,1}.  stats (optional): iterations run,
 * models scored, best count.
 * ---------------------------------------------------------------------------------------- */
int orc_find_essential_mat(const float* pts0, const float* pts1, int64_t n, const double* K, double prob,
                           double threshold, int max_iters, double* E, uint8_t* mask, int* stats) {
    if (stats) stats[0] = stats[1] = stats[2] = 0;
    if (n < 5) return 0;
    double* x0 = (double*)malloc(sizeof(double) * 2 * (size_t)n);
    double* x1 = (double*)malloc(sizeof(double) * 2 * (size_t)n);
    uint8_t* cur = (uint8_t*)malloc((size_t)n);
    k_normalise(pts0, n, K, x0);
    k_normalise(pts1, n, K, x1);
    double thr = threshold;
    thr /= (K[0] + K[4]) / 2;
    const float thr2 = (float)(thr * thr);
    int result = 0;
    if (n == 5) {
        result = orc_five_point(x0, x1, E);
        if (result > 0) memset(mask, 1, (size_t)n);
    } else {
        uint64_t rng = ~(uint64_t)0;
        int niters = max_iters > 1 ? max_iters : 1, best = 0, scored = 0, iter;
        for (iter = 0; iter < niters; ++iter) {
            int idx[5];
            double s0[10], s1[10], models[90];
            get_subset(&rng, (int)n, 5, idx);
            for (int k = 0; k < 5; ++k) {
                s0[2 * k] = x0[2 * idx[k]]; s0[2 * k + 1] = x0[2 * idx[k] + 1];
                s1[2 * k] = x1[2 * idx[k]]; s1[2 * k + 1] = x1[2 * idx[k] + 1];
            }
            const int nmodels = orc_five_point(s0, s1, models);
            for (int m = 0; m < nmodels; ++m) {
                int32_t good;
                orc_score_essential(models + 9 * m, 1, x0, x1, n, thr2, &good, cur);
                ++scored;
                if (good > (best > 4 ? best : 4)) {
                    memcpy(mask, cur, (size_t)n);
                    memcpy(E, models + 9 * m, 9 * sizeof(double));
                    best = good;
                    niters = orc_ransac_update_num_iters(prob, (double)(n - good) / (double)n, 5, niters);
                }
            }
        }
        result = best > 0;
        if (stats) { stats[0] = iter; stats[1] = scored; stats[2] = best; }
    }
    free(x0); free(x1); free(cur);
    return result;
}

/* ------------------------------------------------------------------------------------------
 * cv::decomposeEssentialMat (five-point.cpp): E = U D Vt, det-fixed, W = [0 1 0; -1 0 0; 0 0 1]:
 * R1 = U W Vt, R2 = U W^T Vt, t = U[:, 2].
 * ---------------------------------------------------------------------------------------- */
static void mat3_mul(const double* A, const double* B, double* C) {
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            double s = 0;
            for (int k = 0; k < 3; ++k) s += A[3 * i + k] * B[3 * k + j];
            C[3 * i + j] = s;
        }
}

void orc_decompose_essential(const double* E, double* R1, double* R2, double* t) {
    double w[3], U[9], Vt[9], T[9];
    orc_svd(E, 3, 3, 0, w, U, Vt);
    if (det3(U) < 0)
        for (int k = 0; k < 9; ++k) U[k] *= -1.;
    if (det3(Vt) < 0)
        for (int k = 0; k < 9; ++k) Vt[k] *= -1.;
    const double W[9] = {0, 1, 0, -1, 0, 0, 0, 0, 1}, Wt[9] = {0, -1, 0, 1, 0, 0, 0, 0, 1};
    mat3_mul(U, W, T);
    mat3_mul(T, Vt, R1);
    mat3_mul(U, Wt, T);
    mat3_mul(T, Vt, R2);
    t[0] = U[2] * 1.0; t[1] = U[5] * 1.0; t[2] = U[8] * 1.0;
}

/* ------------------------------------------------------------------------------------------
 * cv2.recoverPose(E, points1, points2, K)                                         sfm.py:311
 * candidates (R1,t) (R2,t) (R1,-t) (R2,-t); each triangulates every K-normalised point against
 * [I|0] (orc_recover_pose_score); the first candidate whose count is >= all the others wins
 * (the cascade of >= tests).  distanceThresh = 50.  mask is {0,255}; returns the count.
 * ---------------------------------------------------------------------------------------- */
int orc_recover_pose(const double* E, const float* pts0, const float* pts1, int64_t n, const double* K, double dist,
                     int rows, double* R, double* t, uint8_t* mask) {
    double R1[9], R2[9], tt[3], Ps[4 * 12];
    orc_decompose_essential(E, R1, R2, tt);
    const double* Rs[4] = {R1, R2, R1, R2};
    const double sg[4] = {1, 1, -1, -1};
    for (int c = 0; c < 4; ++c)
        for (int i = 0; i < 3; ++i) {
            for (int j = 0; j < 3; ++j) Ps[12 * c + 4 * i + j] = Rs[c][3 * i + j];
            Ps[12 * c + 4 * i + 3] = sg[c] * tt[i];
        }
    double* x0 = (double*)malloc(sizeof(double) * 2 * (size_t)(n > 0 ? n : 1));
    double* x1 = (double*)malloc(sizeof(double) * 2 * (size_t)(n > 0 ? n : 1));
    uint8_t* masks = (uint8_t*)malloc(4 * (size_t)(n > 0 ? n : 1));
    k_normalise(pts0, n, K, x0);
    k_normalise(pts1, n, K, x1);
    int32_t g[4];
    orc_recover_pose_score(Ps, 4, x0, x1, n, dist, rows, g, masks);
    int k;
    if (g[0] >= g[1] && g[0] >= g[2] && g[0] >= g[3]) k = 0;
    else if (g[1] >= g[0] && g[1] >= g[2] && g[1] >= g[3]) k = 1;
    else if (g[2] >= g[0] && g[2] >= g[1] && g[2] >= g[3]) k = 2;
    else k = 3;
    memcpy(R, Rs[k], 9 * sizeof(double));
    for (int i = 0; i < 3; ++i) t[i] = sg[k] * tt[i];
    if (mask) memcpy(mask, masks + (size_t)k * (size_t)n, (size_t)n);
    free(x0); free(x1); free(masks);
    return g[k];
}

/* ------------------------------------------------------------------------------------------
 * EPnP (calib3d/epnp.cpp — Lepetit, Moreno-Noguer, Fua; OpenCV keeps the authors' code).
 * Routine by routine: choose_control_points, compute_barycentric_coordinates, fill_M, M^T M,
 * SVD, compute_L_6x10, compute_rho, find_betas_approx_{1,2,3} (cvSolve SVD), gauss_newton
 * (5 steps, Householder qr_solve), compute_R_and_t (compute_ccs / compute_pcs / solve_for_sign /
 * estimate_R_and_t), reprojection_error; the smallest mean error wins.
 * Xw n x 3, uv n x 2 pixels (doubles), K = (fu, fv, uc, vc).  n <= EPNP_MAXN.
 * ---------------------------------------------------------------------------------------- */
#define EPNP_MAXN 64
typedef struct {
    int n;
    double fu, fv, uc, vc;
    const double* pws;
    const double* us;
    double alphas[4 * EPNP_MAXN], pcs[3 * EPNP_MAXN];
    double cws[4][3], ccs[4][3];
} epnp_t;

static double dot3(const double* a, const double* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
static double dist2(const double* p1, const double* p2) {
    return (p1[0] - p2[0]) * (p1[0] - p2[0]) + (p1[1] - p2[1]) * (p1[1] - p2[1]) + (p1[2] - p2[2]) * (p1[2] - p2[2]);
}

/* Householder QR least squares, the authors' routine including its pivot scan (which starts at row k twice and
 * never looks at the last row).  A is nr x nc row-major and is destroyed. */
static void epnp_qr_solve(double* pA, int nr, int nc, double* pb, double* pX) {
    double A1[8], A2[8];
    double* ppAkk = pA;
    for (int k = 0; k < nc; ++k) {
        double* ppAik1 = ppAkk;
        double eta = fabs(*ppAik1);
        for (int i = k + 1; i < nr; ++i) {
            const double elt = fabs(*ppAik1);
            if (eta < elt) eta = elt;
            ppAik1 += nc;
        }
        if (eta == 0) {
            A1[k] = A2[k] = 0.0;
            return;
        }
        double* ppAik2 = ppAkk;
        double sum2 = 0.0;
        const double inv_eta = 1. / eta;
        for (int i = k; i < nr; ++i) {
            *ppAik2 *= inv_eta;
            sum2 += *ppAik2 * *ppAik2;
            ppAik2 += nc;
        }
        double sigma = sqrt(sum2);
        if (*ppAkk < 0) sigma = -sigma;
        *ppAkk += sigma;
        A1[k] = sigma * *ppAkk;
        A2[k] = -eta * sigma;
        for (int j = k + 1; j < nc; ++j) {
            double* ppAik = ppAkk;
            double sum = 0;
            for (int i = k; i < nr; ++i) {
                sum += *ppAik * ppAik[j - k];
                ppAik += nc;
            }
            const double tau = sum / A1[k];
            ppAik = ppAkk;
            for (int i = k; i < nr; ++i) {
                ppAik[j - k] -= tau * *ppAik;
                ppAik += nc;
            }
        }
        ppAkk += nc + 1;
    }
    double* ppAjj = pA;
    for (int j = 0; j < nc; ++j) {
        double* ppAij = ppAjj;
        double tau = 0;
        for (int i = j; i < nr; ++i) {
            tau += *ppAij * pb[i];
            ppAij += nc;
        }
        tau /= A1[j];
        ppAij = ppAjj;
        for (int i = j; i < nr; ++i) {
            pb[i] -= tau * *ppAij;
            ppAij += nc;
        }
        ppAjj += nc + 1;
    }
    pX[nc - 1] = pb[nc - 1] / A2[nc - 1];
    for (int i = nc - 2; i >= 0; --i) {
        const double* ppAij = pA + i * nc + (i + 1);
        double sum = 0;
        for (int j = i + 1; j < nc; ++j) {
            sum += *ppAij * pX[j];
            ++ppAij;
        }
        pX[i] = (pb[i] - sum) / A2[i];
    }
}

static void epnp_gauss_newton(const double* L, const double* rho, double* betas) {
    for (int it = 0; it < 5; ++it) {
        double A[24], b[6], x[4] = {0, 0, 0, 0};
        for (int i = 0; i < 6; ++i) {
            const double* r = L + 10 * i;
            double* a = A + 4 * i;
            a[0] = 2 * r[0] * betas[0] + r[1] * betas[1] + r[3] * betas[2] + r[6] * betas[3];
            a[1] = r[1] * betas[0] + 2 * r[2] * betas[1] + r[4] * betas[2] + r[7] * betas[3];
            a[2] = r[3] * betas[0] + r[4] * betas[1] + 2 * r[5] * betas[2] + r[8] * betas[3];
            a[3] = r[6] * betas[0] + r[7] * betas[1] + r[8] * betas[2] + 2 * r[9] * betas[3];
            b[i] = rho[i] - (r[0] * betas[0] * betas[0] + r[1] * betas[0] * betas[1] + r[2] * betas[1] * betas[1] +
                             r[3] * betas[0] * betas[2] + r[4] * betas[1] * betas[2] + r[5] * betas[2] * betas[2] +
                             r[6] * betas[0] * betas[3] + r[7] * betas[1] * betas[3] + r[8] * betas[2] * betas[3] +
                             r[9] * betas[3] * betas[3]);
        }
        epnp_qr_solve(A, 6, 4, b, x);
        for (int i = 0; i < 4; ++i) betas[i] += x[i];
    }
}

static double epnp_R_and_t(epnp_t* e, const double* ut, const double* betas, double R[3][3], double t[3]) {
    const int n = e->n;
    /* compute_ccs */
    for (int i = 0; i < 4; ++i) e->ccs[i][0] = e->ccs[i][1] = e->ccs[i][2] = 0.0;
    for (int i = 0; i < 4; ++i) {
        const double* v = ut + 12 * (11 - i);
        for (int j = 0; j < 4; ++j)
            for (int k = 0; k < 3; ++k) e->ccs[j][k] += betas[i] * v[3 * j + k];
    }
    /* compute_pcs */
    for (int i = 0; i < n; ++i) {
        const double* a = e->alphas + 4 * i;
        double* pc = e->pcs + 3 * i;
        for (int j = 0; j < 3; ++j) pc[j] = a[0] * e->ccs[0][j] + a[1] * e->ccs[1][j] + a[2] * e->ccs[2][j] + a[3] * e->ccs[3][j];
    }
    /* solve_for_sign */
    if (e->pcs[2] < 0.0) {
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 3; ++j) e->ccs[i][j] = -e->ccs[i][j];
        for (int i = 0; i < 3 * n; ++i) e->pcs[i] = -e->pcs[i];
    }
    /* estimate_R_and_t */
    double pc0[3] = {0, 0, 0}, pw0[3] = {0, 0, 0};
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < 3; ++j) {
            pc0[j] += e->pcs[3 * i + j];
            pw0[j] += e->pws[3 * i + j];
        }
    for (int j = 0; j < 3; ++j) {
        pc0[j] /= n;
        pw0[j] /= n;
    }
    double abt[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, d[3], U[9], Vt[9];
    for (int i = 0; i < n; ++i) {
        const double* pc = e->pcs + 3 * i;
        const double* pw = e->pws + 3 * i;
        for (int j = 0; j < 3; ++j) {
            abt[3 * j] += (pc[j] - pc0[j]) * (pw[0] - pw0[0]);
            abt[3 * j + 1] += (pc[j] - pc0[j]) * (pw[1] - pw0[1]);
            abt[3 * j + 2] += (pc[j] - pc0[j]) * (pw[2] - pw0[2]);
        }
    }
    orc_svd(abt, 3, 3, 0, d, U, Vt); /* cvSVD(ABt, D, U, V): U and V (not transposed): R = U V^T */
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) R[i][j] = U[3 * i] * Vt[j] + U[3 * i + 1] * Vt[3 + j] + U[3 * i + 2] * Vt[6 + j];
    const double det = R[0][0] * R[1][1] * R[2][2] + R[0][1] * R[1][2] * R[2][0] + R[0][2] * R[1][0] * R[2][1] -
                       R[0][2] * R[1][1] * R[2][0] - R[0][1] * R[1][0] * R[2][2] - R[0][0] * R[1][2] * R[2][1];
    if (det < 0) {
        R[2][0] = -R[2][0];
        R[2][1] = -R[2][1];
        R[2][2] = -R[2][2];
    }
    t[0] = pc0[0] - dot3(R[0], pw0);
    t[1] = pc0[1] - dot3(R[1], pw0);
    t[2] = pc0[2] - dot3(R[2], pw0);
    /* reprojection_error */
    double sum2 = 0.0;
    for (int i = 0; i < n; ++i) {
        const double* pw = e->pws + 3 * i;
        const double Xc = dot3(R[0], pw) + t[0], Yc = dot3(R[1], pw) + t[1];
        const double inv_Zc = 1.0 / (dot3(R[2], pw) + t[2]);
        const double ue = e->uc + e->fu * Xc * inv_Zc, ve = e->vc + e->fv * Yc * inv_Zc;
        const double u = e->us[2 * i], v = e->us[2 * i + 1];
        sum2 += sqrt((u - ue) * (u - ue) + (v - ve) * (v - ve));
    }
    return sum2 / n;
}

int orc_epnp(const double* K, const double* Xw, const double* uv, int n, double* Rout, double* tout) {
    if (n < 4 || n > EPNP_MAXN) return -1;
    epnp_t e;
    e.n = n;
    e.fu = K[0]; e.fv = K[4]; e.uc = K[2]; e.vc = K[5];
    e.pws = Xw;
    e.us = uv;
    /* choose_control_points */
    e.cws[0][0] = e.cws[0][1] = e.cws[0][2] = 0;
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < 3; ++j) e.cws[0][j] += Xw[3 * i + j];
    for (int j = 0; j < 3; ++j) e.cws[0][j] /= n;
    {
        double pw0tpw0[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, dc[3], U[9];
        /* cvMulTransposed(PW0, PW0tPW0, 1): sum over the points, one entry at a time */
        for (int a = 0; a < 3; ++a)
            for (int b = 0; b < 3; ++b) {
                double s = 0;
                for (int i = 0; i < n; ++i) s += (Xw[3 * i + a] - e.cws[0][a]) * (Xw[3 * i + b] - e.cws[0][b]);
                pw0tpw0[3 * a + b] = s;
            }
        orc_svd(pw0tpw0, 3, 3, 0, dc, U, NULL);          /* CV_SVD_U_T: uct[3*(i-1)+j] = U[j][i-1] */
        for (int i = 1; i < 4; ++i) {
            const double k = sqrt(dc[i - 1] / n);
            for (int j = 0; j < 3; ++j) e.cws[i][j] = e.cws[0][j] + k * U[3 * j + (i - 1)];
        }
    }
    /* compute_barycentric_coordinates */
    {
        double cc[9], ci[9];
        for (int i = 0; i < 3; ++i)
            for (int j = 1; j < 4; ++j) cc[3 * i + j - 1] = e.cws[j][i] - e.cws[0][i];
        svd_invert(cc, 3, ci);
        for (int i = 0; i < n; ++i) {
            const double* pi = Xw + 3 * i;
            double* a = e.alphas + 4 * i;
            for (int j = 0; j < 3; ++j)
                a[1 + j] = ci[3 * j] * (pi[0] - e.cws[0][0]) + ci[3 * j + 1] * (pi[1] - e.cws[0][1]) + ci[3 * j + 2] * (pi[2] - e.cws[0][2]);
            a[0] = 1.0f - a[1] - a[2] - a[3];
        }
    }
    /* fill_M, M^T M (cvMulTransposed: each entry summed over the 2n rows in order), SVD with CV_SVD_U_T */
    double* M = (double*)malloc(sizeof(double) * 24 * (size_t)n);
    for (int i = 0; i < n; ++i) {
        const double* as = e.alphas + 4 * i;
        const double u = uv[2 * i], v = uv[2 * i + 1];
        double* M1 = M + 24 * i;
        double* M2 = M1 + 12;
        for (int k = 0; k < 4; ++k) {
            M1[3 * k] = as[k] * e.fu; M1[3 * k + 1] = 0.0;        M1[3 * k + 2] = as[k] * (e.uc - u);
            M2[3 * k] = 0.0;        M2[3 * k + 1] = as[k] * e.fv; M2[3 * k + 2] = as[k] * (e.vc - v);
        }
    }
    double mtm[144], d[12], U[144], ut[144];
    for (int a = 0; a < 12; ++a)
        for (int b = 0; b < 12; ++b) {
            double s = 0;
            for (int r = 0; r < 2 * n; ++r) s += M[12 * r + a] * M[12 * r + b];
            mtm[12 * a + b] = s;
        }
    free(M);
    orc_svd(mtm, 12, 12, 0, d, U, NULL);
    for (int i = 0; i < 12; ++i)
        for (int k = 0; k < 12; ++k) ut[12 * i + k] = U[12 * k + i];
    /* compute_L_6x10, compute_rho */
    double L[60], rho[6];
    {
        const double* v[4] = {ut + 12 * 11, ut + 12 * 10, ut + 12 * 9, ut + 12 * 8};
        double dv[4][6][3];
        for (int i = 0; i < 4; ++i) {
            int a = 0, b = 1;
            for (int j = 0; j < 6; ++j) {
                dv[i][j][0] = v[i][3 * a] - v[i][3 * b];
                dv[i][j][1] = v[i][3 * a + 1] - v[i][3 * b + 1];
                dv[i][j][2] = v[i][3 * a + 2] - v[i][3 * b + 2];
                ++b;
                if (b > 3) {
                    ++a;
                    b = a + 1;
                }
            }
        }
        for (int i = 0; i < 6; ++i) {
            double* row = L + 10 * i;
            row[0] = dot3(dv[0][i], dv[0][i]);
            row[1] = 2.0f * dot3(dv[0][i], dv[1][i]);
            row[2] = dot3(dv[1][i], dv[1][i]);
            row[3] = 2.0f * dot3(dv[0][i], dv[2][i]);
            row[4] = 2.0f * dot3(dv[1][i], dv[2][i]);
            row[5] = dot3(dv[2][i], dv[2][i]);
            row[6] = 2.0f * dot3(dv[0][i], dv[3][i]);
            row[7] = 2.0f * dot3(dv[1][i], dv[3][i]);
            row[8] = 2.0f * dot3(dv[2][i], dv[3][i]);
            row[9] = dot3(dv[3][i], dv[3][i]);
        }
        rho[0] = dist2(e.cws[0], e.cws[1]); rho[1] = dist2(e.cws[0], e.cws[2]); rho[2] = dist2(e.cws[0], e.cws[3]);
        rho[3] = dist2(e.cws[1], e.cws[2]); rho[4] = dist2(e.cws[1], e.cws[3]); rho[5] = dist2(e.cws[2], e.cws[3]);
    }
    double Betas[4][4], rep[4], Rs[4][3][3], ts[4][3];
    { /* find_betas_approx_1: [B11 B12 B13 B14] */
        double l[24], b4[4];
        for (int i = 0; i < 6; ++i) { l[4 * i] = L[10 * i]; l[4 * i + 1] = L[10 * i + 1]; l[4 * i + 2] = L[10 * i + 3]; l[4 * i + 3] = L[10 * i + 6]; }
        svd_solve(l, 6, 4, rho, b4);
        double* be = Betas[1];
        if (b4[0] < 0) {
            be[0] = sqrt(-b4[0]); be[1] = -b4[1] / be[0]; be[2] = -b4[2] / be[0]; be[3] = -b4[3] / be[0];
        } else {
            be[0] = sqrt(b4[0]); be[1] = b4[1] / be[0]; be[2] = b4[2] / be[0]; be[3] = b4[3] / be[0];
        }
    }
    epnp_gauss_newton(L, rho, Betas[1]);
    rep[1] = epnp_R_and_t(&e, ut, Betas[1], Rs[1], ts[1]);
    { /* find_betas_approx_2: [B11 B12 B22] */
        double l[18], b3[3];
        for (int i = 0; i < 6; ++i) { l[3 * i] = L[10 * i]; l[3 * i + 1] = L[10 * i + 1]; l[3 * i + 2] = L[10 * i + 2]; }
        svd_solve(l, 6, 3, rho, b3);
        double* be = Betas[2];
        if (b3[0] < 0) {
            be[0] = sqrt(-b3[0]);
            be[1] = (b3[2] < 0) ? sqrt(-b3[2]) : 0.0;
        } else {
            be[0] = sqrt(b3[0]);
            be[1] = (b3[2] > 0) ? sqrt(b3[2]) : 0.0;
        }
        if (b3[1] < 0) be[0] = -be[0];
        be[2] = 0.0;
        be[3] = 0.0;
    }
    epnp_gauss_newton(L, rho, Betas[2]);
    rep[2] = epnp_R_and_t(&e, ut, Betas[2], Rs[2], ts[2]);
    { /* find_betas_approx_3: [B11 B12 B22 B13 B23] */
        double l[30], b5[5];
        for (int i = 0; i < 6; ++i)
            for (int k = 0; k < 5; ++k) l[5 * i + k] = L[10 * i + k];
        svd_solve(l, 6, 5, rho, b5);
        double* be = Betas[3];
        if (b5[0] < 0) {
            be[0] = sqrt(-b5[0]);
            be[1] = (b5[2] < 0) ? sqrt(-b5[2]) : 0.0;
        } else {
            be[0] = sqrt(b5[0]);
            be[1] = (b5[2] > 0) ? sqrt(b5[2]) : 0.0;
        }
        if (b5[1] < 0) be[0] = -be[0];
        be[2] = b5[3] / be[0];
        be[3] = 0.0;
    }
    epnp_gauss_newton(L, rho, Betas[3]);
    rep[3] = epnp_R_and_t(&e, ut, Betas[3], Rs[3], ts[3]);
    int N = 1;
    if (rep[2] < rep[1]) N = 2;
    if (rep[3] < rep[N]) N = 3;
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) Rout[3 * i + j] = Rs[N][i][j];
        tout[i] = ts[N][i];
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * cvProjectPoints2 with dpdr / dpdt and all-zero distortion (calib3d/calibration.cpp): with
 * k = 0 every distortion factor is exactly 1 and every added term exactly 0, so the Jacobian
 * rows are fx * {z, 0, -x z} / fy * {0, z, -y z} and fx * z (dx0 - x dz0) / fy * z (dy0 - y dz0).
 * ---------------------------------------------------------------------------------------- */
static void project_jac(const double* rvec, const double* tvec, const double* K, const double* X, int64_t n,
                        double* proj /*2n*/, double* J /*2n x 6 or NULL*/) {
    double R[9], dRdr[27];
    orc_rodrigues_vec2mat(rvec, R, J ? dRdr : NULL);
    const double fx = K[0], fy = K[4], cx = K[2], cy = K[5];
    for (int64_t i = 0; i < n; ++i) {
        const double Xw = X[3 * i], Yw = X[3 * i + 1], Zw = X[3 * i + 2];
        double x = R[0] * Xw + R[1] * Yw + R[2] * Zw + tvec[0];
        double y = R[3] * Xw + R[4] * Yw + R[5] * Zw + tvec[1];
        double z = R[6] * Xw + R[7] * Yw + R[8] * Zw + tvec[2];
        z = z ? 1. / z : 1;
        x *= z;
        y *= z;
        proj[2 * i] = x * fx + cx;
        proj[2 * i + 1] = y * fy + cy;
        if (J) {
            double* jx = J + 12 * i;
            double* jy = jx + 6;
            for (int j = 0; j < 3; ++j) {
                const double dx0 = Xw * dRdr[9 * j] + Yw * dRdr[9 * j + 1] + Zw * dRdr[9 * j + 2];
                const double dy0 = Xw * dRdr[9 * j + 3] + Yw * dRdr[9 * j + 4] + Zw * dRdr[9 * j + 5];
                const double dz0 = Xw * dRdr[9 * j + 6] + Yw * dRdr[9 * j + 7] + Zw * dRdr[9 * j + 8];
                jx[j] = fx * (z * (dx0 - x * dz0));
                jy[j] = fy * (z * (dy0 - y * dz0));
            }
            jx[3] = fx * z; jx[4] = fx * 0; jx[5] = fx * (-x * z);
            jy[3] = fy * 0; jy[4] = fy * z; jy[5] = fy * (-y * z);
        }
    }
}

/* ------------------------------------------------------------------------------------------
 * cv::solvePnP(..., SOLVEPNP_ITERATIVE) = cvFindExtrinsicCameraParams2 (calibration.cpp) without an
 * extrinsic guess: image points normalised by cvUndistortPoints (x = (u - cx) * (1/fx), zero
 * distortion leaves them as they are), non-planar DLT initialisation (2n x 12 system, SVD of
 * L^T L, 3x3 part re-orthonormalised, t rescaled; at least 6 points), then CvLevMarq on
 * (rvec, tvec): 6 parameters, <= 20 iterations, epsilon FLT_EPSILON, J/err interface
 * (J^T J by cvMulTransposed, J^T e by cvGEMM, lambda = exp(k log 10), k from -3, the DIAGONAL of
 * J^T J scaled by 1 + lambda, solved by SVD).  X n x 3, uv n x 2 as doubles.
 * `init` (6 doubles) replaces the DLT when not NULL.  Returns 0; 1 when the object points are
 * planar (OpenCV starts from a homography there — NOT restated: the caller's fallback model is
 * refined instead); 2 when there are fewer than 6 points for the DLT (OpenCV >= 4.3 throws).
 * ---------------------------------------------------------------------------------------- */
static double norm_l2(const double* a, int n) {
    double s = 0;
    for (int i = 0; i < n; ++i) s += a[i] * a[i];
    return sqrt(s);
}

int orc_pnp_dlt_init(const double* X, const double* uv, int64_t n, const double* K, double* rvec, double* tvec) {
    const double ifx = 1. / K[0], ify = 1. / K[4];
    double Mc[3] = {0, 0, 0}, MM[9], W[3];
    for (int64_t i = 0; i < n; ++i)
        for (int j = 0; j < 3; ++j) Mc[j] += X[3 * i + j];
    for (int j = 0; j < 3; ++j) Mc[j] /= (double)n; /* cvAvg */
    for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 3; ++b) {
            double s = 0;
            for (int64_t i = 0; i < n; ++i) s += (X[3 * i + a] - Mc[a]) * (X[3 * i + b] - Mc[b]);
            MM[3 * a + b] = s;
        }
    orc_svd(MM, 3, 3, 0, W, NULL, NULL);
    if (W[2] / W[1] < 1e-3) return 1;
    if (n < 6) return 2;
    double LL[144];
    {
        double* L = (double*)malloc(sizeof(double) * 24 * (size_t)n);
        for (int64_t i = 0; i < n; ++i) {
            double* l = L + 24 * i;
            const double mx = (uv[2 * i] - K[2]) * ifx, my = (uv[2 * i + 1] - K[5]) * ify;
            const double x = -mx, y = -my;
            l[0] = l[16] = X[3 * i]; l[1] = l[17] = X[3 * i + 1]; l[2] = l[18] = X[3 * i + 2]; l[3] = l[19] = 1.;
            l[4] = l[5] = l[6] = l[7] = 0.;
            l[12] = l[13] = l[14] = l[15] = 0.;
            l[8] = x * X[3 * i]; l[9] = x * X[3 * i + 1]; l[10] = x * X[3 * i + 2]; l[11] = x;
            l[20] = y * X[3 * i]; l[21] = y * X[3 * i + 1]; l[22] = y * X[3 * i + 2]; l[23] = y;
        }
        for (int a = 0; a < 12; ++a)
            for (int b = 0; b < 12; ++b) {
                double s = 0;
                for (int64_t r = 0; r < 2 * n; ++r) s += L[12 * r + a] * L[12 * r + b];
                LL[12 * a + b] = s;
            }
        free(L);
    }
    double LW[12], LV[144];
    orc_svd(LL, 12, 12, 0, LW, NULL, LV);
    double RRt[12];
    memcpy(RRt, LV + 11 * 12, sizeof(RRt));
    double RR[9], tt[3];
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) RR[3 * i + j] = RRt[4 * i + j];
        tt[i] = RRt[4 * i + 3];
    }
    if (det3(RR) < 0) {
        for (int k = 0; k < 9; ++k) RR[k] *= -1;
        for (int k = 0; k < 3; ++k) tt[k] *= -1;
    }
    const double sc = norm_l2(RR, 9);
    if (!(fabs(sc) > DBL_EPSILON)) return 2;
    double w3[3], U[9], Vt[9], R[9];
    orc_svd(RR, 3, 3, 0, w3, U, Vt);
    mat3_mul(U, Vt, R);
    const double f = norm_l2(R, 9) / sc;
    for (int k = 0; k < 3; ++k) tvec[k] = tt[k] * f;
    orc_rodrigues_mat2vec(R, rvec);
    return 0;
}

/* The 28 sums of one Levenberg-Marquardt sweep — upper triangle of J^T J (21), J^T e (6), |e|^2 — in ONE FIXED TREE, the same
 * in this file and in the HIP library (its pnp_sweep_kernel / pnp_sweep_fold_kernel; docs/oracle.md).  OpenCV forms them with cvMulTransposed /
 * cvGEMM / cvNorm, whose SIMD summation order is not pinned by anything in the reference; a plain sequential sum here and a
 * parallel one on the device would differ in the last bits, LM's accept / reject test (errNorm > prevErrNorm) would now and
 * then go the other way, and because every camera is registered against points triangulated from the earlier ones the two
 * 57-camera chains would drift apart by a factor ~2.5 per frame.  With one tree both sides are bit-identical:
 *   term of point o:   Ju[a] Ju[b] + Jv[a] Jv[b]   |   Ju[a] ru + Jv[a] rv   |   ru ru + rv rv      (u-row product first)
 *   slots:             G = min(ceil(n / 1024), 64) groups of 1024 slots; point o goes to slot o mod 1024 G and a slot adds its
 *                      points in increasing o, starting from 0
 *   a group:           16 runs of 64 consecutive slots; a run is folded by the butterfly  v[l] += v[l ^ s], s = 32, 16, .. 1
 *                      (all 64 values at once; the result is read at l = 0); the group's sum = 0 + run 0 + run 1 + ... + run 15
 *   total:             G = 1: the group's sum;  G > 1: 0 + group 0 + group 1 + ...                                          */
#define LM_SLOTS 1024
#define LM_MAXG 64
/* An INDEPENDENT evaluation of the same 28 sums, for the tests only (orc_set_lm_sum_mode(1)): one long-double accumulator per
 * sum, points in index order — no slots, no butterflies, nothing shared with the device's tree.  tests/test_oracle_solvers.py
 * holds the tree to it within 1e-12 relative and the Levenberg-Marquardt result to the one this plain sum gives: agreement
 * between the HIP library and the oracle is then not "by construction of one shared tree" alone. */
static int g_lm_sum_mode = 0;
void orc_set_lm_sum_mode(int mode) { g_lm_sum_mode = mode; }
static void lm_plain_sums(const double* J, const double* err, int64_t n, double* out) {
    long double a[28];
    for (int k = 0; k < 28; ++k) a[k] = 0.0L;
    for (int64_t o = 0; o < n; ++o) {
        const long double ru = err[2 * o], rv = err[2 * o + 1];
        a[27] += ru * ru + rv * rv;
        if (J) {
            const double* ju = J + 12 * o;
            const double* jv = ju + 6;
            int q = 0;
            for (int i = 0; i < 6; ++i)
                for (int j = i; j < 6; ++j) a[q++] += (long double)ju[i] * ju[j] + (long double)jv[i] * jv[j];
            for (int i = 0; i < 6; ++i) a[21 + i] += (long double)ju[i] * ru + (long double)jv[i] * rv;
        }
    }
    for (int k = J ? 0 : 27; k < 28; ++k) out[k] = (double)a[k];
}
static void lm_tree_sums(const double* J /*2n x 6 or NULL*/, const double* err /*2n*/, int64_t n, double* out /*28*/) {
    if (g_lm_sum_mode == 1) { lm_plain_sums(J, err, n, out); return; }
    int64_t G = (n + LM_SLOTS - 1) / LM_SLOTS;
    if (G > LM_MAXG) G = LM_MAXG;
    if (G < 1) G = 1;
    const int first = J ? 0 : 27;
    double* acc = (double*)calloc((size_t)(G * LM_SLOTS) * 28, sizeof(double));
    for (int64_t o = 0; o < n; ++o) {
        double* a = acc + 28 * (o % (G * LM_SLOTS));
        const double ru = err[2 * o], rv = err[2 * o + 1];
        a[27] += ru * ru + rv * rv;
        if (J) {
            const double* ju = J + 12 * o;
            const double* jv = ju + 6;
            int q = 0;
            for (int i = 0; i < 6; ++i)
                for (int j = i; j < 6; ++j) a[q++] += ju[i] * ju[j] + jv[i] * jv[j];
            for (int i = 0; i < 6; ++i) a[21 + i] += ju[i] * ru + jv[i] * rv;
        }
    }
    for (int k = first; k < 28; ++k) {
        double total = 0;
        for (int64_t g = 0; g < G; ++g) {
            double grp = 0;
            for (int w = 0; w < LM_SLOTS / 64; ++w) {
                double v[64], t[64];
                for (int l = 0; l < 64; ++l) v[l] = acc[28 * (g * LM_SLOTS + 64 * w + l) + k];
                for (int sft = 32; sft >= 1; sft >>= 1) {
                    for (int l = 0; l < 64; ++l) t[l] = v[l] + v[l ^ sft];
                    memcpy(v, t, sizeof(v));
                }
                grp += v[0];
            }
            if (G == 1) total = grp;
            else total += grp;
        }
        out[k] = total;
    }
    free(acc);
}

/* tests: the 28 sums of J (2n x 6) and err (2n) by the fixed tree (mode 0) or by the plain long-double sum (mode 1) */
void orc_lm_sums(const double* J, const double* err, int64_t n, int mode, double* out28) {
    const int keep = g_lm_sum_mode;
    g_lm_sum_mode = mode;
    lm_tree_sums(J, err, n, out28);
    g_lm_sum_mode = keep;
}

int orc_levmarq_pose(const double* X, const double* uv, int64_t n, const double* K, double* rvec, double* tvec, int* iters_out) {
    const int max_iter = 20;
    const double epsilon = FLT_EPSILON, LOG10 = log(10.);
    double param[6], prev[6], JtJ[36], JtErr[6], sums[28];
    double* J = (double*)malloc(sizeof(double) * 12 * (size_t)n);
    double* err = (double*)malloc(sizeof(double) * 2 * (size_t)n);
    memcpy(param, rvec, 3 * sizeof(double));
    memcpy(param + 3, tvec, 3 * sizeof(double));
    int lambdaLg10 = -3, iters = 0;
    double prevErrNorm = DBL_MAX, errNorm = 0;
    /* CvLevMarq::step(): diagonal * (1 + lambda), SVD solve, param = prevParam - delta */
#define LM_STEP()                                                                    \
    do {                                                                             \
        const double lambda = exp(lambdaLg10 * LOG10);                               \
        double A[36], dx[6];                                                         \
        memcpy(A, JtJ, sizeof(A));                                                   \
        for (int i_ = 0; i_ < 6; ++i_) A[7 * i_] *= 1. + lambda;                     \
        svd_solve(A, 6, 6, JtErr, dx);                                               \
        for (int i_ = 0; i_ < 6; ++i_) param[i_] = prev[i_] - dx[i_];                \
    } while (0)
#define LM_ERR()                                                                     \
    do {                                                                             \
        project_jac(param, param + 3, K, X, n, err, NULL);                           \
        for (int64_t i_ = 0; i_ < 2 * n; ++i_) err[i_] = err[i_] - uv[i_];           \
        lm_tree_sums(NULL, err, n, sums);                                            \
    } while (0)
    for (;;) {
        /* STARTED / CALC_J: Jacobian and error at `param` */
        project_jac(param, param + 3, K, X, n, err, J);
        for (int64_t i = 0; i < 2 * n; ++i) err[i] = err[i] - uv[i];
        lm_tree_sums(J, err, n, sums);
        {
            int q = 0;
            for (int a = 0; a < 6; ++a)
                for (int b = a; b < 6; ++b) JtJ[6 * a + b] = JtJ[6 * b + a] = sums[q++];
            for (int a = 0; a < 6; ++a) JtErr[a] = sums[21 + a];
        }
        memcpy(prev, param, sizeof(prev));
        if (iters == 0) prevErrNorm = sqrt(sums[27]);
        LM_STEP();
        LM_ERR();
        /* CHECK_ERR */
        for (;;) {
            errNorm = sqrt(sums[27]);
            if (errNorm > prevErrNorm) {
                if (++lambdaLg10 <= 16) {
                    LM_STEP();
                    LM_ERR();
                    continue;
                }
            }
            break;
        }
        lambdaLg10 = lambdaLg10 - 1 > -16 ? lambdaLg10 - 1 : -16;
        double dn = 0, pn = 0;
        for (int i = 0; i < 6; ++i) {
            dn += (param[i] - prev[i]) * (param[i] - prev[i]);
            pn += prev[i] * prev[i];
        }
        ++iters;
        if (iters >= max_iter || sqrt(dn) / sqrt(pn) < epsilon) break; /* cvNorm(param, prevParam, CV_RELATIVE_L2) */
        prevErrNorm = errNorm;
    }
#undef LM_STEP
#undef LM_ERR
    memcpy(rvec, param, 3 * sizeof(double));
    memcpy(tvec, param + 3, 3 * sizeof(double));
    if (iters_out) *iters_out = iters;
    free(J);
    free(err);
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * solvePnP(SOLVEPNP_P3P) — what solvePnPRansac runs when it is given exactly FOUR points
 * (model_points = npoints = 4: no RANSAC, every point an inlier).  OpenCV's p3p.cpp solves the
 * perspective-three-point problem on the first three correspondences (Gao et al. 2003: a
 * quartic from the law of cosines, then an absolute orientation) and keeps, of its up to four
 * poses, the one that reprojects the FOURTH point best.  That source is not available here
 * (un-vendored third party), so this restatement takes the classical route to the same quartic
 * problem — Grunert's elimination in Haralick et al.'s notation (IJCV 1994): with
 * a = |P2 P3|, b = |P1 P3|, c = |P1 P2| and the cosines of the angles between the bearings,
 * s2 = u s1, s3 = v s1, v is a root of A4 v^4 + ... + A0 — solved with solvePoly, the pose from
 * the two orthonormal frames of the triangle, and OpenCV's selection rule (squared pixel error
 * of point 4).  Every minimal solver returns the roots of the same system; which formula
 * produced them shows in the last digits only.  Returns 1 with R (row-major), t, or 0.
 * ---------------------------------------------------------------------------------------- */
static void p3p_frame(const double* Q /*3 points x 3*/, double* F /*3x3 row-major, columns e1 e2 e3*/) {
    double e1[3], w[3], e3[3], e2[3];
    for (int k = 0; k < 3; ++k) { e1[k] = Q[3 + k] - Q[k]; w[k] = Q[6 + k] - Q[k]; }
    double n1 = sqrt(e1[0] * e1[0] + e1[1] * e1[1] + e1[2] * e1[2]);
    for (int k = 0; k < 3; ++k) e1[k] = e1[k] / n1;
    e3[0] = e1[1] * w[2] - e1[2] * w[1];
    e3[1] = e1[2] * w[0] - e1[0] * w[2];
    e3[2] = e1[0] * w[1] - e1[1] * w[0];
    double n3 = sqrt(e3[0] * e3[0] + e3[1] * e3[1] + e3[2] * e3[2]);
    for (int k = 0; k < 3; ++k) e3[k] = e3[k] / n3;
    e2[0] = e3[1] * e1[2] - e3[2] * e1[1];
    e2[1] = e3[2] * e1[0] - e3[0] * e1[2];
    e2[2] = e3[0] * e1[1] - e3[1] * e1[0];
    for (int k = 0; k < 3; ++k) { F[3 * k] = e1[k]; F[3 * k + 1] = e2[k]; F[3 * k + 2] = e3[k]; }
}

int orc_p3p(const double* K, const double* Xw /*4 x 3*/, const double* uv /*4 x 2 pixels*/, double* Rout, double* tout) {
    const double fx = K[0], fy = K[4], cx = K[2], cy = K[5];
    double f[3][3];
    for (int i = 0; i < 3; ++i) {
        const double x = (uv[2 * i] - cx) / fx, y = (uv[2 * i + 1] - cy) / fy;
        const double nrm = sqrt(x * x + y * y + 1.0);
        f[i][0] = x / nrm; f[i][1] = y / nrm; f[i][2] = 1.0 / nrm;
    }
    double d12[3], d13[3], d23[3];
    for (int k = 0; k < 3; ++k) { d23[k] = Xw[3 + k] - Xw[6 + k]; d13[k] = Xw[k] - Xw[6 + k]; d12[k] = Xw[k] - Xw[3 + k]; }
    const double a2 = d23[0] * d23[0] + d23[1] * d23[1] + d23[2] * d23[2];
    const double b2 = d13[0] * d13[0] + d13[1] * d13[1] + d13[2] * d13[2];
    const double c2 = d12[0] * d12[0] + d12[1] * d12[1] + d12[2] * d12[2];
    if (!(a2 > 0) || !(b2 > 0) || !(c2 > 0)) return 0;
    const double ca = f[1][0] * f[2][0] + f[1][1] * f[2][1] + f[1][2] * f[2][2];
    const double cb = f[0][0] * f[2][0] + f[0][1] * f[2][1] + f[0][2] * f[2][2];
    const double cg = f[0][0] * f[1][0] + f[0][1] * f[1][1] + f[0][2] * f[1][2];
    const double k1 = (a2 - c2) / b2, k2 = (a2 + c2) / b2, k3 = (b2 - c2) / b2, k4 = (b2 - a2) / b2;
    const double cab = c2 / b2, aab = a2 / b2;
    double A[5];
    A[4] = (k1 - 1) * (k1 - 1) - 4 * cab * ca * ca;
    A[3] = 4 * (k1 * (1 - k1) * cb - (1 - k2) * ca * cg + 2 * cab * ca * ca * cb);
    A[2] = 2 * (k1 * k1 - 1 + 2 * k1 * k1 * cb * cb + 2 * k3 * ca * ca - 4 * k2 * ca * cb * cg + 2 * k4 * cg * cg);
    A[1] = 4 * (-k1 * (1 + k1) * cb + 2 * aab * cg * cg * cb - (1 - k2) * ca * cg);
    A[0] = (1 + k1) * (1 + k1) - 4 * aab * cg * cg;
    double rre[4], rim[4];
    const int nr = orc_solve_poly(A, 4, rre, rim, 300);
    int found = 0;
    double best = 0;
    for (int r = 0; r < nr; ++r) {
        const double v = rre[r];
        if (!(fabs(rim[r]) <= 1e-9 * (1.0 + fabs(v))) || !(v > 0)) continue;
        const double den = 2 * (cg - v * ca);
        if (den == 0) continue;
        const double u = ((k1 - 1) * v * v - 2 * k1 * cb * v + 1 + k1) / den;
        if (!(u > 0)) continue;
        const double q = 1 + v * v - 2 * v * cb;
        if (!(q > 0)) continue;
        const double s1 = sqrt(b2 / q), s2 = u * s1, s3 = v * s1;
        double Y[9], FY[9], FP[9], R[9], t[3];
        for (int k = 0; k < 3; ++k) { Y[k] = s1 * f[0][k]; Y[3 + k] = s2 * f[1][k]; Y[6 + k] = s3 * f[2][k]; }
        p3p_frame(Y, FY);
        p3p_frame(Xw, FP);
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) R[3 * i + j] = FY[3 * i] * FP[3 * j] + FY[3 * i + 1] * FP[3 * j + 1] + FY[3 * i + 2] * FP[3 * j + 2];
        for (int i = 0; i < 3; ++i) t[i] = Y[i] - (R[3 * i] * Xw[0] + R[3 * i + 1] * Xw[1] + R[3 * i + 2] * Xw[2]);
        int finite = 1;
        for (int k = 0; k < 9; ++k) finite = finite && isfinite(R[k]);
        for (int k = 0; k < 3; ++k) finite = finite && isfinite(t[k]);
        if (!finite) continue;
        const double* P4 = Xw + 9;
        const double x4 = R[0] * P4[0] + R[1] * P4[1] + R[2] * P4[2] + t[0];
        const double y4 = R[3] * P4[0] + R[4] * P4[1] + R[5] * P4[2] + t[1];
        const double z4 = R[6] * P4[0] + R[7] * P4[1] + R[8] * P4[2] + t[2];
        const double eu = fx * (x4 / z4) + cx - uv[6], ev = fy * (y4 / z4) + cy - uv[7];
        const double err = eu * eu + ev * ev;
        if (!found || err < best) {
            best = err;
            found = 1;
            memcpy(Rout, R, sizeof(R));
            memcpy(tout, t, sizeof(t));
        }
    }
    return found;
}

/* ------------------------------------------------------------------------------------------
 * cv2.solvePnPRansac(objectPoints, imagePoints, K, distCoeffs = zeros(5,1), <rvec slot>)  sfm.py:67
 * (calib3d/solvepnp.cpp) with every tunable at its default: iterationsCount 100,
 * reprojectionError 8.0, confidence 0.99, flags ITERATIVE.  model_points = 5, minimal solver
 * EPnP on the sample: solvePnP(EPNP) first runs undistortPoints, whose OUTPUT TYPE is that of
 * the image points — float32 — and epnp::init_points maps it back with u = x*fu + uc, so the
 * sample's pixels make a round trip through float32 normalised coordinates.  Model = (rvec,
 * tvec) via Rodrigues; error = squared pixel distance of the float32 projection, as float, <=
 * (float)(8*8).  After RANSAC the inliers (as doubles) go through solvePnP(ITERATIVE); the
 * returned inliers are those of the best RANSAC model.
 * n == 4 is OpenCV's P3P branch (orc_p3p above); n < 4 is an assertion there and returns -1 here.
 * Returns 1 on success, 0 if RANSAC found no model.  status_out (optional): 0 DLT init,
 * 1 planar fallback, 2 too few inliers for the DLT (RANSAC model refined instead).
 * ---------------------------------------------------------------------------------------- */
int orc_solve_pnp_ransac(const float* X, const float* uv, int64_t n, const double* K, int iterations, float reproj_error,
                         double confidence, double* rvec, double* tvec, int32_t* inliers, int64_t* n_inliers,
                         double* ransac_model, int* status_out) {
    if (n_inliers) *n_inliers = 0;
    if (status_out) *status_out = 0;
    if (n < 4) return -1;
    const double ifx = 1. / K[0], ify = 1. / K[4];
    if (n == 4) {
        /* model_points == npoints == 4: a plain solvePnP(P3P), every point an inlier */
        double Xs[12], us[8], R[9], t[3];
        for (int k = 0; k < 4; ++k) {
            for (int j = 0; j < 3; ++j) Xs[3 * k + j] = (double)X[3 * k + j];
            us[2 * k] = (double)(float)(((double)uv[2 * k] - K[2]) * ifx) * K[0] + K[2];
            us[2 * k + 1] = (double)(float)(((double)uv[2 * k + 1] - K[5]) * ify) * K[4] + K[5];
        }
        const int ok = orc_p3p(K, Xs, us, R, t);
        if (ok) {
            orc_rodrigues_mat2vec(R, rvec);
            memcpy(tvec, t, sizeof(t));
            for (int k = 0; k < 4; ++k) inliers[k] = k;
            if (n_inliers) *n_inliers = 4;
            if (ransac_model) { memcpy(ransac_model, rvec, 24); memcpy(ransac_model + 3, tvec, 24); }
        }
        return ok;
    }
    uint8_t* cur = (uint8_t*)malloc((size_t)n);
    uint8_t* bestmask = (uint8_t*)malloc((size_t)n);
    double best_model[6] = {0, 0, 0, 0, 0, 0};
    int best = 0;
    const float thr2 = (float)((double)reproj_error * (double)reproj_error);
    if (n == 5) {
        /* model_points == npoints: a plain solvePnP(EPNP) on all five, every point an inlier */
        double Xs[15], us[10], R[9], t[3];
        for (int k = 0; k < 5; ++k) {
            for (int j = 0; j < 3; ++j) Xs[3 * k + j] = (double)X[3 * k + j];
            us[2 * k] = (double)(float)(((double)uv[2 * k] - K[2]) * ifx) * K[0] + K[2];
            us[2 * k + 1] = (double)(float)(((double)uv[2 * k + 1] - K[5]) * ify) * K[4] + K[5];
        }
        int ok = orc_epnp(K, Xs, us, 5, R, t) == 0;
        if (ok) {
            orc_rodrigues_mat2vec(R, rvec);
            memcpy(tvec, t, sizeof(t));
            for (int k = 0; k < 5; ++k) inliers[k] = k;
            if (n_inliers) *n_inliers = 5;
            if (ransac_model) { memcpy(ransac_model, rvec, 24); memcpy(ransac_model + 3, tvec, 24); }
        }
        free(cur); free(bestmask);
        return ok;
    }
    uint64_t rng = ~(uint64_t)0;
    int niters = iterations > 1 ? iterations : 1;
    for (int iter = 0; iter < niters; ++iter) {
        int idx[5];
        double Xs[15], us[10], R[9], t[3], model[6];
        get_subset(&rng, (int)n, 5, idx);
        for (int k = 0; k < 5; ++k) {
            for (int j = 0; j < 3; ++j) Xs[3 * k + j] = (double)X[3 * idx[k] + j];
            us[2 * k] = (double)(float)(((double)uv[2 * idx[k]] - K[2]) * ifx) * K[0] + K[2];
            us[2 * k + 1] = (double)(float)(((double)uv[2 * idx[k] + 1] - K[5]) * ify) * K[4] + K[5];
        }
        if (orc_epnp(K, Xs, us, 5, R, t) != 0) continue;
        int finite = 1;
        for (int k = 0; k < 9; ++k) finite &= isfinite(R[k]) != 0;
        for (int k = 0; k < 3; ++k) finite &= isfinite(t[k]) != 0;
        if (!finite) continue; /* (a NaN model scores zero inliers in OpenCV: same outcome) */
        orc_rodrigues_mat2vec(R, model);
        memcpy(model + 3, t, sizeof(t));
        int32_t good;
        orc_score_pnp(model, 1, K, X, uv, n, thr2, &good, cur);
        if (good > (best > 4 ? best : 4)) {
            memcpy(bestmask, cur, (size_t)n);
            memcpy(best_model, model, sizeof(model));
            best = good;
            niters = orc_ransac_update_num_iters(confidence, (double)(n - good) / (double)n, 5, niters);
        }
    }
    if (best <= 0) {
        free(cur); free(bestmask);
        return 0;
    }
    if (ransac_model) memcpy(ransac_model, best_model, sizeof(best_model));
    double* Xi = (double*)malloc(sizeof(double) * 3 * (size_t)best);
    double* ui = (double*)malloc(sizeof(double) * 2 * (size_t)best);
    int64_t m = 0;
    for (int64_t i = 0; i < n; ++i)
        if (bestmask[i]) {
            for (int j = 0; j < 3; ++j) Xi[3 * m + j] = (double)X[3 * i + j];
            ui[2 * m] = (double)uv[2 * i];
            ui[2 * m + 1] = (double)uv[2 * i + 1];
            inliers[m] = (int32_t)i;
            ++m;
        }
    if (n_inliers) *n_inliers = m;
    double r0[3], t0[3];
    const int st = orc_pnp_dlt_init(Xi, ui, m, K, r0, t0);
    if (st != 0) { /* planar / too few points: refine the RANSAC model (see orc_pnp_dlt_init) */
        memcpy(r0, best_model, sizeof(r0));
        memcpy(t0, best_model + 3, sizeof(t0));
    }
    if (status_out) *status_out = st;
    orc_levmarq_pose(Xi, ui, m, K, r0, t0, NULL);
    memcpy(rvec, r0, sizeof(r0));
    memcpy(tvec, t0, sizeof(t0));
    free(Xi); free(ui); free(cur); free(bestmask);
    return 1;
}
