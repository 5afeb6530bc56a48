// Host-side hypothesis generators of the RANSAC entry points (cv2.findEssentialMat sfm.py:307, cv2.recoverPose
// sfm.py:311, cv2.solvePnPRansac sfm.py:67).  OpenCV keeps hypothesis GENERATION sequential on the host — one
// five-point / EPnP solve on five correspondences per RANSAC iteration — and so does this library: what scales with the
// data (scoring every correspondence against every hypothesis, the cheirality vote, the Levenberg-Marquardt sweeps)
// runs in the HIP kernels of ransac.hip / residual.hip; this header is the generation side, plain C++ on a few dozen
// doubles.  Each routine follows the OpenCV routine named at its head operation for operation (cv::SVD's one-sided
// Jacobi with its sweep order and thresholds, cv::solvePoly's Durand-Kerner start values, epnp.cpp's routines), because
// the minimal solvers are not well-conditioned functions of their input: EPnP on five points has a two-dimensional
// null space whose basis — and with it the returned pose — depends on the eigen-solver's rotation order, and the order
// of the five-point models decides RANSAC ties.  Small fixed-capacity matrices, no heap, no device code.
#pragma once
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstring>

namespace sfm {
namespace host {

// cv::RNG — multiply-with-carry generator; RANSACPointSetRegistrator seeds it with 2^64 - 1.
struct CvRng {
    uint64_t state;
    explicit CvRng(uint64_t s = ~uint64_t(0)) : state(s) {}
    uint32_t next() {
        state = (uint64_t)(uint32_t)state * 4164903690u + (uint32_t)(state >> 32);
        return (uint32_t)state;
    }
    int uniform(int a, int b) { return a == b ? a : (int)(next() % (uint32_t)(b - a) + (uint32_t)a); }
    // getSubset: `count` distinct indices, a duplicate is redrawn in place
    void subset(int count, int model_points, int* idx) {
        for (int i = 0; i < model_points; ++i) {
            for (;;) {
                const int v = uniform(0, count);
                bool dup = false;
                for (int j = 0; j < i; ++j) dup = dup || idx[j] == v;
                if (!dup) {
                    idx[i] = v;
                    break;
                }
            }
        }
    }
};

// RANSACUpdateNumIters (ptsetreg.cpp)
inline int ransac_update_num_iters(double p, double ep, int model_points, int max_iters) {
    p = std::fmax(std::fmin(p, 1.), 0.);
    ep = std::fmax(std::fmin(ep, 1.), 0.);
    double num = std::fmax(1. - p, DBL_MIN);
    double denom = 1. - std::pow(1. - ep, model_points);
    if (denom < DBL_MIN) return 0;
    num = std::log(num);
    denom = std::log(denom);
    return denom >= 0 || -num >= max_iters * (-denom) ? max_iters : (int)std::lrint(num / denom);
}

// ------------------------------------------------------------------------------------------------ cv::SVD
// One-sided (Hestenes) Jacobi on the rows of `a` (= the columns of the decomposed matrix), OpenCV's JacobiSVDImpl_:
// cyclic pairs (i < j), pair skipped when |p| <= 10 eps sqrt(a b), at most max(m, 30) sweeps, descending selection sort.
// Dot products / rotated norms in the two-lane order of the 128-bit double SIMD path for m >= 4.
constexpr int kSvdMaxN = 12, kSvdMaxM = 24;

inline double hypot_cv(double a, double b) {
    a = std::fabs(a);
    b = std::fabs(b);
    if (a > b) {
        b /= a;
        return a * std::sqrt(1 + b * b);
    }
    if (b > 0) {
        a /= b;
        return b * std::sqrt(1 + a * a);
    }
    return 0;
}

struct Svd {
    int m = 0, n = 0, ucols = 0, vrows = 0;
    double w[kSvdMaxN];
    double u[kSvdMaxM * kSvdMaxM];    // m x ucols, row-major
    double vt[kSvdMaxM * kSvdMaxM];   // vrows x n, row-major

    static void jacobi(double* a, int mm, int nn, double* W, double* v) {
        const double eps = DBL_EPSILON * 10;
        const int max_iter = mm > 30 ? mm : 30;
        auto dot2 = [mm](const double* x, const double* y) {
            double p = 0;
            int k = 0;
            if (mm >= 4) {
                double s0 = 0, s1 = 0;
                for (; k <= mm - 2; k += 2) {
                    s0 = s0 + x[k] * y[k];
                    s1 = s1 + x[k + 1] * y[k + 1];
                }
                p = s0 + s1;
            }
            for (; k < mm; ++k) p += x[k] * y[k];
            return p;
        };
        for (int i = 0; i < nn; ++i) {
            double sd = 0;
            for (int k = 0; k < mm; ++k) sd += a[i * mm + k] * a[i * mm + k];
            W[i] = sd;
            for (int k = 0; k < nn; ++k) v[i * nn + k] = k == i ? 1.0 : 0.0;
        }
        for (int iter = 0; iter < max_iter; ++iter) {
            bool changed = false;
            for (int i = 0; i < nn - 1; ++i)
                for (int j = i + 1; j < nn; ++j) {
                    double *Ai = a + i * mm, *Aj = a + j * mm;
                    double aa = W[i], bb = W[j], p = dot2(Ai, Aj);
                    if (std::fabs(p) <= eps * std::sqrt(aa * bb)) continue;
                    p *= 2;
                    const double beta = aa - bb, gamma = hypot_cv(p, beta);
                    double c, s;
                    if (beta < 0) {
                        const double delta = (gamma - beta) * 0.5;
                        s = std::sqrt(delta / gamma);
                        c = p / (gamma * s * 2);
                    } else {
                        c = std::sqrt((gamma + beta) / (gamma * 2));
                        s = p / (gamma * c * 2);
                    }
                    int k = 0;
                    aa = bb = 0;
                    if (mm >= 4) {
                        double a0 = 0, a1 = 0, b0 = 0, b1 = 0;
                        for (; k <= mm - 2; k += 2) {
                            const double t0 = c * Ai[k] + s * Aj[k], t1 = c * Aj[k] - s * Ai[k];
                            const double u0 = c * Ai[k + 1] + s * Aj[k + 1], u1 = c * Aj[k + 1] - s * Ai[k + 1];
                            Ai[k] = t0; Aj[k] = t1; Ai[k + 1] = u0; Aj[k + 1] = u1;
                            a0 = a0 + t0 * t0; b0 = b0 + t1 * t1;
                            a1 = a1 + u0 * u0; b1 = b1 + u1 * u1;
                        }
                        aa = a0 + a1;
                        bb = b0 + b1;
                    }
                    for (; k < mm; ++k) {
                        const double t0 = c * Ai[k] + s * Aj[k], t1 = -s * Ai[k] + c * Aj[k];
                        Ai[k] = t0; Aj[k] = t1;
                        aa += t0 * t0;
                        bb += t1 * t1;
                    }
                    W[i] = aa;
                    W[j] = bb;
                    changed = true;
                    double *Vi = v + i * nn, *Vj = v + j * nn;
                    for (k = 0; k < nn; ++k) {
                        const double t0 = c * Vi[k] + s * Vj[k], t1 = -s * Vi[k] + c * Vj[k];
                        Vi[k] = t0; Vj[k] = t1;
                    }
                }
            if (!changed) break;
        }
        for (int i = 0; i < nn; ++i) {
            double sd = 0;
            for (int k = 0; k < mm; ++k) sd += a[i * mm + k] * a[i * mm + k];
            W[i] = std::sqrt(sd);
        }
        for (int i = 0; i < nn - 1; ++i) {
            int j = i;
            for (int k = i + 1; k < nn; ++k)
                if (W[j] < W[k]) j = k;
            if (i != j) {
                std::swap(W[i], W[j]);
                for (int k = 0; k < mm; ++k) std::swap(a[i * mm + k], a[j * mm + k]);
                for (int k = 0; k < nn; ++k) std::swap(v[i * nn + k], v[j * nn + k]);
            }
        }
    }

    // cv::SVD::compute(A[, FULL_UV]); A is rows x cols, row-major.
    void compute(const double* A, int rows, int cols, bool full_uv = false) {
        m = rows;
        n = cols;
        const bool at = m < n;
        const int mm = at ? n : m, nn = at ? m : n;
        const int urows = full_uv ? mm : nn;
        double a[kSvdMaxM * kSvdMaxM], v[kSvdMaxN * kSvdMaxN], W[kSvdMaxN];
        std::memset(a, 0, sizeof(a));
        for (int i = 0; i < nn; ++i)
            for (int k = 0; k < mm; ++k) a[i * mm + k] = at ? A[i * n + k] : A[k * n + i];
        jacobi(a, mm, nn, W, v);
        // left vectors: normalised rows; rows of (numerically) zero singular value and the FULL_UV complement are
        // drawn from RNG(0x12345678) (+-1/m by bit 8), Gram-Schmidt'ed twice against the previous rows with an L1
        // renormalisation after every projection, then L2-normalised — this is what makes the five-point solver's
        // null-space basis a deterministic function of its input
        const double minval = DBL_MIN, eps = DBL_EPSILON * 10;
        CvRng rng(0x12345678u);
        for (int i = 0; i < urows; ++i) {
            double sd = i < nn ? W[i] : 0;
            for (int ii = 0; ii < 100 && sd <= minval; ++ii) {
                const double val0 = 1. / mm;
                for (int k = 0; k < mm; ++k) a[i * mm + k] = (rng.next() & 256) != 0 ? val0 : -val0;
                for (int iter = 0; iter < 2; ++iter)
                    for (int j = 0; j < i; ++j) {
                        sd = 0;
                        for (int k = 0; k < mm; ++k) sd += a[i * mm + k] * a[j * mm + k];
                        double asum = 0;
                        for (int k = 0; k < mm; ++k) {
                            const double t = a[i * mm + k] - sd * a[j * mm + k];
                            a[i * mm + k] = t;
                            asum += std::fabs(t);
                        }
                        asum = asum > eps * 100 ? 1 / asum : 0;
                        for (int k = 0; k < mm; ++k) a[i * mm + k] *= asum;
                    }
                sd = 0;
                for (int k = 0; k < mm; ++k) sd += a[i * mm + k] * a[i * mm + k];
                sd = std::sqrt(sd);
            }
            const double s = sd > minval ? 1 / sd : 0.;
            for (int k = 0; k < mm; ++k) a[i * mm + k] *= s;
        }
        for (int i = 0; i < nn; ++i) w[i] = W[i];
        if (!at) {
            ucols = urows;
            vrows = nn;
            for (int k = 0; k < mm; ++k)
                for (int i = 0; i < urows; ++i) u[k * urows + i] = a[i * mm + k];
            std::memcpy(vt, v, sizeof(double) * nn * nn);
        } else {
            ucols = nn;
            vrows = urows;
            for (int k = 0; k < nn; ++k)
                for (int i = 0; i < nn; ++i) u[k * nn + i] = v[i * nn + k];
            std::memcpy(vt, a, sizeof(double) * urows * mm);
        }
    }

    // cv::solve(A, b, DECOMP_SVD) back-substitution (SVBkSb, one right-hand side): singular values <= 2 eps sum(w) dropped
    void back_subst(const double* b, double* x) const {
        const int nm = m < n ? m : n;
        double threshold = 0;
        for (int i = 0; i < nm; ++i) threshold += w[i];
        threshold *= DBL_EPSILON * 2;
        for (int j = 0; j < n; ++j) x[j] = 0;
        for (int i = 0; i < nm; ++i) {
            double wi = w[i];
            if (std::fabs(wi) <= threshold) continue;
            wi = 1 / wi;
            double s = 0;
            for (int j = 0; j < m; ++j) s += u[j * ucols + i] * b[j];
            s *= wi;
            for (int j = 0; j < n; ++j) x[j] = x[j] + s * vt[i * n + j];
        }
    }
    // cv::invert(A, DECOMP_SVD) of a square matrix
    void inverse(double* inv) const {
        double threshold = 0, buf[kSvdMaxN];
        for (int i = 0; i < n; ++i) threshold += w[i];
        threshold *= DBL_EPSILON * 2;
        for (int j = 0; j < n * n; ++j) inv[j] = 0;
        for (int i = 0; i < n; ++i) {
            double wi = w[i];
            if (std::fabs(wi) <= threshold) continue;
            wi = 1 / wi;
            for (int j = 0; j < n; ++j) buf[j] = u[j * ucols + i] * wi;
            for (int r = 0; r < n; ++r)
                for (int j = 0; j < n; ++j) inv[r * n + j] += vt[i * n + r] * buf[j];
        }
    }
};

inline double det3(const double* M) {
    return M[0] * (M[4] * M[8] - M[5] * M[7]) - M[1] * (M[3] * M[8] - M[5] * M[6]) + M[2] * (M[3] * M[7] - M[4] * M[6]);
}
inline void mul3(const double* A, const double* B, double* C) {
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            double s = 0;
            for (int k = 0; k < 3; ++k) s += A[3 * i + k] * B[3 * k + j];
            C[3 * i + j] = s;
        }
}

// ------------------------------------------------------------------------------------------------ cv::Rodrigues
// matrix -> vector (calibration.cpp cvRodrigues2): orthonormalise by SVD, axis from the antisymmetric part, angle from
// the trace, the near-pi branch from the diagonal.
inline void rodrigues_mat2vec(const double* Rin, double* r) {
    Svd sv;
    sv.compute(Rin, 3, 3);
    double R[9];
    mul3(sv.u, sv.vt, R);
    double rx = R[7] - R[5], ry = R[2] - R[6], rz = R[3] - R[1];
    const double s = std::sqrt((rx * rx + ry * ry + rz * rz) * 0.25);
    double c = (R[0] + R[4] + R[8] - 1) * 0.5;
    c = c > 1. ? 1. : c < -1. ? -1. : c;
    double theta = std::acos(c);
    if (s < 1e-5) {
        if (c > 0) {
            rx = ry = rz = 0;
        } else {
            double t = (R[0] + 1) * 0.5;
            rx = std::sqrt(t > 0. ? t : 0.);
            t = (R[4] + 1) * 0.5;
            ry = std::sqrt(t > 0. ? t : 0.) * (R[1] < 0 ? -1. : 1.);
            t = (R[8] + 1) * 0.5;
            rz = std::sqrt(t > 0. ? t : 0.) * (R[2] < 0 ? -1. : 1.);
            if (std::fabs(rx) < std::fabs(ry) && std::fabs(rx) < std::fabs(rz) && (R[5] > 0) != (ry * rz > 0)) rz = -rz;
            theta /= std::sqrt(rx * rx + ry * ry + rz * rz);
            rx *= theta; ry *= theta; rz *= theta;
        }
    } else {
        double vth = 1 / (2 * s);
        vth *= theta;
        rx *= vth; ry *= vth; rz *= vth;
    }
    r[0] = rx; r[1] = ry; r[2] = rz;
}

// ------------------------------------------------------------------------------------------------ cv::solvePoly
// Durand-Kerner from the powers of (1 + i), <= max_iters sweeps, stop only when nothing moved.  c[k] multiplies x^k.
struct Cx {
    double re, im;
};
inline Cx operator*(Cx a, Cx b) { return {a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re}; }
inline Cx operator-(Cx a, Cx b) { return {a.re - b.re, a.im - b.im}; }
inline Cx operator+(Cx a, Cx b) { return {a.re + b.re, a.im + b.im}; }
inline Cx operator/(Cx a, Cx b) {
    const double t = 1. / (b.re * b.re + b.im * b.im);
    return {(a.re * b.re + a.im * b.im) * t, (-a.re * b.im + a.im * b.re) * t};
}

inline int solve_poly(const double* c, int deg, Cx* roots, int max_iters = 300) {
    Cx coeffs[16];
    int n = deg;
    for (int i = 0; i <= n; ++i) coeffs[i] = {c[i], 0};
    for (; n > 1; --n)
        if (std::fabs(coeffs[n].re) + std::fabs(coeffs[n].im) > DBL_EPSILON) break;
    Cx p{1, 0};
    const Cx r{1, 1};
    for (int i = 0; i < n; ++i) {
        roots[i] = p;
        p = p * r;
    }
    for (int iter = 0; iter < max_iters; ++iter) {
        double max_diff = 0;
        for (int i = 0; i < n; ++i) {
            p = roots[i];
            Cx num = coeffs[n], denom = coeffs[n];
            for (int j = 0; j < n; ++j) {
                num = num * p + coeffs[n - j - 1];
                if (j != i) denom = denom * (p - roots[j]);
            }
            num = num / denom;
            roots[i] = p - num;
            max_diff = std::fmax(max_diff, std::sqrt(num.re * num.re + num.im * num.im));
        }
        if (max_diff <= 0) break;
    }
    for (int i = 0; i < n; ++i)
        if (std::fabs(roots[i].im) < 1e-100) roots[i].im = 0;
    return n;
}

// ------------------------------------------------------------------------------------------------ five-point
// EMEstimatorCallback::runKernel (five-point.cpp): null space of the 5 x 9 epipolar system (full SVD), the ten cubic
// constraints on E = x E0 + y E1 + z E2 + E3 over Nister's monomial order, elimination of the first ten monomials,
// det B(z) = 0 (degree 10), (x, y) from the null vector of B(z).  Polynomials in (x, y, z) of total degree <= 3 are
// kept as 4 x 4 x 4 coefficient cubes.  Up to 10 unit-Frobenius-norm models, in solvePoly's root order.
struct Poly3 {
    double c[4][4][4];
    Poly3() { std::memset(c, 0, sizeof(c)); }
    static Poly3 linear(double cx, double cy, double cz, double c1) {
        Poly3 p;
        p.c[1][0][0] = cx; p.c[0][1][0] = cy; p.c[0][0][1] = cz; p.c[0][0][0] = c1;
        return p;
    }
    Poly3 operator*(const Poly3& b) const {
        Poly3 r;
        for (int i = 0; i < 4; ++i)
            for (int j = 0; i + j < 4; ++j)
                for (int k = 0; i + j + k < 4; ++k) {
                    const double av = c[i][j][k];
                    if (av == 0) continue;
                    for (int l = 0; i + l < 4; ++l)
                        for (int mm = 0; j + mm < 4; ++mm)
                            for (int o = 0; k + o < 4; ++o)
                                if (i + j + k + l + mm + o < 4) r.c[i + l][j + mm][k + o] += av * b.c[l][mm][o];
                }
        return r;
    }
    void axpy(double a, const Poly3& x) {
        const double* xs = &x.c[0][0][0];
        double* ys = &c[0][0][0];
        for (int i = 0; i < 64; ++i) ys[i] += a * xs[i];
    }
};

inline int five_point(const double* x1, const double* x2, double* E_out) {
    static const int mono[20][3] = {{3, 0, 0}, {0, 3, 0}, {2, 1, 0}, {1, 2, 0}, {2, 0, 1}, {2, 0, 0}, {0, 2, 1},
                                    {0, 2, 0}, {1, 1, 1}, {1, 1, 0}, {1, 0, 2}, {1, 0, 1}, {1, 0, 0}, {0, 1, 2},
                                    {0, 1, 1}, {0, 1, 0}, {0, 0, 3}, {0, 0, 2}, {0, 0, 1}, {0, 0, 0}};
    double Q[45];
    for (int i = 0; i < 5; ++i) {
        const double a1 = x1[2 * i], b1 = x1[2 * i + 1], a2 = x2[2 * i], b2 = x2[2 * i + 1];
        double* q = Q + 9 * i;
        q[0] = a1 * a2; q[1] = b1 * a2; q[2] = a2;
        q[3] = a1 * b2; q[4] = b1 * b2; q[5] = b2;
        q[6] = a1;      q[7] = b1;      q[8] = 1.0;
    }
    Svd sv;
    sv.compute(Q, 5, 9, true);
    const double* EE[4] = {sv.vt + 45, sv.vt + 54, sv.vt + 63, sv.vt + 72};
    Poly3 e[3][3];
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) e[r][c] = Poly3::linear(EE[0][3 * r + c], EE[1][3 * r + c], EE[2][3 * r + c], EE[3][3 * r + c]);
    Poly3 rows[10];
    {   // det(E) along the first row
        static const int cof[3][2] = {{1, 2}, {0, 2}, {0, 1}};
        for (int c = 0; c < 3; ++c) {
            Poly3 minor = e[1][cof[c][0]] * e[2][cof[c][1]];
            minor.axpy(-1.0, e[1][cof[c][1]] * e[2][cof[c][0]]);
            rows[0].axpy(c == 1 ? -1.0 : 1.0, e[0][c] * minor);
        }
    }
    Poly3 eet[3][3];
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c)
            for (int k = 0; k < 3; ++k) eet[r][c].axpy(1.0, e[r][k] * e[c][k]);
    Poly3 tr = eet[0][0];
    tr.axpy(1.0, eet[1][1]);
    tr.axpy(1.0, eet[2][2]);
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) {
            Poly3& out = rows[1 + 3 * r + c];
            for (int k = 0; k < 3; ++k) out.axpy(2.0, eet[r][k] * e[k][c]);
            out.axpy(-1.0, tr * e[r][c]);
        }
    double A[10][20];
    for (int r = 0; r < 10; ++r)
        for (int mm = 0; mm < 20; ++mm) A[r][mm] = rows[r].c[mono[mm][0]][mono[mm][1]][mono[mm][2]];
    for (int col = 0; col < 10; ++col) {          // A[:, :10]^-1 A[:, 10:] by LU with partial pivoting
        int piv = col;
        for (int r = col + 1; r < 10; ++r)
            if (std::fabs(A[r][col]) > std::fabs(A[piv][col])) piv = r;
        if (std::fabs(A[piv][col]) < DBL_EPSILON) return 0;
        if (piv != col)
            for (int mm = 0; mm < 20; ++mm) std::swap(A[col][mm], A[piv][mm]);
        const double d = 1 / A[col][col];
        for (int r = col + 1; r < 10; ++r) {
            const double f = A[r][col] * d;
            for (int mm = col; mm < 20; ++mm) A[r][mm] -= f * A[col][mm];
        }
    }
    for (int col = 9; col >= 0; --col) {
        const double d = 1 / A[col][col];
        for (int mm = 10; mm < 20; ++mm) {
            double s = A[col][mm];
            for (int k = col + 1; k < 10; ++k) s -= A[col][k] * A[k][mm];
            A[col][mm] = s * d;
        }
    }
    double b[3][13];                                 // B(z) rows: [x (z^3..1) | y (z^3..1) | 1 (z^4..1)]
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
    double pz[3][3][5];
    const int dg[3] = {3, 3, 4};
    for (int i = 0; i < 3; ++i) {
        for (int k = 0; k < 4; ++k) { pz[i][0][k] = b[i][3 - k]; pz[i][1][k] = b[i][7 - k]; }
        for (int k = 0; k < 5; ++k) pz[i][2][k] = b[i][12 - k];
    }
    auto up_mul = [](const double* a, int da, const double* bb, int db, double* out) {
        for (int i = 0; i <= da + db; ++i) out[i] = 0;
        for (int i = 0; i <= da; ++i)
            for (int j = 0; j <= db; ++j) out[i + j] += a[i] * bb[j];
    };
    double c[11] = {0};
    static const int perm[6][4] = {{0, 1, 2, 1}, {1, 2, 0, 1}, {2, 0, 1, 1}, {2, 1, 0, -1}, {1, 0, 2, -1}, {0, 2, 1, -1}};
    for (int p = 0; p < 6; ++p) {
        double t[11], uu[11];
        const int c0 = perm[p][0], c1 = perm[p][1], c2 = perm[p][2];
        up_mul(pz[0][c0], dg[c0], pz[1][c1], dg[c1], t);
        up_mul(t, dg[c0] + dg[c1], pz[2][c2], dg[c2], uu);
        for (int k = 0; k <= 10; ++k) c[k] += perm[p][3] * uu[k];
    }
    for (int k = 0; k < 11; ++k)
        if (!std::isfinite(c[k])) return 0;
    Cx roots[10];
    const int nroots = solve_poly(c, 10, roots);
    int count = 0;
    for (int i = 0; i < nroots; ++i) {
        if (std::fabs(roots[i].im) > 1e-10) continue;
        const double z1 = roots[i].re, z2 = z1 * z1, z3 = z2 * z1, z4 = z3 * z1;
        double bz[9];
        for (int j = 0; j < 3; ++j) {
            const double* br = b[j];
            bz[3 * j + 0] = br[0] * z3 + br[1] * z2 + br[2] * z1 + br[3];
            bz[3 * j + 1] = br[4] * z3 + br[5] * z2 + br[6] * z1 + br[7];
            bz[3 * j + 2] = br[8] * z4 + br[9] * z3 + br[10] * z2 + br[11] * z1 + br[12];
        }
        Svd sz;
        sz.compute(bz, 3, 3);
        const double* xy1 = sz.vt + 6;
        if (std::fabs(xy1[2]) < 1e-10) continue;
        const double x = xy1[0] / xy1[2], y = xy1[1] / xy1[2];
        double Ev[9], nrm = 0;
        for (int k = 0; k < 9; ++k) {
            Ev[k] = EE[0][k] * x + EE[1][k] * y + EE[2][k] * z1 + EE[3][k];
            nrm += Ev[k] * Ev[k];
        }
        nrm = std::sqrt(nrm);
        for (int k = 0; k < 9; ++k) E_out[9 * count + k] = Ev[k] / nrm;
        ++count;
    }
    return count;
}

// cv::decomposeEssentialMat: R1 = U W Vt, R2 = U W^T Vt, t = U[:, 2] (det-fixed SVD)
inline void decompose_essential(const double* E, double* R1, double* R2, double* t) {
    Svd sv;
    sv.compute(E, 3, 3);
    double U[9], Vt[9], T[9];
    std::memcpy(U, sv.u, sizeof(U));
    std::memcpy(Vt, sv.vt, sizeof(Vt));
    if (det3(U) < 0)
        for (double& v : U) v *= -1.;
    if (det3(Vt) < 0)
        for (double& v : Vt) v *= -1.;
    const double W[9] = {0, 1, 0, -1, 0, 0, 0, 0, 1}, Wt[9] = {0, -1, 0, 1, 0, 0, 0, 0, 1};
    mul3(U, W, T);
    mul3(T, Vt, R1);
    mul3(U, Wt, T);
    mul3(T, Vt, R2);
    t[0] = U[2] * 1.0; t[1] = U[5] * 1.0; t[2] = U[8] * 1.0;
}

// ------------------------------------------------------------------------------------------------ EPnP
// epnp.cpp (Lepetit, Moreno-Noguer, Fua): control points from the PCA of the sample, barycentric coordinates, the four
// smallest singular vectors of M^T M, three beta initialisations each refined by five Gauss-Newton steps (Householder
// QR, the authors' routine with its pivot-scan quirk), absolute orientation, smallest mean reprojection error wins.
constexpr int kEpnpMaxPts = 64;

class Epnp {
public:
    Epnp(const double* K, const double* Xw, const double* uv, int n) : n_(n), pws_(Xw), us_(uv) {
        fu_ = K[0]; fv_ = K[4]; uc_ = K[2]; vc_ = K[5];
    }
    void compute_pose(double* R_out, double* t_out) {
        choose_control_points();
        barycentric();
        double mtm[144], ut[144];
        {
            double M[2 * kEpnpMaxPts * 12];
            for (int i = 0; i < n_; ++i) {
                const double* as = alphas_ + 4 * i;
                double *M1 = M + 24 * i, *M2 = M1 + 12;
                for (int k = 0; k < 4; ++k) {
                    M1[3 * k] = as[k] * fu_; M1[3 * k + 1] = 0.0;         M1[3 * k + 2] = as[k] * (uc_ - us_[2 * i]);
                    M2[3 * k] = 0.0;         M2[3 * k + 1] = as[k] * fv_; M2[3 * k + 2] = as[k] * (vc_ - us_[2 * i + 1]);
                }
            }
            for (int a = 0; a < 12; ++a)
                for (int b = 0; b < 12; ++b) {
                    double s = 0;
                    for (int r = 0; r < 2 * n_; ++r) s += M[12 * r + a] * M[12 * r + b];
                    mtm[12 * a + b] = s;
                }
        }
        Svd sv;
        sv.compute(mtm, 12, 12);
        for (int i = 0; i < 12; ++i)
            for (int k = 0; k < 12; ++k) ut[12 * i + k] = sv.u[12 * k + i];
        double L[60], rho[6];
        build_L_rho(ut, L, rho);
        double betas[4][4], rep[4], Rs[4][3][3], ts[4][3];
        {   // approximation 1: [B11 B12 B13 B14]
            double l[24], b4[4];
            for (int i = 0; i < 6; ++i) { l[4 * i] = L[10 * i]; l[4 * i + 1] = L[10 * i + 1]; l[4 * i + 2] = L[10 * i + 3]; l[4 * i + 3] = L[10 * i + 6]; }
            Svd s;
            s.compute(l, 6, 4);
            s.back_subst(rho, b4);
            double* be = betas[1];
            if (b4[0] < 0) {
                be[0] = std::sqrt(-b4[0]); be[1] = -b4[1] / be[0]; be[2] = -b4[2] / be[0]; be[3] = -b4[3] / be[0];
            } else {
                be[0] = std::sqrt(b4[0]); be[1] = b4[1] / be[0]; be[2] = b4[2] / be[0]; be[3] = b4[3] / be[0];
            }
        }
        gauss_newton(L, rho, betas[1]);
        rep[1] = pose_from_betas(ut, betas[1], Rs[1], ts[1]);
        {   // approximation 2: [B11 B12 B22]
            double l[18], b3[3];
            for (int i = 0; i < 6; ++i) { l[3 * i] = L[10 * i]; l[3 * i + 1] = L[10 * i + 1]; l[3 * i + 2] = L[10 * i + 2]; }
            Svd s;
            s.compute(l, 6, 3);
            s.back_subst(rho, b3);
            double* be = betas[2];
            if (b3[0] < 0) {
                be[0] = std::sqrt(-b3[0]);
                be[1] = (b3[2] < 0) ? std::sqrt(-b3[2]) : 0.0;
            } else {
                be[0] = std::sqrt(b3[0]);
                be[1] = (b3[2] > 0) ? std::sqrt(b3[2]) : 0.0;
            }
            if (b3[1] < 0) be[0] = -be[0];
            be[2] = 0.0;
            be[3] = 0.0;
        }
        gauss_newton(L, rho, betas[2]);
        rep[2] = pose_from_betas(ut, betas[2], Rs[2], ts[2]);
        {   // approximation 3: [B11 B12 B22 B13 B23]
            double l[30], b5[5];
            for (int i = 0; i < 6; ++i)
                for (int k = 0; k < 5; ++k) l[5 * i + k] = L[10 * i + k];
            Svd s;
            s.compute(l, 6, 5);
            s.back_subst(rho, b5);
            double* be = betas[3];
            if (b5[0] < 0) {
                be[0] = std::sqrt(-b5[0]);
                be[1] = (b5[2] < 0) ? std::sqrt(-b5[2]) : 0.0;
            } else {
                be[0] = std::sqrt(b5[0]);
                be[1] = (b5[2] > 0) ? std::sqrt(b5[2]) : 0.0;
            }
            if (b5[1] < 0) be[0] = -be[0];
            be[2] = b5[3] / be[0];
            be[3] = 0.0;
        }
        gauss_newton(L, rho, betas[3]);
        rep[3] = pose_from_betas(ut, betas[3], Rs[3], ts[3]);
        int N = 1;
        if (rep[2] < rep[1]) N = 2;
        if (rep[3] < rep[N]) N = 3;
        for (int i = 0; i < 3; ++i) {
            for (int j = 0; j < 3; ++j) R_out[3 * i + j] = Rs[N][i][j];
            t_out[i] = ts[N][i];
        }
    }

private:
    int n_;
    const double *pws_, *us_;
    double fu_, fv_, uc_, vc_;
    double alphas_[4 * kEpnpMaxPts], pcs_[3 * kEpnpMaxPts], cws_[4][3], ccs_[4][3];

    static double dot(const double* a, const double* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
    static double dist2(const double* p1, const double* p2) {
        return (p1[0] - p2[0]) * (p1[0] - p2[0]) + (p1[1] - p2[1]) * (p1[1] - p2[1]) + (p1[2] - p2[2]) * (p1[2] - p2[2]);
    }
    void choose_control_points() {
        cws_[0][0] = cws_[0][1] = cws_[0][2] = 0;
        for (int i = 0; i < n_; ++i)
            for (int j = 0; j < 3; ++j) cws_[0][j] += pws_[3 * i + j];
        for (int j = 0; j < 3; ++j) cws_[0][j] /= n_;
        double ptp[9];
        for (int a = 0; a < 3; ++a)
            for (int b = 0; b < 3; ++b) {
                double s = 0;
                for (int i = 0; i < n_; ++i) s += (pws_[3 * i + a] - cws_[0][a]) * (pws_[3 * i + b] - cws_[0][b]);
                ptp[3 * a + b] = s;
            }
        Svd sv;
        sv.compute(ptp, 3, 3);
        for (int i = 1; i < 4; ++i) {
            const double k = std::sqrt(sv.w[i - 1] / n_);
            for (int j = 0; j < 3; ++j) cws_[i][j] = cws_[0][j] + k * sv.u[3 * j + (i - 1)];
        }
    }
    void barycentric() {
        double cc[9], ci[9];
        for (int i = 0; i < 3; ++i)
            for (int j = 1; j < 4; ++j) cc[3 * i + j - 1] = cws_[j][i] - cws_[0][i];
        Svd sv;
        sv.compute(cc, 3, 3);
        sv.inverse(ci);
        for (int i = 0; i < n_; ++i) {
            const double* pi = pws_ + 3 * i;
            double* a = alphas_ + 4 * i;
            for (int j = 0; j < 3; ++j)
                a[1 + j] = ci[3 * j] * (pi[0] - cws_[0][0]) + ci[3 * j + 1] * (pi[1] - cws_[0][1]) + ci[3 * j + 2] * (pi[2] - cws_[0][2]);
            a[0] = 1.0f - a[1] - a[2] - a[3];
        }
    }
    void build_L_rho(const double* ut, double* L, double* rho) const {
        const double* v[4] = {ut + 12 * 11, ut + 12 * 10, ut + 12 * 9, ut + 12 * 8};
        double dv[4][6][3];
        for (int i = 0; i < 4; ++i) {
            int a = 0, b = 1;
            for (int j = 0; j < 6; ++j) {
                for (int k = 0; k < 3; ++k) dv[i][j][k] = v[i][3 * a + k] - v[i][3 * b + k];
                if (++b > 3) {
                    ++a;
                    b = a + 1;
                }
            }
        }
        for (int i = 0; i < 6; ++i) {
            double* row = L + 10 * i;
            row[0] = dot(dv[0][i], dv[0][i]);
            row[1] = 2.0f * dot(dv[0][i], dv[1][i]);
            row[2] = dot(dv[1][i], dv[1][i]);
            row[3] = 2.0f * dot(dv[0][i], dv[2][i]);
            row[4] = 2.0f * dot(dv[1][i], dv[2][i]);
            row[5] = dot(dv[2][i], dv[2][i]);
            row[6] = 2.0f * dot(dv[0][i], dv[3][i]);
            row[7] = 2.0f * dot(dv[1][i], dv[3][i]);
            row[8] = 2.0f * dot(dv[2][i], dv[3][i]);
            row[9] = dot(dv[3][i], dv[3][i]);
        }
        rho[0] = dist2(cws_[0], cws_[1]); rho[1] = dist2(cws_[0], cws_[2]); rho[2] = dist2(cws_[0], cws_[3]);
        rho[3] = dist2(cws_[1], cws_[2]); rho[4] = dist2(cws_[1], cws_[3]); rho[5] = dist2(cws_[2], cws_[3]);
    }
    static void qr_solve(double* pA, int nr, int nc, double* pb, double* pX) {
        double A1[8], A2[8];
        double* ppAkk = pA;
        for (int k = 0; k < nc; ++k) {
            double* ppAik1 = ppAkk;
            double eta = std::fabs(*ppAik1);
            for (int i = k + 1; i < nr; ++i) {       // (the authors' scan: starts on row k again, never reaches the last row)
                const double elt = std::fabs(*ppAik1);
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
            double sigma = std::sqrt(sum2);
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
    static void gauss_newton(const double* L, const double* rho, double* betas) {
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
            qr_solve(A, 6, 4, b, x);
            for (int i = 0; i < 4; ++i) betas[i] += x[i];
        }
    }
    double pose_from_betas(const double* ut, const double* betas, double R[3][3], double t[3]) {
        for (int i = 0; i < 4; ++i) ccs_[i][0] = ccs_[i][1] = ccs_[i][2] = 0.0;
        for (int i = 0; i < 4; ++i) {
            const double* v = ut + 12 * (11 - i);
            for (int j = 0; j < 4; ++j)
                for (int k = 0; k < 3; ++k) ccs_[j][k] += betas[i] * v[3 * j + k];
        }
        for (int i = 0; i < n_; ++i) {
            const double* a = alphas_ + 4 * i;
            double* pc = pcs_ + 3 * i;
            for (int j = 0; j < 3; ++j) pc[j] = a[0] * ccs_[0][j] + a[1] * ccs_[1][j] + a[2] * ccs_[2][j] + a[3] * ccs_[3][j];
        }
        if (pcs_[2] < 0.0) {
            for (int i = 0; i < 4; ++i)
                for (int j = 0; j < 3; ++j) ccs_[i][j] = -ccs_[i][j];
            for (int i = 0; i < 3 * n_; ++i) pcs_[i] = -pcs_[i];
        }
        double pc0[3] = {0, 0, 0}, pw0[3] = {0, 0, 0};
        for (int i = 0; i < n_; ++i)
            for (int j = 0; j < 3; ++j) {
                pc0[j] += pcs_[3 * i + j];
                pw0[j] += pws_[3 * i + j];
            }
        for (int j = 0; j < 3; ++j) {
            pc0[j] /= n_;
            pw0[j] /= n_;
        }
        double abt[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        for (int i = 0; i < n_; ++i) {
            const double* pc = pcs_ + 3 * i;
            const double* pw = pws_ + 3 * i;
            for (int j = 0; j < 3; ++j) {
                abt[3 * j] += (pc[j] - pc0[j]) * (pw[0] - pw0[0]);
                abt[3 * j + 1] += (pc[j] - pc0[j]) * (pw[1] - pw0[1]);
                abt[3 * j + 2] += (pc[j] - pc0[j]) * (pw[2] - pw0[2]);
            }
        }
        Svd sv;
        sv.compute(abt, 3, 3);
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) R[i][j] = sv.u[3 * i] * sv.vt[j] + sv.u[3 * i + 1] * sv.vt[3 + j] + sv.u[3 * i + 2] * sv.vt[6 + j];
        const double det = R[0][0] * R[1][1] * R[2][2] + R[0][1] * R[1][2] * R[2][0] + R[0][2] * R[1][0] * R[2][1] -
                           R[0][2] * R[1][1] * R[2][0] - R[0][1] * R[1][0] * R[2][2] - R[0][0] * R[1][2] * R[2][1];
        if (det < 0) {
            R[2][0] = -R[2][0];
            R[2][1] = -R[2][1];
            R[2][2] = -R[2][2];
        }
        t[0] = pc0[0] - dot(R[0], pw0);
        t[1] = pc0[1] - dot(R[1], pw0);
        t[2] = pc0[2] - dot(R[2], pw0);
        double sum2 = 0.0;
        for (int i = 0; i < n_; ++i) {
            const double* pw = pws_ + 3 * i;
            const double Xc = dot(R[0], pw) + t[0], Yc = dot(R[1], pw) + t[1];
            const double inv_Zc = 1.0 / (dot(R[2], pw) + t[2]);
            const double ue = uc_ + fu_ * Xc * inv_Zc, ve = vc_ + fv_ * Yc * inv_Zc;
            const double u = us_[2 * i], v = us_[2 * i + 1];
            sum2 += std::sqrt((u - ue) * (u - ue) + (v - ve) * (v - ve));
        }
        return sum2 / n_;
    }
};

// ------------------------------------------------------------------------------------------------ P3P
// solvePnP(SOLVEPNP_P3P): what solvePnPRansac runs on exactly four points (model_points = npoints = 4: no RANSAC).  The
// perspective-three-point problem on the first three correspondences — Grunert's quartic in Haralick et al.'s notation
// (a = |P2 P3|, b = |P1 P3|, c = |P1 P2|, s2 = u s1, s3 = v s1), roots by solve_poly, pose from the two orthonormal frames of
// the triangle — and, as OpenCV's p3p class does, the pose that reprojects the fourth point best.  R row-major.
inline void p3p_frame(const double* Q, double* F) {
    double e1[3], w[3], e3[3], e2[3];
    for (int k = 0; k < 3; ++k) {
        e1[k] = Q[3 + k] - Q[k];
        w[k] = Q[6 + k] - Q[k];
    }
    const double n1 = std::sqrt(e1[0] * e1[0] + e1[1] * e1[1] + e1[2] * e1[2]);
    for (int k = 0; k < 3; ++k) e1[k] = e1[k] / n1;
    e3[0] = e1[1] * w[2] - e1[2] * w[1];
    e3[1] = e1[2] * w[0] - e1[0] * w[2];
    e3[2] = e1[0] * w[1] - e1[1] * w[0];
    const double n3 = std::sqrt(e3[0] * e3[0] + e3[1] * e3[1] + e3[2] * e3[2]);
    for (int k = 0; k < 3; ++k) e3[k] = e3[k] / n3;
    e2[0] = e3[1] * e1[2] - e3[2] * e1[1];
    e2[1] = e3[2] * e1[0] - e3[0] * e1[2];
    e2[2] = e3[0] * e1[1] - e3[1] * e1[0];
    for (int k = 0; k < 3; ++k) {
        F[3 * k] = e1[k];
        F[3 * k + 1] = e2[k];
        F[3 * k + 2] = e3[k];
    }
}

inline bool p3p(const double* K, const double* Xw /*4 x 3*/, const double* uv /*4 x 2 pixels*/, double* Rout, double* tout) {
    const double fx = K[0], fy = K[4], cx = K[2], cy = K[5];
    double f[3][3];
    for (int i = 0; i < 3; ++i) {
        const double x = (uv[2 * i] - cx) / fx, y = (uv[2 * i + 1] - cy) / fy;
        const double nrm = std::sqrt(x * x + y * y + 1.0);
        f[i][0] = x / nrm;
        f[i][1] = y / nrm;
        f[i][2] = 1.0 / nrm;
    }
    double d12[3], d13[3], d23[3];
    for (int k = 0; k < 3; ++k) {
        d23[k] = Xw[3 + k] - Xw[6 + k];
        d13[k] = Xw[k] - Xw[6 + k];
        d12[k] = Xw[k] - Xw[3 + k];
    }
    const double a2 = d23[0] * d23[0] + d23[1] * d23[1] + d23[2] * d23[2];
    const double b2 = d13[0] * d13[0] + d13[1] * d13[1] + d13[2] * d13[2];
    const double c2 = d12[0] * d12[0] + d12[1] * d12[1] + d12[2] * d12[2];
    if (!(a2 > 0) || !(b2 > 0) || !(c2 > 0)) return false;
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
    Cx roots[4];
    const int nr = solve_poly(A, 4, roots);
    bool found = false;
    double best = 0;
    for (int r = 0; r < nr; ++r) {
        const double v = roots[r].re;
        if (!(std::fabs(roots[r].im) <= 1e-9 * (1.0 + std::fabs(v))) || !(v > 0)) continue;
        const double den = 2 * (cg - v * ca);
        if (den == 0) continue;
        const double u = ((k1 - 1) * v * v - 2 * k1 * cb * v + 1 + k1) / den;
        if (!(u > 0)) continue;
        const double q = 1 + v * v - 2 * v * cb;
        if (!(q > 0)) continue;
        const double s1 = std::sqrt(b2 / q), s2 = u * s1, s3 = v * s1;
        double Y[9], FY[9], FP[9], R[9], t[3];
        for (int k = 0; k < 3; ++k) {
            Y[k] = s1 * f[0][k];
            Y[3 + k] = s2 * f[1][k];
            Y[6 + k] = s3 * f[2][k];
        }
        p3p_frame(Y, FY);
        p3p_frame(Xw, FP);
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) R[3 * i + j] = FY[3 * i] * FP[3 * j] + FY[3 * i + 1] * FP[3 * j + 1] + FY[3 * i + 2] * FP[3 * j + 2];
        for (int i = 0; i < 3; ++i) t[i] = Y[i] - (R[3 * i] * Xw[0] + R[3 * i + 1] * Xw[1] + R[3 * i + 2] * Xw[2]);
        bool finite = true;
        for (int k = 0; k < 9; ++k) finite = finite && std::isfinite(R[k]);
        for (int k = 0; k < 3; ++k) finite = finite && std::isfinite(t[k]);
        if (!finite) continue;
        const double* P4 = Xw + 9;
        const double x4 = R[0] * P4[0] + R[1] * P4[1] + R[2] * P4[2] + t[0];
        const double y4 = R[3] * P4[0] + R[4] * P4[1] + R[5] * P4[2] + t[1];
        const double z4 = R[6] * P4[0] + R[7] * P4[1] + R[8] * P4[2] + t[2];
        const double eu = fx * (x4 / z4) + cx - uv[6], ev = fy * (y4 / z4) + cy - uv[7];
        const double err = eu * eu + ev * ev;
        if (!found || err < best) {
            best = err;
            found = true;
            std::memcpy(Rout, R, sizeof(R));
            std::memcpy(tout, t, sizeof(t));
        }
    }
    return found;
}

// ------------------------------------------------------------------------------------------------ ITERATIVE init
// cvFindExtrinsicCameraParams2's non-planar initialisation: 2n x 12 DLT on normalised image points, SVD of L^T L, the
// 3x3 part re-orthonormalised, translation rescaled.  Returns 0, 1 (planar object: not handled here) or 2 (< 6 points).
template <typename Real>
inline int pnp_dlt_init(const Real* X, const Real* uv, const int32_t* sel, int64_t n, const double* K, double* rvec, double* tvec) {
    const double ifx = 1. / K[0], ify = 1. / K[4];
    auto P = [&](int64_t i, int j) { return (double)X[3 * (sel ? sel[i] : i) + j]; };
    auto Q = [&](int64_t i, int j) { return (double)uv[2 * (sel ? sel[i] : i) + j]; };
    double Mc[3] = {0, 0, 0}, MM[9];
    for (int64_t i = 0; i < n; ++i)
        for (int j = 0; j < 3; ++j) Mc[j] += P(i, j);
    for (int j = 0; j < 3; ++j) Mc[j] /= (double)n;
    for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 3; ++b) {
            double s = 0;
            for (int64_t i = 0; i < n; ++i) s += (P(i, a) - Mc[a]) * (P(i, b) - Mc[b]);
            MM[3 * a + b] = s;
        }
    Svd sv;
    sv.compute(MM, 3, 3);
    if (sv.w[2] / sv.w[1] < 1e-3) return 1;
    if (n < 6) return 2;
    double LL[144];
    std::memset(LL, 0, sizeof(LL));
    // L^T L (cvMulTransposed): every entry is the sum over the 2n rows IN ORDER of the products of two columns.  Row 2i =
    // [X Y Z 1 | 0 0 0 0 | xX xY xZ x], row 2i+1 = [0 0 0 0 | X Y Z 1 | yX yY yZ y]: the structurally zero products add
    // exactly nothing, so one pass over the points with the upper triangle's accumulators gives the same sums bit for
    // bit without a 2n x 12 buffer.
    for (int64_t i = 0; i < n; ++i) {
        const double Xh[4] = {P(i, 0), P(i, 1), P(i, 2), 1.};
        const double x = -((Q(i, 0) - K[2]) * ifx), y = -((Q(i, 1) - K[5]) * ify);
        double r0[12], r1[12];
        for (int c = 0; c < 4; ++c) {
            r0[c] = Xh[c]; r0[4 + c] = 0.; r0[8 + c] = c == 3 ? x : x * Xh[c];
            r1[c] = 0.; r1[4 + c] = Xh[c]; r1[8 + c] = c == 3 ? y : y * Xh[c];
        }
        static const int nz0[8] = {0, 1, 2, 3, 8, 9, 10, 11}, nz1[8] = {4, 5, 6, 7, 8, 9, 10, 11};
        for (int p = 0; p < 8; ++p)
            for (int q = p; q < 8; ++q) LL[12 * nz0[p] + nz0[q]] += r0[nz0[p]] * r0[nz0[q]];
        for (int p = 0; p < 8; ++p)
            for (int q = p; q < 8; ++q) LL[12 * nz1[p] + nz1[q]] += r1[nz1[p]] * r1[nz1[q]];
    }
    for (int a = 0; a < 12; ++a)
        for (int b = 0; b < a; ++b) LL[12 * a + b] = LL[12 * b + a];
    sv.compute(LL, 12, 12);
    double RR[9], tt[3];
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) RR[3 * i + j] = sv.vt[11 * 12 + 4 * i + j];
        tt[i] = sv.vt[11 * 12 + 4 * i + 3];
    }
    if (det3(RR) < 0) {
        for (double& v : RR) v *= -1;
        for (double& v : tt) v *= -1;
    }
    double sc = 0;
    for (double v : RR) sc += v * v;
    sc = std::sqrt(sc);
    if (!(std::fabs(sc) > DBL_EPSILON)) return 2;
    Svd sr;
    sr.compute(RR, 3, 3);
    double R[9], rn = 0;
    mul3(sr.u, sr.vt, R);
    for (double v : R) rn += v * v;
    const double f = std::sqrt(rn) / sc;
    for (int k = 0; k < 3; ++k) tvec[k] = tt[k] * f;
    rodrigues_mat2vec(R, rvec);
    return 0;
}

}  // namespace host
}  // namespace sfm
