// Host-side minimal solver of solvePnPRansac (sfm.py:67): EPnP (Lepetit, Moreno-Noguer, Fua 2009) on the 5-point
// samples — the hypothesis generator between two device scoring launches.  The incremental driver calls it ~4 times
// per registered camera; in NumPy it was 60 % of a 57-camera run (1.2 ms per call), here it is a few microseconds.
// Pure C++ (no device code); exported through the same C ABI as the kernels.  Restates sfm_mvs_amd/hostgeom.py::epnp
// step for step (control points from the PCA of the sample, barycentric coordinates, the 4 smallest eigenvectors of
// M^T M, the three beta approximations + 5 Gauss-Newton steps each, Procrustes pose, smallest reprojection error wins).
#include "common.h"
#include <cmath>
#include <cstring>
#include <limits>

namespace {

// Cyclic Jacobi eigen-decomposition of a symmetric n x n matrix (n <= 12): A = V diag(w) V^T, eigenvalues DESCENDING,
// eigenvectors in the COLUMNS of V.
template <int N>
void sym_eig(const double (&Ain)[N][N], double (&w)[N], double (&V)[N][N]) {
    double A[N][N];
    std::memcpy(A, Ain, sizeof(A));
    for (int i = 0; i < N; ++i)
        for (int j = 0; j < N; ++j) V[i][j] = i == j ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 60; ++sweep) {
        double off = 0, diag = 0;
        for (int i = 0; i < N; ++i) {
            diag += A[i][i] * A[i][i];
            for (int j = i + 1; j < N; ++j) off += A[i][j] * A[i][j];
        }
        if (off <= 1e-30 * (diag + off) || off == 0) break;
        for (int p = 0; p < N; ++p)
            for (int q = p + 1; q < N; ++q) {
                if (A[p][q] == 0) continue;
                const double theta = (A[q][q] - A[p][p]) / (2 * A[p][q]);
                const double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1));
                const double c = 1 / std::sqrt(t * t + 1), s = t * c;
                for (int k = 0; k < N; ++k) {
                    const double akp = A[k][p], akq = A[k][q];
                    A[k][p] = c * akp - s * akq;
                    A[k][q] = s * akp + c * akq;
                }
                for (int k = 0; k < N; ++k) {
                    const double apk = A[p][k], aqk = A[q][k];
                    A[p][k] = c * apk - s * aqk;
                    A[q][k] = s * apk + c * aqk;
                }
                for (int k = 0; k < N; ++k) {
                    const double vkp = V[k][p], vkq = V[k][q];
                    V[k][p] = c * vkp - s * vkq;
                    V[k][q] = s * vkp + c * vkq;
                }
            }
    }
    for (int i = 0; i < N; ++i) w[i] = A[i][i];
    for (int i = 0; i < N - 1; ++i) {          // selection sort, descending
        int m = i;
        for (int j = i + 1; j < N; ++j)
            if (w[j] > w[m]) m = j;
        if (m != i) {
            std::swap(w[i], w[m]);
            for (int k = 0; k < N; ++k) std::swap(V[k][i], V[k][m]);
        }
    }
}

// Minimum-norm least squares x = argmin |A x - b| for a 6 x NC system (NC <= 5) through the eigen-decomposition of
// A^T A (singular values below rcond * sigma_max are dropped, like numpy.linalg.lstsq's default).
template <int NC>
void lstsq6(const double (&A)[6][NC], const double (&b)[6], double (&x)[NC]) {
    double AtA[NC][NC], Atb[NC], w[NC], V[NC][NC];
    for (int i = 0; i < NC; ++i) {
        Atb[i] = 0;
        for (int r = 0; r < 6; ++r) Atb[i] += A[r][i] * b[r];
        for (int j = 0; j < NC; ++j) {
            double s = 0;
            for (int r = 0; r < 6; ++r) s += A[r][i] * A[r][j];
            AtA[i][j] = s;
        }
    }
    sym_eig<NC>(AtA, w, V);
    const double cut = w[0] * (6 * std::numeric_limits<double>::epsilon()) * (6 * std::numeric_limits<double>::epsilon());
    for (int i = 0; i < NC; ++i) x[i] = 0;
    for (int k = 0; k < NC; ++k) {
        if (!(w[k] > cut) || !(w[k] > 0)) continue;
        double proj = 0;
        for (int i = 0; i < NC; ++i) proj += V[i][k] * Atb[i];
        proj /= w[k];
        for (int i = 0; i < NC; ++i) x[i] += V[i][k] * proj;
    }
}

// R = U V^T of the SVD of a 3x3 matrix B (det fixed like hostgeom: negative → last ROW of R negated)
void procrustes_rotation(const double (&B)[3][3], double (&R)[3][3]) {
    // B^T B = V S^2 V^T;  U = B V S^-1 (columns with tiny singular value completed by cross product)
    double BtB[3][3], w[3], V[3][3];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            double s = 0;
            for (int k = 0; k < 3; ++k) s += B[k][i] * B[k][j];
            BtB[i][j] = s;
        }
    sym_eig<3>(BtB, w, V);
    double U[3][3];
    for (int c = 0; c < 3; ++c) {
        double col[3] = {0, 0, 0}, nrm = 0;
        for (int r = 0; r < 3; ++r) {
            for (int k = 0; k < 3; ++k) col[r] += B[r][k] * V[k][c];
            nrm += col[r] * col[r];
        }
        nrm = std::sqrt(nrm);
        for (int r = 0; r < 3; ++r) U[r][c] = nrm > 0 ? col[r] / nrm : 0;
    }
    if (!(w[2] > 1e-24 * w[0])) {     // rank-deficient: third left vector = u0 x u1 (any sign: fixed by the det test below)
        U[0][2] = U[1][0] * U[2][1] - U[2][0] * U[1][1];
        U[1][2] = U[2][0] * U[0][1] - U[0][0] * U[2][1];
        U[2][2] = U[0][0] * U[1][1] - U[1][0] * U[0][1];
    }
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            double s = 0;
            for (int k = 0; k < 3; ++k) s += U[i][k] * V[j][k];
            R[i][j] = s;
        }
    const double det = R[0][0] * (R[1][1] * R[2][2] - R[1][2] * R[2][1]) - R[0][1] * (R[1][0] * R[2][2] - R[1][2] * R[2][0]) +
                       R[0][2] * (R[1][0] * R[2][1] - R[1][1] * R[2][0]);
    if (det < 0)
        for (int j = 0; j < 3; ++j) R[2][j] = -R[2][j];
}

constexpr int kMaxPts = 64;

}  // namespace

// K_host 9 doubles (row-major), Xw_host [n x 3] doubles, uv_host [n x 2] doubles (pixels), 4 <= n <= 64.
// Outputs R_host (9 doubles, row-major) and t_host (3 doubles).
extern "C" int sfm_host_epnp(const double* K, const double* Xw, const double* uv, int n, double* R_out, double* t_out) {
    SFM_CHECK_ARG(K && Xw && uv && R_out && t_out, "sfm_host_epnp: null pointer");
    SFM_CHECK_ARG(n >= 4 && n <= kMaxPts, "sfm_host_epnp: n must be in [4, %d] (got %d)", kMaxPts, n);
    const double fu = K[0], fv = K[4], uc = K[2], vc = K[5];
    // control points: centroid + principal directions scaled by sqrt(lambda / n)
    double cws[4][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
    for (int i = 0; i < n; ++i)
        for (int k = 0; k < 3; ++k) cws[0][k] += Xw[3 * i + k];
    for (int k = 0; k < 3; ++k) cws[0][k] /= n;
    double PtP[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}}, dc[3], Uc[3][3];
    for (int i = 0; i < n; ++i)
        for (int a = 0; a < 3; ++a)
            for (int b = 0; b < 3; ++b) PtP[a][b] += (Xw[3 * i + a] - cws[0][a]) * (Xw[3 * i + b] - cws[0][b]);
    sym_eig<3>(PtP, dc, Uc);
    for (int i = 1; i < 4; ++i) {
        const double k = std::sqrt(std::fmax(dc[i - 1], 0.0) / n);
        for (int a = 0; a < 3; ++a) cws[i][a] = cws[0][a] + k * Uc[a][i - 1];
    }
    // barycentric coordinates: solve CC a = (X - c0), CC columns = c_i - c_0 (pseudo-inverse if singular)
    double CC[3][3], CCi[3][3];
    for (int a = 0; a < 3; ++a)
        for (int i = 0; i < 3; ++i) CC[a][i] = cws[i + 1][a] - cws[0][a];
    {
        const double det = CC[0][0] * (CC[1][1] * CC[2][2] - CC[1][2] * CC[2][1]) - CC[0][1] * (CC[1][0] * CC[2][2] - CC[1][2] * CC[2][0]) +
                           CC[0][2] * (CC[1][0] * CC[2][1] - CC[1][1] * CC[2][0]);
        double scale = 0;
        for (int a = 0; a < 3; ++a)
            for (int b = 0; b < 3; ++b) scale = std::fmax(scale, std::fabs(CC[a][b]));
        if (std::fabs(det) > 1e-14 * scale * scale * scale && det != 0) {
            CCi[0][0] = (CC[1][1] * CC[2][2] - CC[1][2] * CC[2][1]) / det;
            CCi[0][1] = (CC[0][2] * CC[2][1] - CC[0][1] * CC[2][2]) / det;
            CCi[0][2] = (CC[0][1] * CC[1][2] - CC[0][2] * CC[1][1]) / det;
            CCi[1][0] = (CC[1][2] * CC[2][0] - CC[1][0] * CC[2][2]) / det;
            CCi[1][1] = (CC[0][0] * CC[2][2] - CC[0][2] * CC[2][0]) / det;
            CCi[1][2] = (CC[0][2] * CC[1][0] - CC[0][0] * CC[1][2]) / det;
            CCi[2][0] = (CC[1][0] * CC[2][1] - CC[1][1] * CC[2][0]) / det;
            CCi[2][1] = (CC[0][1] * CC[2][0] - CC[0][0] * CC[2][1]) / det;
            CCi[2][2] = (CC[0][0] * CC[1][1] - CC[0][1] * CC[1][0]) / det;
        } else {   // pinv(CC) = V S^+ U^T through the eigen-decomposition of CC^T CC
            double CtC[3][3], w[3], V[3][3];
            for (int i = 0; i < 3; ++i)
                for (int j = 0; j < 3; ++j) {
                    double s = 0;
                    for (int k = 0; k < 3; ++k) s += CC[k][i] * CC[k][j];
                    CtC[i][j] = s;
                }
            sym_eig<3>(CtC, w, V);
            for (int i = 0; i < 3; ++i)
                for (int j = 0; j < 3; ++j) {
                    double s = 0;
                    for (int k = 0; k < 3; ++k)
                        if (w[k] > 1e-15 * 1e-15 * 9 * w[0] && w[k] > 0) {
                            double vtct = 0;                    // (V^T CC^T)[k][j]
                            for (int m = 0; m < 3; ++m) vtct += V[m][k] * CC[j][m];
                            s += V[i][k] * vtct / w[k];
                        }
                    CCi[i][j] = s;
                }
        }
    }
    double alphas[kMaxPts][4];
    for (int i = 0; i < n; ++i) {
        double s = 0;
        for (int a = 0; a < 3; ++a) {
            double v = 0;
            for (int b = 0; b < 3; ++b) v += CCi[a][b] * (Xw[3 * i + b] - cws[0][b]);
            alphas[i][a + 1] = v;
            s += v;
        }
        alphas[i][0] = 1 - s;
    }
    // M^T M (12 x 12) from the 2n x 12 rows
    double MtM[12][12];
    std::memset(MtM, 0, sizeof(MtM));
    for (int i = 0; i < n; ++i) {
        double r0[12], r1[12];
        for (int j = 0; j < 4; ++j) {
            r0[3 * j] = alphas[i][j] * fu; r0[3 * j + 1] = 0;                  r0[3 * j + 2] = alphas[i][j] * (uc - uv[2 * i]);
            r1[3 * j] = 0;                 r1[3 * j + 1] = alphas[i][j] * fv;  r1[3 * j + 2] = alphas[i][j] * (vc - uv[2 * i + 1]);
        }
        for (int a = 0; a < 12; ++a)
            for (int b = 0; b < 12; ++b) MtM[a][b] += r0[a] * r0[b] + r1[a] * r1[b];
    }
    double ew[12], EV[12][12];
    sym_eig<12>(MtM, ew, EV);
    double v[4][12];                                               // v[0] = eigenvector of the SMALLEST eigenvalue
    for (int i = 0; i < 4; ++i)
        for (int k = 0; k < 12; ++k) v[i][k] = EV[k][11 - i];
    static const int pa[6] = {0, 0, 0, 1, 1, 2}, pb[6] = {1, 2, 3, 2, 3, 3};
    double dv[4][6][3];
    for (int i = 0; i < 4; ++i)
        for (int p = 0; p < 6; ++p)
            for (int k = 0; k < 3; ++k) dv[i][p][k] = v[i][3 * pa[p] + k] - v[i][3 * pb[p] + k];
    auto dot3 = [](const double* a, const double* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; };
    double L[6][10], rho[6];
    for (int p = 0; p < 6; ++p) {
        const double *d0 = dv[0][p], *d1 = dv[1][p], *d2 = dv[2][p], *d3 = dv[3][p];
        L[p][0] = dot3(d0, d0); L[p][1] = 2 * dot3(d0, d1); L[p][2] = dot3(d1, d1); L[p][3] = 2 * dot3(d0, d2);
        L[p][4] = 2 * dot3(d1, d2); L[p][5] = dot3(d2, d2); L[p][6] = 2 * dot3(d0, d3); L[p][7] = 2 * dot3(d1, d3);
        L[p][8] = 2 * dot3(d2, d3); L[p][9] = dot3(d3, d3);
        rho[p] = 0;
        for (int k = 0; k < 3; ++k) rho[p] += (cws[pa[p]][k] - cws[pb[p]][k]) * (cws[pa[p]][k] - cws[pb[p]][k]);
    }
    double betas[3][4];
    {   // approximation 1: betas from columns {0, 1, 3, 6}
        double A[6][4], b4[4];
        static const int cols[4] = {0, 1, 3, 6};
        for (int p = 0; p < 6; ++p)
            for (int c = 0; c < 4; ++c) A[p][c] = L[p][cols[c]];
        lstsq6<4>(A, rho, b4);
        if (b4[0] < 0) {
            const double b0 = std::sqrt(-b4[0]);
            betas[0][0] = b0; betas[0][1] = -b4[1] / b0; betas[0][2] = -b4[2] / b0; betas[0][3] = -b4[3] / b0;
        } else {
            const double b0 = std::sqrt(b4[0]);
            betas[0][0] = b0; betas[0][1] = b4[1] / b0; betas[0][2] = b4[2] / b0; betas[0][3] = b4[3] / b0;
        }
    }
    {   // approximation 2: columns {0, 1, 2}
        double A[6][3], b3[3];
        for (int p = 0; p < 6; ++p)
            for (int c = 0; c < 3; ++c) A[p][c] = L[p][c];
        lstsq6<3>(A, rho, b3);
        double b0, b1;
        if (b3[0] < 0) { b0 = std::sqrt(-b3[0]); b1 = b3[2] < 0 ? std::sqrt(-b3[2]) : 0.0; }
        else { b0 = std::sqrt(b3[0]); b1 = b3[2] > 0 ? std::sqrt(b3[2]) : 0.0; }
        if (b3[1] < 0) b0 = -b0;
        betas[1][0] = b0; betas[1][1] = b1; betas[1][2] = 0; betas[1][3] = 0;
    }
    {   // approximation 3: columns {0, 1, 2, 3, 4}
        double A[6][5], b5[5];
        for (int p = 0; p < 6; ++p)
            for (int c = 0; c < 5; ++c) A[p][c] = L[p][c];
        lstsq6<5>(A, rho, b5);
        double b0, b1;
        if (b5[0] < 0) { b0 = std::sqrt(-b5[0]); b1 = b5[2] < 0 ? std::sqrt(-b5[2]) : 0.0; }
        else { b0 = std::sqrt(b5[0]); b1 = b5[2] > 0 ? std::sqrt(b5[2]) : 0.0; }
        if (b5[1] < 0) b0 = -b0;
        betas[2][0] = b0; betas[2][1] = b1; betas[2][2] = b0 != 0 ? b5[3] / b0 : 0.0; betas[2][3] = 0;
    }
    double pw0[3] = {cws[0][0], cws[0][1], cws[0][2]};            // = mean of Xw
    double best_err = std::numeric_limits<double>::infinity();
    bool have = false;
    for (int c = 0; c < 3; ++c) {
        double b[4] = {betas[c][0], betas[c][1], betas[c][2], betas[c][3]};
        for (int it = 0; it < 5; ++it) {                           // Gauss-Newton on the 6 distance constraints
            double A[6][4], r[6], dx[4];
            for (int p = 0; p < 6; ++p) {
                const double* l = L[p];
                A[p][0] = 2 * l[0] * b[0] + l[1] * b[1] + l[3] * b[2] + l[6] * b[3];
                A[p][1] = l[1] * b[0] + 2 * l[2] * b[1] + l[4] * b[2] + l[7] * b[3];
                A[p][2] = l[3] * b[0] + l[4] * b[1] + 2 * l[5] * b[2] + l[8] * b[3];
                A[p][3] = l[6] * b[0] + l[7] * b[1] + l[8] * b[2] + 2 * l[9] * b[3];
                r[p] = rho[p] - (l[0] * b[0] * b[0] + l[1] * b[0] * b[1] + l[2] * b[1] * b[1] + l[3] * b[0] * b[2] + l[4] * b[1] * b[2] +
                                 l[5] * b[2] * b[2] + l[6] * b[0] * b[3] + l[7] * b[1] * b[3] + l[8] * b[2] * b[3] + l[9] * b[3] * b[3]);
            }
            lstsq6<4>(A, r, dx);
            for (int k = 0; k < 4; ++k) b[k] += dx[k];
        }
        // pose from the betas
        double ccs[4][3];
        for (int j = 0; j < 4; ++j)
            for (int k = 0; k < 3; ++k) ccs[j][k] = b[0] * v[0][3 * j + k] + b[1] * v[1][3 * j + k] + b[2] * v[2][3 * j + k] + b[3] * v[3][3 * j + k];
        double pcs[kMaxPts][3], pc0[3] = {0, 0, 0};
        for (int i = 0; i < n; ++i)
            for (int k = 0; k < 3; ++k) {
                pcs[i][k] = alphas[i][0] * ccs[0][k] + alphas[i][1] * ccs[1][k] + alphas[i][2] * ccs[2][k] + alphas[i][3] * ccs[3][k];
            }
        if (pcs[0][2] < 0)
            for (int i = 0; i < n; ++i)
                for (int k = 0; k < 3; ++k) pcs[i][k] = -pcs[i][k];
        for (int i = 0; i < n; ++i)
            for (int k = 0; k < 3; ++k) pc0[k] += pcs[i][k];
        for (int k = 0; k < 3; ++k) pc0[k] /= n;
        double ABt[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}}, R[3][3], t[3];
        for (int i = 0; i < n; ++i)
            for (int a = 0; a < 3; ++a)
                for (int bb = 0; bb < 3; ++bb) ABt[a][bb] += (pcs[i][a] - pc0[a]) * (Xw[3 * i + bb] - pw0[bb]);
        procrustes_rotation(ABt, R);
        for (int a = 0; a < 3; ++a) t[a] = pc0[a] - (R[a][0] * pw0[0] + R[a][1] * pw0[1] + R[a][2] * pw0[2]);
        double err = 0;
        for (int i = 0; i < n; ++i) {
            const double X = R[0][0] * Xw[3 * i] + R[0][1] * Xw[3 * i + 1] + R[0][2] * Xw[3 * i + 2] + t[0];
            const double Y = R[1][0] * Xw[3 * i] + R[1][1] * Xw[3 * i + 1] + R[1][2] * Xw[3 * i + 2] + t[1];
            const double Z = R[2][0] * Xw[3 * i] + R[2][1] * Xw[3 * i + 1] + R[2][2] * Xw[3 * i + 2] + t[2];
            const double ue = uc + fu * X / Z, ve = vc + fv * Y / Z;
            err += std::sqrt((uv[2 * i] - ue) * (uv[2 * i] - ue) + (uv[2 * i + 1] - ve) * (uv[2 * i + 1] - ve));
        }
        err /= n;
        if (!std::isfinite(err)) err = std::numeric_limits<double>::infinity();
        if (!have || err < best_err) {          // first candidate, then strictly better ones (hostgeom's selection)
            have = true;
            best_err = err;
            for (int a = 0; a < 3; ++a) {
                for (int bb = 0; bb < 3; ++bb) R_out[3 * a + bb] = R[a][bb];
                t_out[a] = t[a];
            }
        }
    }
    return SFM_OK;
}
