// libpvnet_pnp.so -- host-side pose refinement for the voted key-points (include/pvnet_pnp.h).
//
// What it replaces: the Ceres problem of lib/utils/extend_utils/src/uncertainty_pnp.cpp:61-92 (one 2-residual block per
// key-point, residual = W * (project(R(aa) X + t) - x), uncertainty_pnp.cpp:18-35) and, with identity weights, the
// Levenberg-Marquardt stage of cv2.solvePnP(SOLVEPNP_ITERATIVE) behind lib/utils/evaluation_utils.py:19-52.
// Dense LM on the 6 pose parameters (angle-axis, translation) with analytic Jacobians; 2*pn x 6 with pn = 9.
// Plain C++17, no dependencies; built by pvnet_amd/build.py with g++.
#include "pvnet_pnp.h"

#include <cfloat>
#include <cmath>
#include <cstdlib>
#include <ctime>
#include <cstring>
#include <thread>
#include <vector>

namespace {

void cross_matrix(const double* v, double* M) {  // [v]x, row-major
    M[0] = 0; M[1] = -v[2]; M[2] = v[1];
    M[3] = v[2]; M[4] = 0; M[5] = -v[0];
    M[6] = -v[1]; M[7] = v[0]; M[8] = 0;
}
void matmul3(const double* A, const double* B, double* C) {
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) C[i * 3 + j] = A[i * 3] * B[j] + A[i * 3 + 1] * B[3 + j] + A[i * 3 + 2] * B[6 + j];
}

// R = exp([w]x) and the right Jacobian Jr(w) of SO(3): R(w + d) = R(w) exp([Jr d]x) + O(d^2)
void rotation_and_right_jacobian(const double* w, double* R, double* Jr) {
    const double th2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2], th = std::sqrt(th2);
    double a, b, c;  // sin(th)/th, (1 - cos th)/th^2, (th - sin th)/th^3
    if (th < 1e-5) {  // series: accurate to O(th^4)
        a = 1.0 - th2 / 6.0;
        b = 0.5 - th2 / 24.0;
        c = 1.0 / 6.0 - th2 / 120.0;
    } else {
        a = std::sin(th) / th;
        b = (1.0 - std::cos(th)) / th2;
        c = (th - std::sin(th)) / (th2 * th);
    }
    double Wx[9], W2[9];
    cross_matrix(w, Wx);
    matmul3(Wx, Wx, W2);
    for (int i = 0; i < 9; ++i) {
        const double I = (i % 4 == 0) ? 1.0 : 0.0;
        R[i] = I + a * Wx[i] + b * W2[i];
        if (Jr) Jr[i] = I - b * Wx[i] + c * W2[i];
    }
}

struct Problem {
    const double *x2, *x3, *wgt, *K;
    int pn;
};

// residuals r[2 pn] and (optionally) Jacobian J[2 pn][6] at pose p; false if a point falls on / behind the camera plane
bool evaluate(const Problem& P, const double* p, double* r, double* J) {
    double R[9], Jr[9];
    rotation_and_right_jacobian(p, R, J ? Jr : nullptr);
    const double fx = P.K[0], fy = P.K[4], px = P.K[2], py = P.K[5];
    for (int i = 0; i < P.pn; ++i) {
        const double* X = P.x3 + 3 * i;
        const double RX[3] = {R[0] * X[0] + R[1] * X[1] + R[2] * X[2], R[3] * X[0] + R[4] * X[1] + R[5] * X[2],
                              R[6] * X[0] + R[7] * X[1] + R[8] * X[2]};
        const double Y[3] = {RX[0] + p[3], RX[1] + p[4], RX[2] + p[5]};
        if (!(std::fabs(Y[2]) > 1e-12)) return false;
        const double iz = 1.0 / Y[2];
        const double dx = fx * Y[0] * iz + px - P.x2[2 * i], dy = fy * Y[1] * iz + py - P.x2[2 * i + 1];
        double wxx = 1.0, wxy = 0.0, wyy = 1.0;
        if (P.wgt) { wxx = P.wgt[3 * i]; wxy = P.wgt[3 * i + 1]; wyy = P.wgt[3 * i + 2]; }
        r[2 * i] = wxx * dx + wxy * dy;          // uncertainty_pnp.cpp:29-30
        r[2 * i + 1] = wxy * dx + wyy * dy;
        if (!J) continue;
        // d(proj)/dY (2x3), dY/dt = I, dY/dw = -R [X]x Jr
        const double A[6] = {fx * iz, 0.0, -fx * Y[0] * iz * iz, 0.0, fy * iz, -fy * Y[1] * iz * iz};
        double Xx[9], RXx[9], D[9];
        cross_matrix(X, Xx);
        matmul3(R, Xx, RXx);
        matmul3(RXx, Jr, D);
        double Jp[12];  // unweighted 2x6
        for (int c = 0; c < 3; ++c) {
            Jp[c] = -(A[0] * D[c] + A[1] * D[3 + c] + A[2] * D[6 + c]);
            Jp[6 + c] = -(A[3] * D[c] + A[4] * D[3 + c] + A[5] * D[6 + c]);
            Jp[3 + c] = A[c];
            Jp[9 + c] = A[3 + c];
        }
        for (int c = 0; c < 6; ++c) {
            J[(2 * i) * 6 + c] = wxx * Jp[c] + wxy * Jp[6 + c];
            J[(2 * i + 1) * 6 + c] = wxy * Jp[c] + wyy * Jp[6 + c];
        }
    }
    return true;
}

// solve the symmetric positive definite 6x6 system A x = b by Cholesky; false if A is not positive definite
bool solve6(const double* A, const double* b, double* x) {
    double L[36];
    std::memset(L, 0, sizeof(L));
    for (int i = 0; i < 6; ++i)
        for (int j = 0; j <= i; ++j) {
            double s = A[i * 6 + j];
            for (int k = 0; k < j; ++k) s -= L[i * 6 + k] * L[j * 6 + k];
            if (i == j) {
                if (!(s > 0.0) || !std::isfinite(s)) return false;
                L[i * 6 + i] = std::sqrt(s);
            } else {
                L[i * 6 + j] = s / L[j * 6 + j];
            }
        }
    double y[6];
    for (int i = 0; i < 6; ++i) {
        double s = b[i];
        for (int k = 0; k < i; ++k) s -= L[i * 6 + k] * y[k];
        y[i] = s / L[i * 6 + i];
    }
    for (int i = 5; i >= 0; --i) {
        double s = y[i];
        for (int k = i + 1; k < 6; ++k) s -= L[k * 6 + i] * x[k];
        x[i] = s / L[i * 6 + i];
    }
    return true;
}

constexpr int MAX_PN = 4096;

// eigen-decomposition of a symmetric n x n matrix (row-major, n <= 12) by cyclic Jacobi rotations:
// A -> diagonal (eigenvalues in w), V columns = eigenvectors
void jacobi_eigen(double* A, int n, double* V, double* w) {
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) V[i * n + j] = i == j ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 60; ++sweep) {
        double off = 0, diag = 0;
        for (int i = 0; i < n; ++i)
            for (int j = 0; j < n; ++j) (i == j ? diag : off) += A[i * n + j] * A[i * n + j];
        if (off <= 1e-30 * diag || off == 0.0) break;
        for (int p = 0; p < n - 1; ++p)
            for (int q = p + 1; q < n; ++q) {
                const double apq = A[p * n + q];
                if (apq == 0.0) continue;
                const double theta = (A[q * n + q] - A[p * n + p]) / (2.0 * apq);
                const double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
                const double c = 1.0 / std::sqrt(t * t + 1.0), sn = t * c;
                for (int k = 0; k < n; ++k) {  // A <- J^T A J, applied as column then row rotations
                    const double akp = A[k * n + p], akq = A[k * n + q];
                    A[k * n + p] = c * akp - sn * akq;
                    A[k * n + q] = sn * akp + c * akq;
                }
                for (int k = 0; k < n; ++k) {
                    const double apk = A[p * n + k], aqk = A[q * n + k];
                    A[p * n + k] = c * apk - sn * aqk;
                    A[q * n + k] = sn * apk + c * aqk;
                }
                for (int k = 0; k < n; ++k) {
                    const double vkp = V[k * n + p], vkq = V[k * n + q];
                    V[k * n + p] = c * vkp - sn * vkq;
                    V[k * n + q] = sn * vkp + c * vkq;
                }
            }
    }
    for (int i = 0; i < n; ++i) w[i] = A[i * n + i];
}

double det3(const double* M) {
    return M[0] * (M[4] * M[8] - M[5] * M[7]) - M[1] * (M[3] * M[8] - M[5] * M[6]) + M[2] * (M[3] * M[7] - M[4] * M[6]);
}

bool inverse3(const double* M, double* I) {
    const double d = det3(M);
    if (!(std::fabs(d) > 1e-300)) return false;
    I[0] = (M[4] * M[8] - M[5] * M[7]) / d; I[1] = (M[2] * M[7] - M[1] * M[8]) / d; I[2] = (M[1] * M[5] - M[2] * M[4]) / d;
    I[3] = (M[5] * M[6] - M[3] * M[8]) / d; I[4] = (M[0] * M[8] - M[2] * M[6]) / d; I[5] = (M[2] * M[3] - M[0] * M[5]) / d;
    I[6] = (M[3] * M[7] - M[4] * M[6]) / d; I[7] = (M[1] * M[6] - M[0] * M[7]) / d; I[8] = (M[0] * M[4] - M[1] * M[3]) / d;
    return true;
}

// Linear start of the pose (what stands in for the first stage of cv2.solvePnP's ITERATIVE flag): DLT on normalised
// image points with conditioned object points, the 3x3 block projected onto SO(3).  false on degenerate input.
bool dlt_pose(const double* x2, const double* x3, const double* K, int pn, double* R, double* t) {
    double Ki[9];
    if (pn < 6 || !inverse3(K, Ki)) return false;
    double c[3] = {0, 0, 0};
    for (int i = 0; i < pn; ++i)
        for (int a = 0; a < 3; ++a) c[a] += x3[3 * i + a] / pn;
    double s = 0;
    for (int i = 0; i < pn; ++i)
        for (int a = 0; a < 3; ++a) s += (x3[3 * i + a] - c[a]) * (x3[3 * i + a] - c[a]);
    s = std::sqrt(s / pn) + 1e-12;
    double M[144];  // A^T A of the 2 pn x 12 DLT system
    std::memset(M, 0, sizeof(M));
    for (int i = 0; i < pn; ++i) {
        const double u = x2[2 * i], v = x2[2 * i + 1];
        const double zn = Ki[6] * u + Ki[7] * v + Ki[8];
        const double xn = (Ki[0] * u + Ki[1] * v + Ki[2]) / zn, yn = (Ki[3] * u + Ki[4] * v + Ki[5]) / zn;
        const double Xh[4] = {(x3[3 * i] - c[0]) / s, (x3[3 * i + 1] - c[1]) / s, (x3[3 * i + 2] - c[2]) / s, 1.0};
        double r0[12], r1[12];
        for (int a = 0; a < 4; ++a) {
            r0[a] = Xh[a]; r0[4 + a] = 0; r0[8 + a] = -xn * Xh[a];
            r1[a] = 0; r1[4 + a] = Xh[a]; r1[8 + a] = -yn * Xh[a];
        }
        for (int a = 0; a < 12; ++a)
            for (int b = 0; b < 12; ++b) M[a * 12 + b] += r0[a] * r0[b] + r1[a] * r1[b];
    }
    double V[144], w[12];
    jacobi_eigen(M, 12, V, w);
    int kmin = 0;
    for (int i = 1; i < 12; ++i)
        if (w[i] < w[kmin]) kmin = i;
    double P[12];
    for (int i = 0; i < 12; ++i) P[i] = V[i * 12 + kmin];
    double P3[9] = {P[0], P[1], P[2], P[4], P[5], P[6], P[8], P[9], P[10]};
    if (det3(P3) < 0) {
        for (double& v : P) v = -v;
        for (double& v : P3) v = -v;
    }
    // SVD of the 3x3 block through the eigen-decomposition of P3^T P3 = V S^2 V^T;  R = U V^T = P3 V S^-1 V^T
    double G[9], W3[9], e[3];
    for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 3; ++b) G[a * 3 + b] = P3[a] * P3[b] + P3[3 + a] * P3[3 + b] + P3[6 + a] * P3[6 + b];
    jacobi_eigen(G, 3, W3, e);
    double sv[3], smean = 0;
    for (int a = 0; a < 3; ++a) {
        if (!(e[a] > 1e-300)) return false;
        sv[a] = std::sqrt(e[a]);
        smean += sv[a] / 3.0;
    }
    double T[9];  // V S^-1 V^T
    for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 3; ++b)
            T[a * 3 + b] = W3[a * 3] * W3[b * 3] / sv[0] + W3[a * 3 + 1] * W3[b * 3 + 1] / sv[1] + W3[a * 3 + 2] * W3[b * 3 + 2] / sv[2];
    matmul3(P3, T, R);
    if (det3(R) < 0)
        for (int a = 0; a < 9; ++a) R[a] = -R[a];
    for (int a = 0; a < 3; ++a) {  // undo the conditioning X = (Xw - c) / s
        const double tp = P[4 * a + 3] / smean - (R[a * 3] * c[0] + R[a * 3 + 1] * c[1] + R[a * 3 + 2] * c[2]) / s;
        t[a] = tp * s;
    }
    return std::isfinite(t[0]) && std::isfinite(t[1]) && std::isfinite(t[2]);
}

}  // namespace

extern "C" {

int pvnet_pnp_evaluate(const double* pts2d, const double* pts3d, const double* wgt2d, const double* K, const double* rt,
                       int pn, double* residuals, double* jacobian) {
    if (!pts2d || !pts3d || !K || !rt || !residuals || pn < 1 || pn > MAX_PN) return -1;
    const Problem P{pts2d, pts3d, wgt2d, K, pn};
    return evaluate(P, rt, residuals, jacobian) ? 0 : 1;
}

int pvnet_pnp_refine(const double* pts2d, const double* pts3d, const double* wgt2d, const double* K,
                     const double* init_rt, double* result_rt, int pn, int max_iterations, double* final_cost) {
    if (!pts2d || !pts3d || !K || !init_rt || !result_rt || pn < 3 || pn > MAX_PN) return -1;
    if (max_iterations <= 0) max_iterations = 100;
    const Problem P{pts2d, pts3d, wgt2d, K, pn};
    double* r = new double[2 * pn];
    double* rn = new double[2 * pn];
    double* J = new double[12 * pn];
    double p[6], pnw[6];
    std::memcpy(p, init_rt, sizeof(p));
    std::memcpy(result_rt, init_rt, sizeof(p));
    auto cost_of = [&](const double* res) {
        double c = 0;
        for (int i = 0; i < 2 * pn; ++i) c += res[i] * res[i];
        return 0.5 * c;
    };
    int it = 0;
    double cost = 0.0;
    if (evaluate(P, p, r, J)) {
        cost = cost_of(r);
        double lambda = 1e-4, nu = 2.0;  // Marquardt damping on diag(J^T J), Nielsen's update
        for (; it < max_iterations; ++it) {
            double H[36], g[6];
            for (int a = 0; a < 6; ++a) {
                g[a] = 0;
                for (int i = 0; i < 2 * pn; ++i) g[a] -= J[i * 6 + a] * r[i];
                for (int b = 0; b <= a; ++b) {
                    double s = 0;
                    for (int i = 0; i < 2 * pn; ++i) s += J[i * 6 + a] * J[i * 6 + b];
                    H[a * 6 + b] = H[b * 6 + a] = s;
                }
            }
            double gmax = 0;
            for (int a = 0; a < 6; ++a) gmax = std::fmax(gmax, std::fabs(g[a]));
            if (gmax < 1e-14) break;  // stationary
            bool stepped = false, tiny = false;
            for (int tries = 0; tries < 40 && !stepped; ++tries) {
                double A[36], d[6];
                std::memcpy(A, H, sizeof(A));
                for (int a = 0; a < 6; ++a) A[a * 6 + a] += lambda * (H[a * 6 + a] > 1e-300 ? H[a * 6 + a] : 1.0);
                if (solve6(A, g, d)) {
                    double dn = 0, xn = 0;
                    for (int a = 0; a < 6; ++a) { pnw[a] = p[a] + d[a]; dn += d[a] * d[a]; xn += p[a] * p[a]; }
                    if (std::sqrt(dn) <= 1e-15 * (std::sqrt(xn) + 1e-15)) { tiny = true; break; }
                    if (evaluate(P, pnw, rn, nullptr)) {
                        const double cn = cost_of(rn);
                        double pred = 0;  // predicted decrease 0.5 d^T (lambda D d + g)
                        for (int a = 0; a < 6; ++a)
                            pred += 0.5 * d[a] * (lambda * (H[a * 6 + a] > 1e-300 ? H[a * 6 + a] : 1.0) * d[a] + g[a]);
                        const double rho = pred > 0 ? (cost - cn) / pred : -1.0;
                        if (cn < cost && rho > 0) {
                            std::memcpy(p, pnw, sizeof(p));
                            const double rel = (cost - cn) / (cost > 1e-300 ? cost : 1e-300);
                            cost = cn;
                            const double t = 2.0 * rho - 1.0;
                            lambda *= std::fmax(1.0 / 3.0, 1.0 - t * t * t);
                            nu = 2.0;
                            stepped = true;
                            if (rel < 1e-16) tiny = true;
                            continue;
                        }
                    }
                }
                lambda *= nu;
                nu *= 2.0;
            }
            if (!stepped || tiny) break;
            if (!evaluate(P, p, r, J)) break;
        }
        std::memcpy(result_rt, p, sizeof(p));
    }
    if (final_cost) *final_cost = cost;
    delete[] r;
    delete[] rn;
    delete[] J;
    return it;
}

void uncertainty_pnp(double* pts2d, double* pts3d, double* wgt2d, double* K, double* init_rt, double* result_rt,
                     int pn) {
    if (pvnet_pnp_refine(pts2d, pts3d, wgt2d, K, init_rt, result_rt, pn, 100, nullptr) < 0 && init_rt && result_rt)
        std::memcpy(result_rt, init_rt, 6 * sizeof(double));
}

int pvnet_pnp_solve(const double* pts2d, const double* pts3d, const double* wgt2d, const double* K, double* result_rt,
                    int pn) {
    if (!pts2d || !pts3d || !K || !result_rt || pn < 6 || pn > MAX_PN) return -1;
    double R[9], t[3], x0[6];
    if (!dlt_pose(pts2d, pts3d, K, pn, R, t)) return -2;
    pvnet_matrix_to_angle_axis(R, x0);
    x0[3] = t[0]; x0[4] = t[1]; x0[5] = t[2];
    int it = pvnet_pnp_refine(pts2d, pts3d, nullptr, K, x0, result_rt, pn, 200, nullptr);
    if (it >= 0 && wgt2d) {  // uncertainty-driven PnP: the weighted problem from the unweighted optimum
        std::memcpy(x0, result_rt, sizeof(x0));
        const int it2 = pvnet_pnp_refine(pts2d, pts3d, wgt2d, K, x0, result_rt, pn, 200, nullptr);
        it = it2 < 0 ? it2 : it + it2;
    }
    return it;
}

// The poses of a batch are independent (each is ~35 us of dense 6x6 LM): from eight images on they are shared out over a few
// threads -- contiguous blocks, so the result of every image is what the serial loop computes (tests/test_pnp.py) -- replacing
// the serial loop of tools/train_linemod.py:210-218.  PVNET_PNP_THREADS sets the number (default: min(8, hardware threads / 2), at least four poses per thread).
static int solve_range(const double* pts2d, const double* pts3d, const double* wgt2d, const double* K, double* result_rt, int i0,
                       int i1, int pn, int* bad_args) {
    int bad = 0;
    for (int i = i0; i < i1; ++i) {
        const int rc = pvnet_pnp_solve(pts2d + (size_t)i * pn * 2, pts3d, wgt2d ? wgt2d + (size_t)i * pn * 3 : nullptr, K,
                                       result_rt + (size_t)i * 6, pn);
        if (rc == -1) { *bad_args = 1; return bad; }
        if (rc < 0) {
            ++bad;
            for (int a = 0; a < 6; ++a) result_rt[(size_t)i * 6 + a] = 0.0;
        }
    }
    return bad;
}

int pvnet_pnp_solve_batch(const double* pts2d, const double* pts3d, const double* wgt2d, const double* K,
                          double* result_rt, int n, int pn) {
    if (n < 0) return -1;
    int nt = (int)std::thread::hardware_concurrency() / 2;   // (half the hardware threads: with all eight of an 8-thread host the batch
    if (nt > 8) nt = 8;                                      //  of 32 took 0.8-1.6 ms against 0.66-0.79 ms on four and 1.45-1.7 ms serial)
    if (const char* e = std::getenv("PVNET_PNP_THREADS")) nt = std::atoi(e);
    if (nt > n / 4) nt = n / 4;   // at least four poses per thread: a thread costs about one pose to start
    if (nt <= 1) {
        int bad_args = 0;
        const int bad = solve_range(pts2d, pts3d, wgt2d, K, result_rt, 0, n, pn, &bad_args);
        return bad_args ? -1 : bad;
    }
    std::vector<int> bad(nt, 0), bad_args(nt, 0);
    std::vector<std::thread> th;
    th.reserve(nt - 1);
    auto work = [&](int t) {
        const int i0 = (int)((long long)n * t / nt), i1 = (int)((long long)n * (t + 1) / nt);
        bad[t] = solve_range(pts2d, pts3d, wgt2d, K, result_rt, i0, i1, pn, &bad_args[t]);
    };
    for (int t = 1; t < nt; ++t) th.emplace_back(work, t);
    work(0);
    for (auto& x : th) x.join();
    int total = 0;
    for (int t = 0; t < nt; ++t) {
        if (bad_args[t]) return -1;
        total += bad[t];
    }
    return total;
}

void pvnet_angle_axis_to_matrix(const double* aa, double* R) { rotation_and_right_jacobian(aa, R, nullptr); }

// [n,6] (angle-axis | translation) -> [n,3,4] (R | t); a failed image's zero row gives a zero pose, as the reference's callers expect
void pvnet_pnp_poses_from_rt(const double* rt, double* poses, int n) {
    for (int i = 0; i < n; ++i) {
        const double* o = rt + (size_t)i * 6;
        double* P = poses + (size_t)i * 12;
        bool any = false;
        for (int a = 0; a < 6; ++a) any = any || o[a] != 0.0;
        if (!any) {
            for (int a = 0; a < 12; ++a) P[a] = 0.0;
            continue;
        }
        double R[9];
        rotation_and_right_jacobian(o, R, nullptr);
        for (int r = 0; r < 3; ++r) {
            P[r * 4 + 0] = R[r * 3 + 0];
            P[r * 4 + 1] = R[r * 3 + 1];
            P[r * 4 + 2] = R[r * 3 + 2];
            P[r * 4 + 3] = o[3 + r];
        }
    }
}

void pvnet_matrix_to_angle_axis(const double* R, double* aa) {
    // robust log map: quaternion first (largest-component branch), then angle-axis
    double q[4];  // w x y z
    const double tr = R[0] + R[4] + R[8];
    if (tr > 0) {
        const double s = std::sqrt(tr + 1.0) * 2;
        q[0] = 0.25 * s; q[1] = (R[7] - R[5]) / s; q[2] = (R[2] - R[6]) / s; q[3] = (R[3] - R[1]) / s;
    } else if (R[0] > R[4] && R[0] > R[8]) {
        const double s = std::sqrt(1.0 + R[0] - R[4] - R[8]) * 2;
        q[0] = (R[7] - R[5]) / s; q[1] = 0.25 * s; q[2] = (R[1] + R[3]) / s; q[3] = (R[2] + R[6]) / s;
    } else if (R[4] > R[8]) {
        const double s = std::sqrt(1.0 + R[4] - R[0] - R[8]) * 2;
        q[0] = (R[2] - R[6]) / s; q[1] = (R[1] + R[3]) / s; q[2] = 0.25 * s; q[3] = (R[5] + R[7]) / s;
    } else {
        const double s = std::sqrt(1.0 + R[8] - R[0] - R[4]) * 2;
        q[0] = (R[3] - R[1]) / s; q[1] = (R[2] + R[6]) / s; q[2] = (R[5] + R[7]) / s; q[3] = 0.25 * s;
    }
    if (q[0] < 0) for (double& v : q) v = -v;
    const double sn = std::sqrt(q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    const double k = sn < 1e-12 ? 2.0 : 2.0 * std::atan2(sn, q[0]) / sn;
    aa[0] = k * q[1]; aa[1] = k * q[2]; aa[2] = k * q[3];
}

// ---- farthest-point sampling (lib/utils/extend_utils/src/farthest_point_sampling.cpp:41-176): the reference picks its 8
// object key-points with it (lib/utils/data_utils.py:144: farthest_point_sampling(pts, 8, True)).  Greedy: every point keeps
// the squared distance (float32, x*x + y*y + z*z in that order) to the nearest point selected so far; the next selection is
// the unselected point with the LARGEST such distance, the first one on ties (strict >, from index 0, start value 0).
static int fps_next(const float* min_dist, const unsigned char* taken, int pn) {
    int best = 0;
    float best_d = 0.f;
    for (int i = 0; i < pn; ++i)
        if (!taken[i] && min_dist[i] > best_d) {
            best = i;
            best_d = min_dist[i];
        }
    return best;
}

static void fps_run(const float* pts, int* idxs, int pn, int sn, float* min_dist, unsigned char* taken, int cur) {
    for (int s = 0; s < sn; ++s) {
        taken[cur] = 1;
        idxs[s] = cur;
        if (s == sn - 1) break;
        const float cx = pts[cur * 3], cy = pts[cur * 3 + 1], cz = pts[cur * 3 + 2];
        for (int i = 0; i < pn; ++i) {
            if (taken[i]) continue;
            const float dx = pts[i * 3] - cx, dy = pts[i * 3 + 1] - cy, dz = pts[i * 3 + 2] - cz;
            const float d = dx * dx + dy * dy + dz * dz;
            if (d < min_dist[i]) min_dist[i] = d;
        }
        cur = fps_next(min_dist, taken, pn);
    }
}

// starts from a random point (the reference seeds rand() with the wall clock: farthest_point_sampling.cpp:96-97)
void farthest_point_sampling(float* pts, int* idxs, int pn, int sn) {
    if (!pts || !idxs || pn <= 0 || sn <= 0) return;
    float* min_dist = new float[pn];
    unsigned char* taken = new unsigned char[pn]();
    for (int i = 0; i < pn; ++i) min_dist[i] = FLT_MAX;
    std::srand((unsigned)std::time(nullptr));
    fps_run(pts, idxs, pn, sn, min_dist, taken, std::rand() % pn);
    delete[] min_dist;
    delete[] taken;
}

// deterministic: distances start as those to the centre of the bounding box, the first point is the one farthest from it
// (farthest_point_sampling.cpp:124-160)
void farthest_point_sampling_init_center(float* pts, int* idxs, int pn, int sn) {
    if (!pts || !idxs || pn <= 0 || sn <= 0) return;
    float lo[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, hi[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    for (int i = 0; i < pn; ++i)
        for (int a = 0; a < 3; ++a) {
            hi[a] = pts[i * 3 + a] > hi[a] ? pts[i * 3 + a] : hi[a];
            lo[a] = pts[i * 3 + a] < lo[a] ? pts[i * 3 + a] : lo[a];
        }
    const float half = 1.f / 2.f;  // (max + min) / 2 as a multiplication by the reciprocal, like the reference's operator/
    const float cx = (hi[0] + lo[0]) * half, cy = (hi[1] + lo[1]) * half, cz = (hi[2] + lo[2]) * half;
    float* min_dist = new float[pn];
    unsigned char* taken = new unsigned char[pn]();
    for (int i = 0; i < pn; ++i) {
        const float dx = pts[i * 3] - cx, dy = pts[i * 3 + 1] - cy, dz = pts[i * 3 + 2] - cz;
        min_dist[i] = dx * dx + dy * dy + dz * dz;
    }
    fps_run(pts, idxs, pn, sn, min_dist, taken, fps_next(min_dist, taken, pn));
    delete[] min_dist;
    delete[] taken;
}

}  // extern "C"
