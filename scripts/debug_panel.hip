// Dev tool: first panel of k_ldlt_panel vs a CPU LDL^T (max errors of L, D, W, Y).
#include "ba_dense.hip.h"
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
int main(int argc, char **argv)
{
    const int D = argc > 1 ? atoi(argv[1]) : 200, NB = 64, Dp = ((D + 3 + NB - 1) / NB) * NB, ld = Dp + 64;
    std::vector<double> h((size_t)ld * (Dp + 64), 0.0), v((size_t)D * 8);
    srand(1);
    for (auto &x : v) x = rand() / (double)RAND_MAX - 0.5;
    for (int c = 0; c < D; c++)
        for (int r = c; r < D; r++) {
            double a = (r == c) ? 4.0 : 0.0;
            for (int k = 0; k < 8; k++) a += v[(size_t)r * 8 + k] * v[(size_t)c * 8 + k];
            h[(size_t)c * ld + r] = a;
        }
    for (int c = 0; c < D; c++) h[(size_t)c * ld + D] = rand() / (double)RAND_MAX;
    for (int c = D; c < Dp; c++) h[(size_t)c * ld + c] = 1.0;
    const int nrows = D + 1, ncols = D, nb = std::min(NB, ncols);
    // CPU: factor first nb columns (right-looking on the first block column only)
    std::vector<double> A = h;
    auto at = [&](int r, int c) -> double & { return A[(size_t)c * ld + r]; };
    for (int k = 0; k < nb; k++) {
        const double d = at(k, k);
        for (int i = k + 1; i < nrows; i++) {
            const double y = at(i, k);
            for (int j = k + 1; j <= std::min(i, nb - 1); j++) at(i, j) -= y * at(j, k) / d * 1.0; // uses y_j / d = l_j? careful below
        }
        // proper: l_i = y_i / d; a_ij -= l_i * y_j  with y_j = original column entry
    }
    // redo properly
    A = h;
    std::vector<double> Y((size_t)nrows * nb, 0.0);
    for (int k = 0; k < nb; k++) {
        const double d = at(k, k);
        std::vector<double> ycol(nrows);
        for (int i = k + 1; i < nrows; i++) ycol[i] = at(i, k);
        for (int i = k + 1; i < nrows; i++) {
            const double l = ycol[i] / d;
            for (int j = k + 1; j < nb && j <= i; j++) at(i, j) -= l * ycol[j];
            Y[(size_t)k * nrows + i] = ycol[i];
            at(i, k) = l;
        }
    }
    double *S, *Wp, *Winv;
    hipMalloc(&S, sizeof(double) * h.size()); hipMalloc(&Wp, sizeof(double) * (size_t)ld * NB); hipMalloc(&Winv, sizeof(double) * NB * NB);
    hipMemcpy(S, h.data(), sizeof(double) * h.size(), hipMemcpyHostToDevice);
    hipMemset(Wp, 0, sizeof(double) * (size_t)ld * NB);
    const int below = nrows - NB, g = below > 0 ? (below + 63) / 64 : 1;
    hipLaunchKernelGGL((k_ldlt_panel<double, 64>), dim3(g), dim3(256), 0, 0, nrows, ncols, ld, 0, S, Wp, Winv);
    hipError_t e = hipDeviceSynchronize();
    printf("launch: %s / %s\n", hipGetErrorString(hipGetLastError()), hipGetErrorString(e));
    std::vector<double> g_(h.size()), gw((size_t)ld * NB), gi(NB * NB);
    hipMemcpy(g_.data(), S, sizeof(double) * h.size(), hipMemcpyDeviceToHost);
    hipMemcpy(gw.data(), Wp, sizeof(double) * gw.size(), hipMemcpyDeviceToHost);
    hipMemcpy(gi.data(), Winv, sizeof(double) * gi.size(), hipMemcpyDeviceToHost);
    double eD = 0, eL1 = 0, eL2 = 0, eY = 0; int wi = -1, wj = -1;
    for (int c = 0; c < nb; c++) {
        eD = std::max(eD, std::fabs(g_[(size_t)c * ld + c] - at(c, c)));
        for (int r = c + 1; r < nrows; r++) {
            const double d = std::fabs(g_[(size_t)c * ld + r] - at(r, c));
            if (r < NB) { if (d > eL1) { eL1 = d; wi = r; wj = c; } } else { eL2 = std::max(eL2, d); eY = std::max(eY, std::fabs(gw[(size_t)c * ld + r] - Y[(size_t)c * nrows + r])); }
        }
    }
    // W check: W * L11 = I
    double eW = 0;
    for (int i = 0; i < nb; i++)
        for (int j = 0; j <= i; j++) {
            double a = 0;
            for (int k = j; k <= i; k++) a += gi[i * NB + k] * (k == j ? 1.0 : at(k, j));
            eW = std::max(eW, std::fabs(a - (i == j ? 1.0 : 0.0)));
        }
    printf("D=%d nb=%d: max err D %.2e, L(diag block) %.2e at (%d,%d), L(below) %.2e, Y(below) %.2e, |W L11 - I| %.2e\n", D, nb, eD, eL1, wi, wj, eL2, eY, eW);
    for (int c = 0; c < 20 && c < nb; c++) printf("  d[%d] gpu %.6f cpu %.6f | L[%d+1][%d] gpu %.6f cpu %.6f\n", c, g_[(size_t)c * ld + c], at(c, c), c, c, g_[(size_t)c * ld + c + 1], at(c + 1, c));
    return 0;
}
