/*
 * ba_oracle.c -- TEST INFRASTRUCTURE ONLY (CPU oracle), never the product path.
 *
 * CPU restatement (plain C, single thread, like the reference) of the bundle-adjustment LM hot path of
 * jasvob/BundleAdjustment_Benchmarks; see ba_oracle_impl.h for the per-function citations.
 * PARITY UNPINNED (no reference tests / golden output exist, reference not buildable offline).
 *
 * Build: make -C oracle   ->  oracle/libba_oracle.so
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ---- BAL text loader: bundle_adjustment_large.cpp:59-107 ------------------------------- */
/* Line 1 "N M K"; K lines "cam pt u v"; 9N scalars (omega(3), T(3), f, k1, k2 per camera); 3M scalars.
 * Like the reference's `ifs >>` chain this is whitespace-delimited token parsing. */

int ora_bal_header(const char *path, int *N, int *M, int *K)
{
    FILE *f = fopen(path, "r");
    if (!f) return 2; /* ReturnCodes::WrongInputFile, bundle_adjustment_large.cpp:26-31,50-54 */
    int rc = (fscanf(f, "%d %d %d", N, M, K) == 3) ? 0 : 3;
    fclose(f);
    return rc;
}

int ora_bal_read(const char *path, int N, int M, int K, int *cam_idx, int *pt_idx, double *meas, double *cams9,
                 double *pts)
{
    FILE *f = fopen(path, "r");
    if (!f) return 2;
    int n, m, k;
    if (fscanf(f, "%d %d %d", &n, &m, &k) != 3 || n != N || m != M || k != K) { fclose(f); return 3; }
    for (int i = 0; i < K; i++)
        if (fscanf(f, "%d %d %lf %lf", &cam_idx[i], &pt_idx[i], &meas[2 * (size_t)i], &meas[2 * (size_t)i + 1]) != 4) {
            fclose(f);
            return 3;
        }
    for (size_t i = 0; i < 9 * (size_t)N; i++)
        if (fscanf(f, "%lf", &cams9[i]) != 1) { fclose(f); return 3; }
    for (size_t i = 0; i < 3 * (size_t)M; i++)
        if (fscanf(f, "%lf", &pts[i]) != 1) { fclose(f); return 3; }
    fclose(f);
    return 0;
}

/* Threads of the OpenMP loops (the dense factorisation's trailing update): 1 = the reference's own single-threaded regime. */
#if defined(_OPENMP)
#include <omp.h>
void ora_set_threads(int n) { omp_set_num_threads(n < 1 ? 1 : n); }
int ora_max_threads(void) { return omp_get_max_threads(); }
#else
void ora_set_threads(int n) { (void)n; }
int ora_max_threads(void) { return 1; }
#endif

#define CAT_(a, b) a##b
#define CAT(a, b) CAT_(a, b)

#define S double
#define FN(name) CAT(name, _f64)
#define MINNORMAL 2.2250738585072014e-308 /* DBL_MIN */
#define SQRT sqrt
#define FABS fabs
#define SIN sin
#define COS cos
#define POW pow
#include "ba_oracle_impl.h"
#undef S
#undef FN
#undef MINNORMAL
#undef SQRT
#undef FABS
#undef SIN
#undef COS
#undef POW

#define S float
#define FN(name) CAT(name, _f32)
#define MINNORMAL 1.17549435e-38f /* FLT_MIN */
#define SQRT sqrtf
#define FABS fabsf
#define SIN sinf
#define COS cosf
#define POW powf
#include "ba_oracle_impl.h"
