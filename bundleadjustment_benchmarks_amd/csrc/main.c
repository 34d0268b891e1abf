/*
 * main.c -- drop-in for src/bundle_adjustment_large.cpp of jasvob/BundleAdjustment_Benchmarks: same command line
 * (`<exe> <sparse reconstruction file>`), exit codes (:26-31), stdout protocol (:61-171) and one executable per
 * solver symbol (-DQRKIT / -DQRCHOL / -DCHOLESKY / -DMOREQR / -DQRSPQR, src/CMakeLists.txt:95-178); -DBA_SCALAR_FLOAT stands for
 * `typedef float Scalar;` (src/BATypeUtils.h:6-7).  All work happens behind the C ABI of include/ba_mi355x.h.
 * Extensions: the environment variable BA_MAX_TRIALS bounds the number of LM table rows (benchmarking);
 * BA_CACHE=1 keeps a binary cache `<file>.bacache` of the parsed problem and reuses it when present;
 * BA_WORLD / BA_RANK (one process per GPU of a node, started by hand or by a launcher; BA_DEVICE defaults to BA_RANK) shard the
 * points over the processes, which meet through the file BA_COMM_FILE (default /tmp/ba_mi355x_comm.id, written by rank 0 and removed
 * again once the communicator stands; BA_COMM_NONCE, any string the launcher sets per launch, ties the file to this launch) and
 * then all-reduce the reduced camera system over RCCL inside the library; rank 0 prints.
 */
#define _POSIX_C_SOURCE 199309L
#include "../../include/ba_mi355x.h"

#include <stdio.h>
#include <stdlib.h>
#include <time.h>

#if defined(QRKIT)
#define BA_KIND BA_QRKIT
#elif defined(QRCHOL)
#define BA_KIND BA_QRCHOL
#elif defined(CHOLESKY)
#define BA_KIND BA_CHOLESKY
#elif defined(MOREQR)
#define BA_KIND BA_MOREQR
#elif defined(QRSPQR)
#define BA_KIND BA_QRSPQR
#else
#error "define one of QRKIT, QRCHOL, CHOLESKY, MOREQR, QRSPQR"
#endif

#ifdef BA_SCALAR_FLOAT
#define BA_SCALAR BA_F32
#else
#define BA_SCALAR BA_F64
#endif

/* Utils::showErrorStatistics + Utils::showObjective, src/Utils.h:39-40,65 */
static void show_stats(ba_solver *s, int K)
{
    double st[4];
    if (ba_solver_stats(s, st) != BA_OK) return;
    printf("Mean reprojection error: %g\n", st[0]);
    printf("Inlier mean reprojection error: %g (%d / %d inliers)\n", st[1], (int)st[2], K);
    printf("True objective: %g\n", st[3]);
}

int main(int argc, char *argv[])
{
    if (argc != 2) {
        fprintf(stderr, "Usage: %s <sparse reconstruction file>\n", argv[0]);
        return BA_ERR_USAGE;
    }
    ba_problem *p = NULL;
    int rc = -1;
    char cache[4096];
    const int use_cache = getenv("BA_CACHE") != NULL && snprintf(cache, sizeof cache, "%s.bacache", argv[1]) < (int)sizeof cache;
    if (use_cache) rc = ba_problem_load_cache(cache, &p);
    if (rc != BA_OK) {
        rc = ba_problem_load_bal(argv[1], &p);
        if (rc == BA_OK && use_cache) (void)ba_problem_save_cache(p, cache);
    }
    if (rc == BA_ERR_FILE) {
        fprintf(stderr, "Cannot open %s\n", argv[1]);
        return BA_ERR_FILE;
    }
    if (rc != BA_OK) {
        fprintf(stderr, "Cannot parse %s: %s\n", argv[1], ba_error_string(rc));
        return rc;
    }
    int N, M, K;
    ba_problem_dims(p, &N, &M, &K);
    const int world = getenv("BA_WORLD") ? atoi(getenv("BA_WORLD")) : 1, rank = getenv("BA_RANK") ? atoi(getenv("BA_RANK")) : 0;
    if (world < 1 || rank < 0 || rank >= world) {
        fprintf(stderr, "BA_WORLD / BA_RANK: need 0 <= rank < world\n");
        ba_problem_free(p);
        return BA_ERR_USAGE;
    }
    const int talk = rank == 0;
    if (talk) {
        printf("N(cameras) = %d, M(points) = %d, K(measurements) = %d\n", N, M, K);
        printf("Reading image measurements...\nDone.\n");
        printf("Reading cameras params...\nDone.\n");
        printf("Reading 3D points...\nDone.\n");
    }

    ba_solver *s = NULL;
    const int device = getenv("BA_DEVICE") ? atoi(getenv("BA_DEVICE")) : (world > 1 ? rank : -1);
    rc = ba_solver_create(p, BA_KIND, BA_SCALAR, device, rank, world, &s);
    if (rc != BA_OK) {
        fprintf(stderr, "ba_solver_create: %s\n", ba_error_string(rc));
        ba_problem_free(p);
        return rc;
    }
    if (world > 1) { /* RCCL inside the library; the communicator id travels through a file */
        unsigned char id[BA_COMM_ID_BYTES];
        const char *idf = getenv("BA_COMM_FILE") ? getenv("BA_COMM_FILE") : "/tmp/ba_mi355x_comm.id";
        rc = ba_comm_id_via_file(idf, rank, id);
        if (rc == BA_OK) rc = ba_solver_comm_init(s, id);
        (void)ba_comm_id_file_done(idf, rank); /* the communicator stands (or never will): the id must not outlive this launch */
        if (rc != BA_OK) {
            fprintf(stderr, "communicator set-up (rank %d of %d, %s): %s\n", rank, world, idf, ba_error_string(rc));
            ba_solver_free(s);
            ba_problem_free(p);
            return rc;
        }
    }
    if (talk) show_stats(s, K); else { double st_[4]; (void)ba_solver_stats(s, st_); } /* (the statistics are a collective) */

    ba_lm_params lm;
    ba_lm_params_default(&lm);
    lm.verbose = talk;
    if (getenv("BA_MAX_TRIALS")) lm.max_trials = atoi(getenv("BA_MAX_TRIALS"));
    ba_result res;
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    rc = ba_minimize(s, &lm, NULL, NULL, &res);
    clock_gettime(CLOCK_MONOTONIC, &t1);
    if (rc != BA_OK) {
        fprintf(stderr, "ba_minimize: %s\n", ba_error_string(rc));
        ba_solver_free(s);
        ba_problem_free(p);
        return rc;
    }
    if (talk) {
        printf("lm.minimize(params) ... %gs\n", (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec));
        printf("LM finished with status: %s\n", ba_status_string(res.status));
        show_stats(s, K);
    } else {
        double st_[4];
        (void)ba_solver_stats(s, st_);
    }
    ba_solver_free(s);
    ba_problem_free(p);
    return BA_OK; /* success even if the LM status is not Success, like the reference (:175) */
}
