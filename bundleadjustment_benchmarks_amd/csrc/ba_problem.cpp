// ba_problem.cpp -- host side of the boundary: BAL text loader, writer and the seeded synthetic generator.
//
// Replaces the read loop of the reference driver (src/bundle_adjustment_large.cpp:59-107).  The file format is kept:
//   line 1 "N M K"; K lines "cam pt u v"; 9N scalars (omega(3), T(3), f, k1, k2 per camera); 3M point scalars.
// The reference parses with `ifstream >>`; here the file is read whole and tokenised with strtol/strtod
// (whitespace-delimited tokens, the same grammar) so that a 160 MB synthetic problem loads in about a second.
#include "ba_internal.h"

#include <cerrno>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <numeric>
#include <random>

extern "C" {

const char *ba_version(void) { return "ba_mi355x 0.1 (gfx950)"; }

const char *ba_error_string(int err)
{
    switch (err) {
    case BA_OK: return "ok";
    case BA_ERR_USAGE: return "wrong input parameters";
    case BA_ERR_FILE: return "cannot open input file";
    case BA_ERR_PARSE: return "malformed BAL file";
    case BA_ERR_ARG: return "invalid argument";
    case BA_ERR_HIP: return "HIP runtime error / no MI355X device";
    case BA_ERR_NOMEM: return "out of memory";
    case BA_ERR_COMM: return "communication (RCCL / all-reduce callback) failed";
    }
    return "unknown error";
}

// statusToString, src/Eigen_ext/BacktrackLevMarqQRChol.h:48-63
const char *ba_status_string(int status)
{
    switch (status) {
    case BA_NOT_STARTED: return "Not Started";
    case BA_RUNNING: return "Running";
    case BA_SUCCESS: return "Success (Energy Flatlined)";
    case BA_EXCEEDED_LAMBDA_MAX: return "Success (Exceeded Maximum Lambda)";
    case BA_TOO_MANY_FUN_EVALS: return "Too Many Function Evaluations";
    case BA_MAX_ITERS: return "Maximum Iterations Reached";
    }
    return "Unknown";
}

static int validate(const ba_problem *p)
{
    if (p->N <= 0 || p->M <= 0 || p->K <= 0) return BA_ERR_ARG;
    for (int i = 0; i < p->K; i++)
        if (p->cam_idx[i] < 0 || p->cam_idx[i] >= p->N || p->pt_idx[i] < 0 || p->pt_idx[i] >= p->M) return BA_ERR_PARSE;
    // strtod accepts "nan" / "inf" tokens; a non-finite measurement or parameter would poison every sum on the device
    for (double v : p->meas) if (!std::isfinite(v)) return BA_ERR_PARSE;
    for (double v : p->cams9) if (!std::isfinite(v)) return BA_ERR_PARSE;
    for (double v : p->pts) if (!std::isfinite(v)) return BA_ERR_PARSE;
    return BA_OK;
}

int ba_problem_load_bal(const char *path, ba_problem **out)
{
    if (!path || !out) return BA_ERR_ARG;
    *out = nullptr;
    FILE *f = fopen(path, "rb");
    if (!f) return BA_ERR_FILE;
    fseek(f, 0, SEEK_END);
    long sz = ftell(f);
    fseek(f, 0, SEEK_SET);
    if (sz <= 0) { fclose(f); return BA_ERR_PARSE; }
    std::vector<char> buf((size_t)sz + 1);
    size_t got = fread(buf.data(), 1, (size_t)sz, f);
    fclose(f);
    buf[got] = 0;
    char *c = buf.data(), *e = nullptr;
    auto next_int = [&](long &v) -> bool { errno = 0; v = strtol(c, &e, 10); if (e == c || errno) return false; c = e; return true; };
    auto next_dbl = [&](double &v) -> bool { errno = 0; v = strtod(c, &e); if (e == c) return false; c = e; return true; };
    long N, M, K;
    if (!next_int(N) || !next_int(M) || !next_int(K) || N <= 0 || M <= 0 || K <= 0 || N > (1 << 24) || M > (1L << 30) ||
        K > (1L << 30))
        return BA_ERR_PARSE;
    ba_problem *p = new (std::nothrow) ba_problem;
    if (!p) return BA_ERR_NOMEM;
    p->N = (int)N; p->M = (int)M; p->K = (int)K;
    p->cam_idx.resize(K); p->pt_idx.resize(K); p->meas.resize(2 * (size_t)K);
    p->cams9.resize(9 * (size_t)N); p->pts.resize(3 * (size_t)M);
    bool ok = true;
    for (long k = 0; k < K && ok; k++) {
        long a = 0, b = 0;
        ok = next_int(a) && next_int(b) && next_dbl(p->meas[2 * (size_t)k]) && next_dbl(p->meas[2 * (size_t)k + 1]);
        if (a < 0 || a >= N || b < 0 || b >= M) ok = false; // checked as long: 4294967296 must not wrap into a valid index
        p->cam_idx[k] = (int)a; p->pt_idx[k] = (int)b;
    }
    for (size_t i = 0; i < p->cams9.size() && ok; i++) ok = next_dbl(p->cams9[i]);
    for (size_t i = 0; i < p->pts.size() && ok; i++) ok = next_dbl(p->pts[i]);
    if (!ok || validate(p) != BA_OK) { delete p; return BA_ERR_PARSE; }
    *out = p;
    return BA_OK;
}

int ba_problem_create(int N, int M, int K, const int *cam_idx, const int *pt_idx, const double *meas,
                      const double *cams9, const double *pts, ba_problem **out)
{
    if (!out || !cam_idx || !pt_idx || !meas || !cams9 || !pts || N <= 0 || M <= 0 || K <= 0) return BA_ERR_ARG;
    ba_problem *p = new (std::nothrow) ba_problem;
    if (!p) return BA_ERR_NOMEM;
    p->N = N; p->M = M; p->K = K;
    p->cam_idx.assign(cam_idx, cam_idx + K);
    p->pt_idx.assign(pt_idx, pt_idx + K);
    p->meas.assign(meas, meas + 2 * (size_t)K);
    p->cams9.assign(cams9, cams9 + 9 * (size_t)N);
    p->pts.assign(pts, pts + 3 * (size_t)M);
    int rc = validate(p);
    if (rc) { delete p; return rc == BA_ERR_PARSE ? BA_ERR_ARG : rc; }
    *out = p;
    return BA_OK;
}

void ba_problem_free(ba_problem *p) { delete p; }

int ba_problem_dims(const ba_problem *p, int *N, int *M, int *K)
{
    if (!p) return BA_ERR_ARG;
    if (N) *N = p->N;
    if (M) *M = p->M;
    if (K) *K = p->K;
    return BA_OK;
}

int ba_problem_get(const ba_problem *p, int *cam_idx, int *pt_idx, double *meas, double *cams9, double *pts)
{
    if (!p) return BA_ERR_ARG;
    if (cam_idx) memcpy(cam_idx, p->cam_idx.data(), sizeof(int) * p->K);
    if (pt_idx) memcpy(pt_idx, p->pt_idx.data(), sizeof(int) * p->K);
    if (meas) memcpy(meas, p->meas.data(), sizeof(double) * 2 * (size_t)p->K);
    if (cams9) memcpy(cams9, p->cams9.data(), sizeof(double) * 9 * (size_t)p->N);
    if (pts) memcpy(pts, p->pts.data(), sizeof(double) * 3 * (size_t)p->M);
    return BA_OK;
}

int ba_problem_save_bal(const ba_problem *p, const char *path)
{
    if (!p || !path) return BA_ERR_ARG;
    FILE *f = fopen(path, "w");
    if (!f) return BA_ERR_FILE;
    fprintf(f, "%d %d %d\n", p->N, p->M, p->K);
    for (int k = 0; k < p->K; k++)
        fprintf(f, "%d %d     %.16e %.16e\n", p->cam_idx[k], p->pt_idx[k], p->meas[2 * (size_t)k], p->meas[2 * (size_t)k + 1]);
    for (size_t i = 0; i < p->cams9.size(); i++) fprintf(f, "%.16e\n", p->cams9[i]);
    for (size_t i = 0; i < p->pts.size(); i++) fprintf(f, "%.16e\n", p->pts[i]);
    fclose(f);
    return BA_OK;
}

static const char BA_CACHE_MAGIC[8] = {'B', 'A', 'L', 'c', 'a', 'c', 'h', '1'};

int ba_problem_save_cache(const ba_problem *p, const char *path)
{
    if (!p || !path) return BA_ERR_ARG;
    FILE *f = fopen(path, "wb");
    if (!f) return BA_ERR_FILE;
    const long long hdr[3] = {p->N, p->M, p->K};
    bool ok = fwrite(BA_CACHE_MAGIC, 1, 8, f) == 8 && fwrite(hdr, sizeof(long long), 3, f) == 3;
    ok = ok && fwrite(p->cam_idx.data(), sizeof(int), p->K, f) == (size_t)p->K && fwrite(p->pt_idx.data(), sizeof(int), p->K, f) == (size_t)p->K;
    ok = ok && fwrite(p->meas.data(), sizeof(double), p->meas.size(), f) == p->meas.size();
    ok = ok && fwrite(p->cams9.data(), sizeof(double), p->cams9.size(), f) == p->cams9.size();
    ok = ok && fwrite(p->pts.data(), sizeof(double), p->pts.size(), f) == p->pts.size();
    fclose(f);
    return ok ? BA_OK : BA_ERR_FILE;
}

int ba_problem_load_cache(const char *path, ba_problem **out)
{
    if (!path || !out) return BA_ERR_ARG;
    *out = nullptr;
    FILE *f = fopen(path, "rb");
    if (!f) return BA_ERR_FILE;
    char magic[8];
    long long hdr[3];
    if (fread(magic, 1, 8, f) != 8 || memcmp(magic, BA_CACHE_MAGIC, 8) != 0 || fread(hdr, sizeof(long long), 3, f) != 3 || hdr[0] <= 0 ||
        hdr[1] <= 0 || hdr[2] <= 0 || hdr[0] > (1 << 24) || hdr[1] > (1LL << 30) || hdr[2] > (1LL << 30)) {
        fclose(f);
        return BA_ERR_PARSE;
    }
    ba_problem *p = new (std::nothrow) ba_problem;
    if (!p) { fclose(f); return BA_ERR_NOMEM; }
    p->N = (int)hdr[0]; p->M = (int)hdr[1]; p->K = (int)hdr[2];
    p->cam_idx.resize(p->K); p->pt_idx.resize(p->K); p->meas.resize(2 * (size_t)p->K);
    p->cams9.resize(9 * (size_t)p->N); p->pts.resize(3 * (size_t)p->M);
    bool ok = fread(p->cam_idx.data(), sizeof(int), p->K, f) == (size_t)p->K && fread(p->pt_idx.data(), sizeof(int), p->K, f) == (size_t)p->K;
    ok = ok && fread(p->meas.data(), sizeof(double), p->meas.size(), f) == p->meas.size();
    ok = ok && fread(p->cams9.data(), sizeof(double), p->cams9.size(), f) == p->cams9.size();
    ok = ok && fread(p->pts.data(), sizeof(double), p->pts.size(), f) == p->pts.size();
    ok = ok && fgetc(f) == EOF;
    fclose(f);
    if (!ok || validate(p) != BA_OK) { delete p; return BA_ERR_PARSE; }
    *out = p;
    return BA_OK;
}

// ---- synthetic generator --------------------------------------------------------------------------------
// Cameras on a ring of radius 10 looking at the origin, BAL sign convention (P = R X + T, scene at negative camera z,
// K00 = -f).  std::mt19937_64's output sequence is fixed by the C++ standard; the distributions are written out here
// so that the same (N, M, K, seed) gives the same problem on every platform.
namespace {
struct Rng {
    std::mt19937_64 g;
    explicit Rng(unsigned long long s) : g(s) {}
    double uni() { return (double)(g() >> 11) * (1.0 / 9007199254740992.0); }
    double uni(double a, double b) { return a + (b - a) * uni(); }
    double normal()
    {
        double u1 = uni(), u2 = uni();
        if (u1 < 1e-300) u1 = 1e-300;
        return std::sqrt(-2.0 * std::log(u1)) * std::cos(6.283185307179586476925 * u2);
    }
    unsigned long long below(unsigned long long n) { return g() % n; }
};

void rot_to_rodrigues(const double R[9], double om[3])
{
    // quaternion route, stable for angles up to pi
    double q[4];
    double tr = R[0] + R[4] + R[8];
    if (tr > 0) {
        double s = std::sqrt(tr + 1.0) * 2;
        q[3] = 0.25 * s; q[0] = (R[7] - R[5]) / s; q[1] = (R[2] - R[6]) / s; q[2] = (R[3] - R[1]) / s;
    } else if (R[0] > R[4] && R[0] > R[8]) {
        double s = std::sqrt(1.0 + R[0] - R[4] - R[8]) * 2;
        q[3] = (R[7] - R[5]) / s; q[0] = 0.25 * s; q[1] = (R[1] + R[3]) / s; q[2] = (R[2] + R[6]) / s;
    } else if (R[4] > R[8]) {
        double s = std::sqrt(1.0 + R[4] - R[0] - R[8]) * 2;
        q[3] = (R[2] - R[6]) / s; q[0] = (R[1] + R[3]) / s; q[1] = 0.25 * s; q[2] = (R[5] + R[7]) / s;
    } else {
        double s = std::sqrt(1.0 + R[8] - R[0] - R[4]) * 2;
        q[3] = (R[3] - R[1]) / s; q[0] = (R[2] + R[6]) / s; q[1] = (R[5] + R[7]) / s; q[2] = 0.25 * s;
    }
    if (q[3] < 0) { for (int i = 0; i < 4; i++) q[i] = -q[i]; }
    double sn = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2]);
    double ang = 2.0 * std::atan2(sn, q[3]);
    double k = (sn < 1e-14) ? 2.0 : ang / sn;
    om[0] = q[0] * k; om[1] = q[1] * k; om[2] = q[2] * k;
}

void rodrigues(const double om[3], double R[9])
{
    double th = std::sqrt(om[0] * om[0] + om[1] * om[1] + om[2] * om[2]);
    double J[9] = {0, -om[2], om[1], om[2], 0, -om[0], -om[1], om[0], 0}, J2[9];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            double a = 0;
            for (int k = 0; k < 3; k++) a += J[i * 3 + k] * J[k * 3 + j];
            J2[i * 3 + j] = a;
        }
    for (int i = 0; i < 9; i++) R[i] = (i % 4 == 0) ? 1.0 : 0.0;
    if (th > 1e-6) {
        double c1 = std::sin(th) / th, c2 = (1 - std::cos(th)) / (th * th);
        for (int i = 0; i < 9; i++) R[i] += c1 * J[i] + c2 * J2[i];
    }
}
} // namespace

int ba_problem_synthetic(int N, int M, int K, unsigned long long seed, ba_problem **out)
{
    if (!out || N < 2 || M < 1 || (long long)K < 2LL * M || (long long)K > (long long)M * N) return BA_ERR_ARG;
    ba_problem *p = new (std::nothrow) ba_problem;
    if (!p) return BA_ERR_NOMEM;
    p->N = N; p->M = M; p->K = K;
    Rng rng(seed);
    const double PI2 = 6.283185307179586476925;
    // true cameras
    std::vector<double> cams_true(9 * (size_t)N), Rtrue(9 * (size_t)N);
    for (int i = 0; i < N; i++) {
        double th = PI2 * i / N;
        double C[3] = {10.0 * std::cos(th), 0.5 * std::sin(3 * th), 10.0 * std::sin(th)};
        double nz = std::sqrt(C[0] * C[0] + C[1] * C[1] + C[2] * C[2]);
        double z[3] = {C[0] / nz, C[1] / nz, C[2] / nz}; // camera looks down -z: z axis points away from the scene
        double up[3] = {0, 1, 0};
        double x[3] = {up[1] * z[2] - up[2] * z[1], up[2] * z[0] - up[0] * z[2], up[0] * z[1] - up[1] * z[0]};
        double nx = std::sqrt(x[0] * x[0] + x[1] * x[1] + x[2] * x[2]);
        for (int k = 0; k < 3; k++) x[k] /= nx;
        double y[3] = {z[1] * x[2] - z[2] * x[1], z[2] * x[0] - z[0] * x[2], z[0] * x[1] - z[1] * x[0]};
        double R[9] = {x[0], x[1], x[2], y[0], y[1], y[2], z[0], z[1], z[2]};
        double om[3];
        rot_to_rodrigues(R, om);
        rodrigues(om, &Rtrue[9 * (size_t)i]); // what a loader will reconstruct
        const double *Rr = &Rtrue[9 * (size_t)i];
        double *c = &cams_true[9 * (size_t)i];
        c[0] = om[0]; c[1] = om[1]; c[2] = om[2];
        for (int r = 0; r < 3; r++) c[3 + r] = -(Rr[3 * r] * C[0] + Rr[3 * r + 1] * C[1] + Rr[3 * r + 2] * C[2]);
        c[6] = rng.uni(1300.0, 6000.0);
        c[7] = rng.uni(-4e-8, 1e-8);
        c[8] = rng.uni(-5e-15, 3e-14);
    }
    std::vector<double> pts_true(3 * (size_t)M);
    for (size_t i = 0; i < pts_true.size(); i++) pts_true[i] = rng.uni(-2.0, 2.0);
    // observations per point: 2 + geometric, adjusted so that the sum is exactly K
    std::vector<int> kj(M);
    const double mean_extra = (double)K / M - 2.0;
    const double pgeo = 1.0 / (1.0 + mean_extra); // P(stop)
    long long tot = 0;
    for (int j = 0; j < M; j++) {
        int g = 0;
        if (mean_extra > 0) {
            double u = rng.uni();
            if (u < 1e-300) u = 1e-300;
            g = (pgeo >= 1.0) ? 0 : (int)std::floor(std::log(u) / std::log(1.0 - pgeo));
        }
        kj[j] = std::min(N, 2 + g);
        tot += kj[j];
    }
    while (tot != K) {
        int j = (int)rng.below((unsigned long long)M);
        if (tot < K && kj[j] < N) { kj[j]++; tot++; }
        else if (tot > K && kj[j] > 2) { kj[j]--; tot--; }
    }
    p->cam_idx.resize(K); p->pt_idx.resize(K); p->meas.resize(2 * (size_t)K);
    std::vector<int> win, sel;
    size_t o = 0;
    for (int j = 0; j < M; j++) {
        const int k = kj[j];
        int W = std::max(N / 4, k / 2 + 1);
        if (2 * W + 1 > N) W = (N - 1) / 2;
        sel.clear();
        if (2 * W + 1 < k) { // only possible when k is close to N: take every camera then trim
            win.resize(N);
            std::iota(win.begin(), win.end(), 0);
        } else {
            const int c0 = (int)rng.below((unsigned long long)N);
            win.resize(2 * W + 1);
            for (int t = 0; t < 2 * W + 1; t++) win[t] = ((c0 - W + t) % N + N) % N;
        }
        for (int t = 0; t < k; t++) { // partial Fisher-Yates
            size_t r = t + (size_t)rng.below((unsigned long long)(win.size() - t));
            std::swap(win[t], win[r]);
            sel.push_back(win[t]);
        }
        std::sort(sel.begin(), sel.end());
        const double *X = &pts_true[3 * (size_t)j];
        for (int t = 0; t < k; t++, o++) {
            const int ci = sel[t];
            const double *R = &Rtrue[9 * (size_t)ci];
            const double *c = &cams_true[9 * (size_t)ci];
            double P[3];
            for (int r = 0; r < 3; r++) P[r] = R[3 * r] * X[0] + R[3 * r + 1] * X[1] + R[3 * r + 2] * X[2] + c[3 + r];
            const double xu = P[0] / P[2], yu = P[1] / P[2];
            const double f = c[6], k1 = c[7] * f * f, k2 = c[8] * f * f * f * f;
            const double r2 = xu * xu + yu * yu;
            const double kr = 1 + k1 * r2 + k2 * r2 * r2;
            double u = -f * kr * xu, v = -f * kr * yu;
            u += 0.3 * rng.normal();
            v += 0.3 * rng.normal();
            if (rng.uni() < 0.1) { u += rng.uni(-5.0, 5.0); v += rng.uni(-5.0, 5.0); }
            p->cam_idx[o] = ci; p->pt_idx[o] = j;
            p->meas[2 * o] = u; p->meas[2 * o + 1] = v;
        }
    }
    // perturbed initial parameters
    p->cams9 = cams_true;
    for (int i = 0; i < N; i++)
        for (int c = 0; c < 6; c++) p->cams9[9 * (size_t)i + c] *= 1.0 + 1e-4 * rng.normal();
    p->pts = pts_true;
    for (size_t i = 0; i < p->pts.size(); i++) p->pts[i] += 2e-3 * rng.normal();
    *out = p;
    return BA_OK;
}

void ba_lm_params_default(ba_lm_params *p)
{
    if (!p) return;
    // Lambda() and LMParams(), src/Eigen_ext/BacktrackLevMarqQRChol.h:131-146
    p->lambda_min = 1e-10; p->lambda_max = 1e10; p->lambda_decrease = 10; p->lambda_increase_base = 2;
    p->lambda_init = 1e-3; p->tol_fun = 1e-8; p->max_iter = 1000000; p->max_fun_ev = 1000000;
    p->max_trials = 0; p->verbose = 0;
}

} // extern "C"
