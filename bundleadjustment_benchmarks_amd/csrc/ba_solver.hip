// ba_solver.hip -- device-resident LM solver behind the C ABI of include/ba_mi355x.h.
//
// Host control flow = the LM classes of the reference:
//   src/Eigen_ext/BacktrackLevMarqQRChol.h:204-436   (QRCHOL; QRKIT reuses the loop)
//   src/Eigen_ext/BacktrackLevMarqCholesky.h:190-361 (CHOLESKY)
// Everything the loops call on the functor / linear solver runs as HIP kernels on resident data; per trial only
// a handful of scalars (test energy, rho denominator, |dx|^2) cross PCIe.
// There is NO CPU fallback: without a HIP device ba_solver_create fails with BA_ERR_HIP.
#include "ba_internal.h"
#include "ba_kernels.hip.h"
#include "ba_dense.hip.h"

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <numeric>
#include <queue>
#include <string>

#define HIPCHK(x)                                                                                   \
    do {                                                                                            \
        hipError_t e_ = (x);                                                                        \
        if (e_ != hipSuccess) {                                                                     \
            fprintf(stderr, "ba_mi355x: %s failed: %s (%s:%d)\n", #x, hipGetErrorString(e_), __FILE__, __LINE__); \
            return BA_ERR_HIP;                                                                      \
        }                                                                                           \
    } while (0)

namespace {

constexpr int NB = BA_NB;    // block-column width of the dense LDL^T
constexpr int NAUG = 3;      // augmented rows: D = reduced rhs, D+1 = g_c, D+2 = spare
constexpr int NSCAL = 16;    // device scalar slots
enum { SC_ENERGY = 0, SC_DMAX_P = 1, SC_ETEST = 2, SC_RHO_P = 3, SC_DN_P = 4, SC_RHO_C = 5, SC_DN_C = 6, SC_DMAX_C = 7,
       SC_ST0 = 8 /* ..11 stats */, SC_LAMBDA = 12 /* lambda of the current trial, read by the kernels */,
       SC_ZERO = 13 /* always 0: the 'lambda' of MOREQR's outer factorisation */,
       SC_ERR = 14 /* device error word: a BA_DEVERR_* code written by a kernel whose in-launch hand-off wait ran out */ };
enum { EV_T0 = 0, EV_T1, EV_T2, EV_T3, EV_T4, EV_T5, EV_T6, EV_L0, EV_L1, EV_L0B, EV_L1B /* linearisation, one pair per buffer */, EV_F, EV_N };

template <typename T> struct DevBuf {
    T *p = nullptr;
    size_t n = 0;
    int alloc(size_t count)
    {
        n = count;
        if (count == 0) return BA_OK;
        hipError_t e = hipMalloc((void **)&p, sizeof(T) * count);
        if (e != hipSuccess) { p = nullptr; return e == hipErrorOutOfMemory ? BA_ERR_NOMEM : BA_ERR_HIP; }
        return BA_OK;
    }
    int upload(const std::vector<T> &h)
    {
        int rc = alloc(h.size());
        if (rc) return rc;
        if (!h.empty() && hipMemcpy(p, h.data(), sizeof(T) * h.size(), hipMemcpyHostToDevice) != hipSuccess) return BA_ERR_HIP;
        return BA_OK;
    }
    ~DevBuf() { if (p) (void)hipFree(p); }
};

struct SolverBase {
    virtual ~SolverBase() {}
    virtual int init(const ba_problem *p, ba_solver_kind kind, int rank, int world) = 0;
    virtual int linearize(double *energy, double *diag_max) = 0;
    virtual int try_step(double lambda, double *e_test, double *rho_scale, double *dx_norm) = 0;
    virtual int accept() = 0;
    virtual int stats(double *out4) = 0;
    virtual int get(int what, double *out, size_t n) = 0;
    virtual int set_state(const double *cam15, const double *pts) = 0;
    virtual void set_stream_hook() {}
    virtual int minimize(const ba_lm_params *lm, ba_trial_cb cb, void *user, ba_result *out) = 0;
    virtual int time_phase(int phase, int reps, double lambda, double *ms) = 0;
    virtual int selftest(int which) = 0;
    ba_allreduce_fn ar_fn = nullptr;
    void *ar_user = nullptr;
    hipStream_t st = nullptr;
    bool own_stream = false;
    bool keep = false; // keep a copy of S / rhs before the factorisation (parity tests)
    ba_structure sx;
    ba_timing tm{};
    int rank = 0, world = 1;
};

template <typename T> void host_rodrigues(const T *om, T *R)
{
    // Math::createRotationMatrixRodrigues, src/MathUtils.h:66-82
    const T th = std::sqrt(om[0] * om[0] + om[1] * om[1] + om[2] * om[2]);
    for (int i = 0; i < 9; i++) R[i] = (i % 4 == 0) ? (T)1 : (T)0;
    if (std::fabs(th) > (T)1e-6) {
        const T J[9] = {0, -om[2], om[1], om[2], 0, -om[0], -om[1], om[0], 0};
        T J2[9];
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) {
                T a = 0;
                for (int k = 0; k < 3; k++) a += J[i * 3 + k] * J[k * 3 + j];
                J2[i * 3 + j] = a;
            }
        const T c1 = std::sin(th) / th, c2 = ((T)1.0 - std::cos(th)) / (th * th);
        for (int i = 0; i < 9; i++) R[i] = R[i] + c1 * J[i] + c2 * J2[i];
    }
}

template <typename T> struct Solver final : SolverBase {
    ba_solver_kind kind = BA_CHOLESKY;
    int N = 0, D = 0, Dp = 0, ld = 0, Ml = 0, Kl = 0;
    T tau = (T)0.5; // INLIER_THRESHOLD, src/bundle_adjustment_large.cpp:36
    // structure
    DevBuf<int> d_obs_cam, d_obs_pt, d_pt_ptr, d_pair_hi, d_pair_lo, d_pair_chunk_ptr,
        d_dchunk_ptr, d_cam_dchunk_ptr, d_cam_obs, d_qr_pts, d_flags;
    DevBuf<int4> d_chunk_info; // per chunk of the pair kernel: first entry, count | BA_CHUNK_SINGLE, camera hi, camera lo
    DevBuf<int2> d_ent;        // per entry: row observation, column observation (~point for a self entry)
    DevBuf<int> d_wave_ptr;    // per wavefront of the pair kernel: its range of chunk descriptors
    DevBuf<int> d_red_pairs;   // pairs k_schur_reduce writes: those without entries and those with several chunks
    int nred = 0;
    int schur_grid = 1, schur_wgs = 4 /* workgroups of k_schur_pairs per CU */, schur_bands = 8, schur_nband = 1;
    // state and work arrays
    // linearisation (r, J, J^T r, block diagonals, MOREQR's outer factors): one set per parameter buffer, so that the
    // linearisation at xTest can be enqueued while the trial that produced xTest is still being judged on the host
    DevBuf<T> d_r[2], d_Jc[2], d_Jp[2], d_JcA[2], d_U0[2], d_gp[2], d_V[2], d_gc[2], d_rec0[2], d_dinv0[2], d_tvec0[2], d_tri0[2];
    DevBuf<T> d_cam[2], d_pts[2], d_meas, d_gcg, d_dslab, d_rec, d_dinv, d_tvec, d_tri,
        d_slab, d_S, d_pack, d_Skeep, d_Wp, d_Winv, d_dxc, d_dxp, d_part_e, d_part_pm, d_part_bs, d_part_st, d_scal;
    int cur = 0; // index of x in d_cam / d_pts; 1 - cur is xTest
    T h_scal[NSCAL];
    T *h_lam = nullptr;                      // pinned staging word for lambda
    T *h_pin = nullptr;                      // pinned landing area of the device scalars (speculating trials)
    hipGraphExec_t gexec[2] = {nullptr, nullptr}; // captured trial, one per parity of the parameter double buffer
    bool use_graph = true;
    hipEvent_t ev[EV_N] = {};
    int gK = 0, gM = 0, gB = 0; // grids (observations, points, points x 8 lanes)
    bool have_step = false;
    int num_cus = 256; // of the device the solver lives on

    ~Solver() override
    {
        for (auto &e : ev)
            if (e) (void)hipEventDestroy(e);
        for (auto &g : gexec)
            if (g) (void)hipGraphExecDestroy(g);
        if (h_lam) (void)hipHostFree(h_lam);
        if (h_pin) (void)hipHostFree(h_pin);
        if (own_stream && st) (void)hipStreamDestroy(st);
    }

    int init(const ba_problem *p, ba_solver_kind k, int rk, int wd) override
    {
        kind = k; rank = rk; world = wd;
        int rc = ba_build_structure(p, rk, wd, BA_CHUNK, 32 /* lanes of a k_cam_gram group */, &sx);
        if (rc) return rc;
        N = p->N; D = 9 * N; Ml = sx.Ml; Kl = sx.Kl;
        Dp = ((D + NAUG + NB - 1) / NB) * NB;
        ld = Dp + 64;
        gK = (Kl + 255) / 256; gM = (Ml + 255) / 256;
        if (gK < 1) gK = 1;
        if (gM < 1) gM = 1;
        gB = (int)(((size_t)Ml * 8 + 255) / 256);
        if (gB < 1) gB = 1;
        if (kind != BA_CHOLESKY && sx.kmax > 1024) return BA_ERR_ARG; // more than 1024 observations of one point: not supported by k_elim_qr
        if (!st) { HIPCHK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking)); own_stream = true; }
        for (auto &e : ev) HIPCHK(hipEventCreate(&e));
        HIPCHK(hipHostMalloc((void **)&h_lam, sizeof(T)));
        HIPCHK(hipHostMalloc((void **)&h_pin, sizeof(T) * NSCAL));
        use_graph = getenv("BA_NO_GRAPH") == nullptr;
        {
            int dev = 0;
            hipDeviceProp_t pr;
            if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess && pr.multiProcessorCount > 0)
                num_cus = pr.multiProcessorCount;
        }
#define UP(buf, vec) if ((rc = buf.upload(vec))) return rc
        UP(d_obs_cam, sx.obs_cam); UP(d_obs_pt, sx.obs_pt); UP(d_pt_ptr, sx.pt_ptr); UP(d_pair_hi, sx.pair_hi);
        UP(d_pair_lo, sx.pair_lo);
        UP(d_pair_chunk_ptr, sx.pair_chunk_ptr); UP(d_dchunk_ptr, sx.dchunk_ptr); UP(d_cam_dchunk_ptr, sx.cam_dchunk_ptr);
        UP(d_cam_obs, sx.cam_obs); UP(d_qr_pts, sx.qr_pts);
#undef UP
        {
            if (const char *ev = getenv("BA_SCHUR_WGS")) schur_wgs = std::max(1, std::min(8, atoi(ev)));
            if (const char *ev = getenv("BA_SCHUR_BANDS")) schur_bands = atoi(ev);
            // Chunks dealt to the wavefronts of the persistent pair kernel, longest first, always to the least loaded wavefront
            // (cost = batches of eight entries + a constant per chunk); a wavefront's list is then walked in chunk order.
            if (N > 65535) return BA_ERR_ARG; // (hi | lo << 16 in the chunk descriptor; a reduced matrix of that size would not fit anyway)
            if ((unsigned long long)Kl * BA_REC * sizeof(T) >= 0xf0000000ull) return BA_ERR_ARG; // 32-bit record offsets: <= 15 M observations per shard (fp64)
            {
                // every wavefront of the persistent grid must be resident at once (the dealing assumes they run side by side)
                int nb = 0;
                const hipError_t oe = kind == BA_CHOLESKY ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_schur_pairs<T, true>, 256, 0)
                                                          : hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_schur_pairs<T, false>, 256, 0);
                if (oe == hipSuccess && nb >= 1) schur_wgs = std::min(schur_wgs, nb);
            }
            schur_grid = std::max(1, std::min((sx.nchunks + 3) / 4, schur_wgs * num_cus));
            const int nband = (schur_bands > 1 && schur_grid >= 8 * schur_bands) ? schur_bands : 1;
            schur_grid = schur_grid / nband * nband;
            const int W = 4 * schur_grid, Wb = W / nband;
            std::vector<int> order((size_t)sx.nchunks), owner((size_t)sx.nchunks), wptr((size_t)W + 1, 0);
            auto cost = [&](int c) { return 2 * ((sx.chunk_ptr[c + 1] - sx.chunk_ptr[c] + 7) / 8) + 1; };
            // bands: the chunk list (sorted by row camera, column camera) cut into nband ranges of equal cost; the workgroups with
            // blockIdx % nband == b (one XCD under the observed round-robin placement) own range b.  Wavefront index: see k_schur_pairs.
            std::vector<int> bptr((size_t)nband + 1, sx.nchunks);
            {
                long long tot = 0, acc = 0;
                for (int c = 0; c < sx.nchunks; c++) tot += cost(c);
                bptr[0] = 0;
                for (int c = 0, b = 1; c < sx.nchunks && b < nband; c++) {
                    acc += cost(c);
                    while (b < nband && acc >= tot * b / nband) bptr[b++] = c + 1;
                }
            }
            for (int b = 0; b < nband; b++) {
                order.assign(bptr[b + 1] - bptr[b], 0);
                std::iota(order.begin(), order.end(), bptr[b]);
                std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return cost(x) > cost(y); });
                typedef std::pair<long long, int> load_t; // (load, wavefront): min-heap
                std::priority_queue<load_t, std::vector<load_t>, std::greater<load_t>> heap;
                for (int w = 0; w < Wb; w++) heap.push(load_t(0, b * Wb + w));
                for (int c : order) {
                    load_t t = heap.top();
                    heap.pop();
                    owner[c] = t.second;
                    wptr[t.second + 1]++;
                    heap.push(load_t(t.first + cost(c), t.second));
                }
            }
            schur_nband = nband;
            for (int w = 0; w < W; w++) wptr[w + 1] += wptr[w];
            std::vector<int4> ci((size_t)sx.nchunks);
            {
                std::vector<int> cur(wptr.begin(), wptr.end() - 1);
                for (int c = 0; c < sx.nchunks; c++) { // increasing chunk id inside every wavefront's list
                    const int q = sx.chunk_pair[c];
                    const bool single = sx.pair_chunk_ptr[q + 1] - sx.pair_chunk_ptr[q] == 1;
                    ci[cur[owner[c]]++] = make_int4(sx.chunk_ptr[c], (sx.chunk_ptr[c + 1] - sx.chunk_ptr[c]) | (single ? BA_CHUNK_SINGLE : 0),
                                                    sx.pair_hi[q] | (sx.pair_lo[q] << 16), c);
                }
            }
            if ((rc = d_wave_ptr.upload(wptr))) return rc;
            std::vector<int> red;
            for (int q = 0; q < sx.npairs; q++)
                if (sx.pair_chunk_ptr[q + 1] - sx.pair_chunk_ptr[q] != 1) red.push_back(q);
            nred = (int)red.size();
            if ((rc = d_red_pairs.upload(red))) return rc;
            std::vector<int2> en((size_t)sx.E);
            for (long long e = 0; e < sx.E; e++) { // a self entry carries ~point in place of its column observation (k_schur_pairs)
                const int r_ = sx.ent_r[(size_t)e], c_ = sx.ent_c[(size_t)e];
                en[(size_t)e] = make_int2(r_, r_ == c_ ? ~sx.obs_pt[r_] : c_);
            }
            if ((rc = d_chunk_info.upload(ci)) || (rc = d_ent.upload(en))) return rc;
        }
        // parameters: bundle_adjustment_large.cpp:81-107 (K00 = -f, R = Rodrigues(omega), distortion (k1 f^2, k2 f^4))
        std::vector<T> cam((size_t)15 * N), pts((size_t)3 * (Ml > 0 ? Ml : 1)), meas((size_t)2 * (Kl > 0 ? Kl : 1));
        for (int i = 0; i < N; i++) {
            const double *c = &p->cams9[9 * (size_t)i];
            T om[3] = {(T)c[0], (T)c[1], (T)c[2]}, R[9];
            host_rodrigues<T>(om, R);
            for (int q = 0; q < 9; q++) cam[(size_t)q * N + i] = R[q];
            for (int q = 0; q < 3; q++) cam[(size_t)(9 + q) * N + i] = (T)c[3 + q];
            const T f = (T)c[6], k1 = (T)c[7], k2 = (T)c[8], f2 = f * f;
            cam[(size_t)12 * N + i] = -f / (T)1.0;
            cam[(size_t)13 * N + i] = k1 * f2;
            cam[(size_t)14 * N + i] = k2 * f2 * f2;
        }
        for (int j = 0; j < Ml; j++)
            for (int q = 0; q < 3; q++) pts[(size_t)q * Ml + j] = (T)p->pts[3 * (size_t)(sx.p0 + j) + q];
        for (int i = 0; i < Kl; i++) {
            const int src = sx.perm[sx.o0 + i];
            meas[i] = (T)p->meas[2 * (size_t)src];              // / avg_focal_length (= 1.0, :35,72)
            meas[(size_t)Kl + i] = (T)p->meas[2 * (size_t)src + 1];
        }
        if ((rc = d_cam[0].upload(cam)) || (rc = d_cam[1].upload(cam)) || (rc = d_pts[0].upload(pts)) ||
            (rc = d_pts[1].upload(pts)) || (rc = d_meas.upload(meas)))
            return rc;
        const size_t K1 = Kl > 0 ? Kl : 1, M1 = Ml > 0 ? Ml : 1;
#define AL(buf, n) if ((rc = buf.alloc(n))) return rc
        for (int w = 0; w < 2; w++) {
            AL(d_r[w], 2 * K1); AL(d_Jc[w], 18 * K1); AL(d_JcA[w], 20 * K1); AL(d_Jp[w], 6 * K1); AL(d_U0[w], 6 * M1); AL(d_gp[w], 3 * M1);
            AL(d_V[w], (size_t)81 * N); AL(d_gc[w], (size_t)D);
            if (kind == BA_MOREQR) { AL(d_rec0[w], (size_t)BA_REC * K1); AL(d_dinv0[w], 3 * M1); AL(d_tvec0[w], 3 * M1); AL(d_tri0[w], 6 * M1); }
        }
        AL(d_gcg, (size_t)D); AL(d_dslab, (size_t)BA_SLAB * (sx.ndchunks > 0 ? sx.ndchunks : 1));
        AL(d_rec, (size_t)BA_REC * K1); AL(d_dinv, 3 * M1); AL(d_tvec, 3 * M1); AL(d_tri, 6 * M1);
        AL(d_slab, (size_t)BA_SLAB * (sx.nchunks > 0 ? sx.nchunks : 1));
        AL(d_S, (size_t)ld * (Dp + 64)); AL(d_Wp, (size_t)2 * ld * NB); AL(d_Winv, (size_t)((D + NB - 1) / NB) * NB * NB); AL(d_dxc, (size_t)Dp); AL(d_dxp, 3 * M1);
        if ((rc = d_flags.alloc((size_t)Dp / NB + 2))) return rc;
        AL(d_part_e, (size_t)gK); AL(d_part_pm, (size_t)gM);
        AL(d_part_bs, (size_t)2 * gB); AL(d_part_st, (size_t)4 * gK); AL(d_scal, NSCAL);
#undef AL
        HIPCHK(hipMemset(d_S.p, 0, sizeof(T) * d_S.n));
        HIPCHK(hipMemset(d_Wp.p, 0, sizeof(T) * d_Wp.n));
        HIPCHK(hipMemset(d_scal.p, 0, sizeof(T) * NSCAL));
        HIPCHK(hipMemset(d_dxc.p, 0, sizeof(T) * d_dxc.n));
        HIPCHK(hipMemset(d_part_bs.p, 0, sizeof(T) * d_part_bs.n));
        HIPCHK(hipMemset(d_part_e.p, 0, sizeof(T) * d_part_e.n));
        HIPCHK(hipMemset(d_part_pm.p, 0, sizeof(T) * d_part_pm.n));
        HIPCHK(hipDeviceSynchronize());
        return BA_OK;
    }

    int allreduce(void *buf, size_t count, int op)
    {
        if (world <= 1) return BA_OK;
        if (!ar_fn) return BA_ERR_COMM;
        const auto t0 = std::chrono::steady_clock::now();
        int rc = ar_fn(ar_user, buf, count, sizeof(T) == 8 ? BA_F64 : BA_F32, op, (void *)st);
        tm.comm_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        return rc ? BA_ERR_COMM : BA_OK;
    }

    int fetch_scalars()
    {
        HIPCHK(hipMemcpyAsync(h_scal, d_scal.p, sizeof(T) * NSCAL, hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        return check_device_error();
    }

    // The device error word (SC_ERR) of the scalars just read: a kernel's bounded wait for another workgroup of its launch ran
    // out (k_ldlt_step's row-update flag, k_ldlt_backflow's sentinel poll), so the step is garbage.  Loud, and cleared for the next call.
    int check_device_error()
    {
        if (h_scal[SC_ERR] == (T)0) return BA_OK;
        fprintf(stderr, "ba_mi355x: device error %d: %s\n", (int)h_scal[SC_ERR],
                (int)h_scal[SC_ERR] == BA_DEVERR_ROW_FLAG ? "k_ldlt_step: the look-ahead update of a row block was never announced"
                                                          : "k_ldlt_backflow: an unknown of the backward sweep was never published");
        h_scal[SC_ERR] = 0;
        (void)hipMemsetAsync(d_scal.p + SC_ERR, 0, sizeof(T), st);
        return BA_ERR_HIP;
    }

    double ev_ms(int a, int b)
    {
        float ms = 0;
        if (hipEventElapsedTime(&ms, ev[a], ev[b]) != hipSuccess) return 0;
        return ms;
    }

    void launch_eval(bool jac, int which)
    {
        const T tau2 = tau * tau;
        if (jac)
            hipLaunchKernelGGL((k_eval<T, true>), dim3(gK), dim3(256), 0, st, Kl, N, Ml, d_cam[which].p, d_pts[which].p,
                               d_obs_cam.p, d_obs_pt.p, d_meas.p, tau2, d_r[which].p, d_Jc[which].p, d_Jp[which].p, d_JcA[which].p, d_part_e.p);
        else
            hipLaunchKernelGGL((k_eval<T, false>), dim3(gK), dim3(256), 0, st, Kl, N, Ml, d_cam[which].p, d_pts[which].p,
                               d_obs_cam.p, d_obs_pt.p, d_meas.p, tau2, (T *)nullptr, (T *)nullptr, (T *)nullptr, (T *)nullptr, d_part_e.p);
    }

    void launch_grad(int which)
    {
        hipLaunchKernelGGL((k_point_prep<T>), dim3(gM), dim3(256), 0, st, Ml, Kl, d_pt_ptr.p, d_Jp[which].p, d_r[which].p, d_U0[which].p, d_gp[which].p,
                           d_part_pm.p);
        if (sx.ndchunks > 0)
            hipLaunchKernelGGL((k_cam_gram<T>), dim3((sx.ndchunks + 7) / 8), dim3(256), 0, st, sx.ndchunks, Kl,
                               d_dchunk_ptr.p, d_cam_obs.p, d_JcA[which].p, d_dslab.p);
        hipLaunchKernelGGL((k_cam_gram_reduce<T>), dim3((N * BA_SLAB + 191) / 192), dim3(192), 0, st, N, d_cam_dchunk_ptr.p,
                           d_dslab.p, d_V[which].p, d_gc[which].p);
    }

    // m_functor(x, r); energy; m_functor.df(x, J); JtRes; column norms (BacktrackLevMarqQRChol.h:257-280)
    int linearize(double *energy, double *diag_max) override
    {
        int rc;
        if ((rc = linearize_enqueue(diag_max != nullptr))) return rc;
        if ((rc = allreduce(d_scal.p + SC_ENERGY, 1, 0))) return rc;
        if (diag_max && (rc = allreduce(d_scal.p + SC_DMAX_P, 1, 1))) return rc;
        if ((rc = fetch_scalars())) return rc;
        HIPCHK(hipGetLastError());
        linearize_account();
        if (energy) *energy = (double)h_scal[SC_ENERGY];
        if (diag_max) *diag_max = std::max((double)h_scal[SC_DMAX_P], (double)h_scal[SC_DMAX_C]);
        return BA_OK;
    }

    // The launches of linearize() without the read-back: the energy lands in the device scalar slot SC_ENERGY, which the
    // trial kernels do not touch, so a single-shard LM loop enqueues the trial right behind and reads both results with the
    // one synchronisation of the trial (no host round trip between an accepted step and the next trial).
    // which: the parameter buffer to linearise at -- cur, or 1 - cur = xTest of the trial just enqueued (speculation on its
    // acceptance: the set of linearisation arrays of the other buffer is written, the current one stays intact for a retry).
    int linearize_enqueue(bool want_dmax, int which = -1)
    {
        int rc;
        const bool speculative = which >= 0 && which != cur;
        if (which < 0) which = cur;
        HIPCHK(hipEventRecord(ev[EV_L0 + 2 * which], st));
        launch_eval(true, which);
        launch_grad(which);
        if (kind == BA_MOREQR) // m_solver.compute(J) + Q^T r, once per outer iteration (BacktrackLevMarqMore.h:288-291)
            launch_elim_qr(which, d_scal.p + SC_ZERO, d_rec0[which].p, d_dinv0[which].p, d_tvec0[which].p, d_tri0[which].p);
        ba_red_jobs jobs{};
        int nj = 0;
        jobs.j[nj++] = {d_part_e.p, gK, 0, SC_ENERGY};
        if (want_dmax) {
            // max diag(J^T J): point part per shard, camera part from the (summed over shards) diagonal of J_c^T J_c
            T *tmp = d_dxc.p;
            hipLaunchKernelGGL((k_vdiag<T>), dim3((D + 255) / 256), dim3(256), 0, st, N, d_V[which].p, tmp);
            if ((rc = allreduce(tmp, (size_t)D, 0))) return rc;
            jobs.j[nj++] = {d_part_pm.p, gM, 1, SC_DMAX_P};
            jobs.j[nj++] = {tmp, D, 1, SC_DMAX_C};
        }
        hipLaunchKernelGGL((k_reduce_scalars<T>), dim3(nj), dim3(256), 0, st, jobs, d_scal.p);
        HIPCHK(hipEventRecord(ev[EV_L1 + 2 * which], st));
        if (!speculative) have_step = false;
        return BA_OK;
    }

    void linearize_account() // of the linearisation at the current buffer (its events are complete: its results have been used)
    {
        tm.linearize_ms += ev_ms(EV_L0 + 2 * cur, EV_L1 + 2 * cur);
        tm.n_linearize++;
    }

    void launch_eliminate()
    {
        if (kind == BA_CHOLESKY) {
            hipLaunchKernelGGL((k_elim_chol<T>), dim3(gK), dim3(256), 0, st, Kl, Ml, d_obs_pt.p, d_pt_ptr.p, d_Jc[cur].p, d_Jp[cur].p,
                               d_U0[cur].p, d_gp[cur].p, d_scal.p + SC_LAMBDA, d_rec.p, d_dinv.p, d_tvec.p, d_tri.p);
        } else if (kind == BA_MOREQR) {
            if (Kl > 0) // BacktrackLevMarqMore.h:297-345, the per-trial QR of [R ; sqrt(lambda) I]
                hipLaunchKernelGGL((k_more_trial<T>), dim3(gK), dim3(256), 0, st, Kl, Ml, d_obs_pt.p, d_pt_ptr.p, d_scal.p + SC_LAMBDA,
                                   d_rec0[cur].p, d_tri0[cur].p, d_tvec0[cur].p, d_rec.p, d_dinv.p, d_tvec.p, d_tri.p);
        } else {
            launch_elim_qr(cur, d_scal.p + SC_LAMBDA, d_rec.p, d_dinv.p, d_tvec.p, d_tri.p);
        }
    }

    // per-point QR, one launch per non-empty track-length bucket (ba_structure: lanes per point x observations per lane)
    void launch_elim_qr(int which, const T *lam, T *rec, T *dinv, T *tvec, T *tri)
    {
#define BA_QR(B, L, SLOTS)                                                                                                       \
        if (sx.qr_bucket_ptr[B + 1] > sx.qr_bucket_ptr[B]) {                                                                   \
            const int np_ = sx.qr_bucket_ptr[B + 1] - sx.qr_bucket_ptr[B];                                                       \
            hipLaunchKernelGGL((k_elim_qr<T, L, SLOTS>), dim3(((size_t)np_ * L + 255) / 256), dim3(256), 0, st, np_,           \
                               d_qr_pts.p + sx.qr_bucket_ptr[B], Ml, Kl, d_pt_ptr.p, d_Jc[which].p, d_Jp[which].p, d_r[which].p, lam, rec, dinv, tvec, tri); \
        }
        BA_QR(0, 8, 4)
        BA_QR(1, 16, 4)
        BA_QR(2, 32, 4)
        BA_QR(3, 64, 4)
        BA_QR(4, 64, 16)
#undef BA_QR
    }

    void launch_schur()
    {
        if (sx.nchunks > 0) {
            // persistent: schur_wgs workgroups per CU, every wavefront walks its own balanced list of chunks (see k_schur_pairs)
            const dim3 gp(schur_grid);
#define BA_PAIRS(SC) hipLaunchKernelGGL((k_schur_pairs<T, SC>), gp, dim3(256), 0, st, d_wave_ptr.p, schur_nband, d_chunk_info.p, d_ent.p, d_rec.p, \
                                        (unsigned)(sizeof(T) * d_rec.n), d_tvec.p, Ml, d_slab.p, d_V[cur].p, d_gc[cur].p, D, ld, d_S.p)
            // SCALED: CHOLESKY is the only symbol whose point blocks carry a diagonal D (dinv != 1)
            if (kind == BA_CHOLESKY) BA_PAIRS(true); else BA_PAIRS(false);
#undef BA_PAIRS
        }
        const long long nthr = (long long)nred * BA_SLAB;
        if (nred > 0)
            hipLaunchKernelGGL((k_schur_reduce<T>), dim3((unsigned)((nthr + 191) / 192)), dim3(192), 0, st, nred, d_red_pairs.p, D, ld,
                               d_pair_hi.p, d_pair_lo.p, d_pair_chunk_ptr.p, d_slab.p, d_V[cur].p, d_gc[cur].p, d_S.p);
    }

    void launch_factor_solve() { launch_factor(); launch_backsweep(); }

    void launch_factor() { ba_ldlt_factor<T, NB>(st, D + 1, D, ld, d_S.p, d_Wp.p, d_Winv.p, d_flags.p, (int)d_flags.n, d_scal.p + SC_ERR); }

    // backward sweep: one data-flow launch (k_ldlt_backflow) while its groups are certainly co-resident
    void launch_backsweep() { ba_ldlt_backsweep<T, NB>(st, D, ld, D, d_S.p, d_Winv.p, d_dxc.p, /*armed by k_post_reduce*/ true, num_cus, d_scal.p + SC_ERR); }

    void launch_post_reduce()
    {
        hipLaunchKernelGGL((k_post_reduce<T>), dim3((Dp + 3) / 4), dim3(256), 0, st, D, Dp, ld, d_scal.p + SC_LAMBDA, d_S.p, d_gcg.p, d_dxc.p);
    }

    int set_lambda(T lambda)
    {
        *h_lam = lambda;
        HIPCHK(hipMemcpyAsync(d_scal.p + SC_LAMBDA, h_lam, sizeof(T), hipMemcpyHostToDevice, st));
        return BA_OK;
    }

    void launch_backsub_retract()
    {
        if (Ml > 0) // (an empty shard keeps the zero partial sums written at creation)
        hipLaunchKernelGGL((k_backsub<T, 8>), dim3(gB), dim3(256), 0, st, Ml, d_pt_ptr.p, d_obs_cam.p, d_rec.p, d_dinv.p, d_tvec.p,
                           d_tri.p, d_dxc.p, d_gp[cur].p, d_pts[cur].p, d_scal.p + SC_LAMBDA, d_dxp.p, d_pts[1 - cur].p, d_part_bs.p);
        hipLaunchKernelGGL((k_retract_cams<T>), dim3(1), dim3(256), 0, st, N, d_cam[cur].p, d_dxc.p, d_gcg.p, d_scal.p + SC_LAMBDA,
                           d_cam[1 - cur].p, d_scal.p, (int)SC_RHO_C);
    }

    // Everything one trial enqueues after lambda has been set (single shard: no host interaction in between).
    void launch_trial_kernels()
    {
        launch_eliminate();
        launch_schur();
        launch_post_reduce();
        launch_factor_solve();
        launch_backsub_retract();
        launch_eval(false, 1 - cur);
        ba_red_jobs jobs{};
        jobs.j[0] = {d_part_e.p, gK, 0, SC_ETEST};
        jobs.j[1] = {d_part_bs.p, gB, 0, SC_RHO_P};
        jobs.j[2] = {d_part_bs.p + gB, gB, 0, SC_DN_P};
        hipLaunchKernelGGL((k_reduce_scalars<T>), dim3(3), dim3(256), 0, st, jobs, d_scal.p);
    }

    // m_solver.compute .. dx; xTest = x (+) dx; m_functor(xTest); rhoScale (BacktrackLevMarqQRChol.h:291-375)
    // graph = true (used by minimize on a single shard): the ~110 launches of a trial are captured once per parameter
    // buffer parity into a hipGraph and replayed; lambda reaches the kernels through device memory.
    // speculate (minimize, single shard, replayed trials): the linearisation at xTest is enqueued right behind the trial, in
    // front of the wait for its scalars -- the GPU forms it while the host judges the step and launches the next trial; a
    // rejected step simply leaves it unused (it lives in the other buffer's set of arrays).
    int try_step_impl(double lambda_d, double *e_test, double *rho_scale, double *dx_norm, bool graph, bool speculate = false)
    {
        int rc;
        if ((rc = set_lambda((T)lambda_d))) return rc;
        if (graph && world == 1 && !keep) {
            if (!gexec[cur]) {
                hipGraph_t g = nullptr;
                HIPCHK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
                launch_trial_kernels();
                HIPCHK(hipStreamEndCapture(st, &g));
                HIPCHK(hipGraphInstantiate(&gexec[cur], g, nullptr, nullptr, 0));
                HIPCHK(hipGraphDestroy(g));
            }
            HIPCHK(hipEventRecord(ev[EV_T0], st));
            HIPCHK(hipGraphLaunch(gexec[cur], st));
            HIPCHK(hipEventRecord(ev[EV_T6], st));
            if (speculate) {
                HIPCHK(hipMemcpyAsync(h_pin, d_scal.p, sizeof(T) * NSCAL, hipMemcpyDeviceToHost, st));
                HIPCHK(hipEventRecord(ev[EV_F], st));
                if ((rc = linearize_enqueue(false, 1 - cur))) return rc;
                HIPCHK(hipEventSynchronize(ev[EV_F]));
                for (int i = 0; i < NSCAL; i++) h_scal[i] = h_pin[i];
                if ((rc = check_device_error())) return rc;
            } else if ((rc = fetch_scalars())) return rc;
            HIPCHK(hipGetLastError());
            tm.trial_ms += ev_ms(EV_T0, EV_T6);
            tm.n_graph_trials++;
        } else {
            HIPCHK(hipEventRecord(ev[EV_T0], st));
            launch_eliminate();
            HIPCHK(hipEventRecord(ev[EV_T1], st));
            launch_schur();
            HIPCHK(hipEventRecord(ev[EV_T2], st));
            if (world > 1) {
                // all-reduce only the block-lower trapezoid (matrix + rhs row + g_c row): half the bytes of the full buffer
                const int nbc = Dp / NB;
                const size_t npk = (size_t)64 * ((size_t)nbc * Dp - (size_t)32 * nbc * (nbc - 1));
                if (!d_pack.p && (rc = d_pack.alloc(npk))) return rc;
                const dim3 gpk((Dp + 255) / 256 > 8 ? 8 : (Dp + 255) / 256, Dp);
                hipLaunchKernelGGL((k_pack_lower<T, false>), gpk, dim3(256), 0, st, Dp, ld, d_S.p, d_pack.p);
                if ((rc = allreduce(d_pack.p, npk, 0))) return rc;
                hipLaunchKernelGGL((k_pack_lower<T, true>), gpk, dim3(256), 0, st, Dp, ld, d_S.p, d_pack.p);
            }
            launch_post_reduce();
            if (keep) {
                if (!d_Skeep.p && (rc = d_Skeep.alloc(d_S.n))) return rc;
                HIPCHK(hipMemcpyAsync(d_Skeep.p, d_S.p, sizeof(T) * d_S.n, hipMemcpyDeviceToDevice, st));
            }
            HIPCHK(hipEventRecord(ev[EV_T3], st));
            launch_factor_solve();
            HIPCHK(hipEventRecord(ev[EV_T4], st));
            launch_backsub_retract();
            HIPCHK(hipEventRecord(ev[EV_T5], st));
            launch_eval(false, 1 - cur);
            ba_red_jobs jobs{};
            jobs.j[0] = {d_part_e.p, gK, 0, SC_ETEST};
            jobs.j[1] = {d_part_bs.p, gB, 0, SC_RHO_P};
            jobs.j[2] = {d_part_bs.p + gB, gB, 0, SC_DN_P};
            hipLaunchKernelGGL((k_reduce_scalars<T>), dim3(3), dim3(256), 0, st, jobs, d_scal.p);
            HIPCHK(hipEventRecord(ev[EV_T6], st));
            if ((rc = allreduce(d_scal.p + SC_ETEST, 3, 0))) return rc;
            if ((rc = fetch_scalars())) return rc;
            HIPCHK(hipGetLastError());
            tm.eliminate_ms += ev_ms(EV_T0, EV_T1);
            tm.schur_ms += ev_ms(EV_T1, EV_T2);
            tm.factor_ms += ev_ms(EV_T3, EV_T4);
            tm.backsub_ms += ev_ms(EV_T4, EV_T5);
            tm.test_eval_ms += ev_ms(EV_T5, EV_T6);
            tm.trial_ms += ev_ms(EV_T0, EV_T6);
        }
        tm.n_trials++;
        if (e_test) *e_test = (double)h_scal[SC_ETEST];
        if (rho_scale) *rho_scale = (double)(h_scal[SC_RHO_P] + h_scal[SC_RHO_C]);
        if (dx_norm) *dx_norm = std::sqrt((double)(h_scal[SC_DN_P] + h_scal[SC_DN_C]));
        have_step = true;
        return BA_OK;
    }

    int try_step(double lambda_d, double *e_test, double *rho_scale, double *dx_norm) override
    {
        return try_step_impl(lambda_d, e_test, rho_scale, dx_norm, false);
    }

    int accept() override
    {
        if (!have_step) return BA_ERR_ARG;
        cur = 1 - cur;
        have_step = false;
        return BA_OK;
    }

    int stats(double *out4) override
    {
        hipLaunchKernelGGL((k_stats<T>), dim3(gK), dim3(256), 0, st, Kl, N, Ml, d_cam[cur].p, d_pts[cur].p, d_obs_cam.p,
                           d_obs_pt.p, d_meas.p, tau, d_part_st.p);
        ba_red_jobs jobs{};
        for (int q = 0; q < 4; q++) jobs.j[q] = {d_part_st.p + (size_t)q * gK, gK, 0, SC_ST0 + q};
        hipLaunchKernelGGL((k_reduce_scalars<T>), dim3(4), dim3(256), 0, st, jobs, d_scal.p);
        int rc;
        if ((rc = allreduce(d_scal.p + SC_ST0, 4, 0))) return rc;
        if ((rc = fetch_scalars())) return rc;
        const double K = (double)sx.K;
        // accumulate and divide in Scalar like the reference (Utils.h:24-40)
        out4[0] = (double)(h_scal[SC_ST0] / (T)K);
        out4[1] = (double)(h_scal[SC_ST0 + 1] / h_scal[SC_ST0 + 2]);
        out4[2] = (double)h_scal[SC_ST0 + 2];
        out4[3] = (double)h_scal[SC_ST0 + 3];
        return BA_OK;
    }

    int dl(const T *src, size_t n, std::vector<T> &h)
    {
        h.resize(n);
        HIPCHK(hipStreamSynchronize(st));
        if (n) HIPCHK(hipMemcpy(h.data(), src, sizeof(T) * n, hipMemcpyDeviceToHost));
        return BA_OK;
    }

    int get(int what, double *out, size_t n) override
    {
        std::vector<T> h, h2;
        int rc;
        switch (what) {
        case BA_GET_RESIDUALS: {
            if (n != 2 * (size_t)Kl) return BA_ERR_ARG;
            if ((rc = dl(d_r[cur].p, 2 * (size_t)Kl, h))) return rc;
            // file order inside the shard when the input was sorted; sorted order otherwise (perm documents it)
            for (int i = 0; i < Kl; i++) { out[2 * (size_t)i] = h[i]; out[2 * (size_t)i + 1] = h[(size_t)Kl + i]; }
            return BA_OK;
        }
        case BA_GET_JC: {
            if (n != 18 * (size_t)Kl) return BA_ERR_ARG;
            if ((rc = dl(d_Jc[cur].p, 18 * (size_t)Kl, h))) return rc;
            for (int i = 0; i < Kl; i++)
                for (int q = 0; q < 18; q++) out[18 * (size_t)i + q] = h[(size_t)q * Kl + i];
            return BA_OK;
        }
        case BA_GET_JP: {
            if (n != 6 * (size_t)Kl) return BA_ERR_ARG;
            if ((rc = dl(d_Jp[cur].p, 6 * (size_t)Kl, h))) return rc;
            for (int i = 0; i < Kl; i++)
                for (int q = 0; q < 6; q++) out[6 * (size_t)i + q] = h[(size_t)q * Kl + i];
            return BA_OK;
        }
        case BA_GET_GRAD: {
            if (n != 3 * (size_t)Ml + D) return BA_ERR_ARG;
            if ((rc = dl(d_gp[cur].p, 3 * (size_t)Ml, h)) || (rc = dl(d_gc[cur].p, D, h2))) return rc;
            for (int j = 0; j < Ml; j++)
                for (int q = 0; q < 3; q++) out[3 * (size_t)j + q] = h[(size_t)q * Ml + j];
            for (int c = 0; c < D; c++) out[3 * (size_t)Ml + c] = h2[c];
            return BA_OK;
        }
        case BA_GET_S:
        case BA_GET_RHS: {
            if (!d_Skeep.p) return BA_ERR_ARG;
            if ((rc = dl(d_Skeep.p, d_Skeep.n, h))) return rc;
            if (what == BA_GET_RHS) {
                if (n != (size_t)D) return BA_ERR_ARG;
                for (int c = 0; c < D; c++) out[c] = h[(size_t)c * ld + D];
            } else {
                if (n != (size_t)D * D) return BA_ERR_ARG;
                for (int c = 0; c < D; c++)
                    for (int rr = c; rr < D; rr++) out[(size_t)c * D + rr] = out[(size_t)rr * D + c] = h[(size_t)c * ld + rr];
            }
            return BA_OK;
        }
        case BA_GET_DX: {
            if (n != 3 * (size_t)Ml + D) return BA_ERR_ARG;
            if ((rc = dl(d_dxp.p, 3 * (size_t)Ml, h)) || (rc = dl(d_dxc.p, D, h2))) return rc;
            for (int j = 0; j < Ml; j++)
                for (int q = 0; q < 3; q++) out[3 * (size_t)j + q] = h[(size_t)q * Ml + j];
            for (int c = 0; c < D; c++) out[3 * (size_t)Ml + c] = h2[c];
            return BA_OK;
        }
        case BA_GET_CAMS:
        case BA_GET_CAMS_TEST: {
            if (n != 15 * (size_t)N) return BA_ERR_ARG;
            if ((rc = dl(d_cam[what == BA_GET_CAMS ? cur : 1 - cur].p, 15 * (size_t)N, h))) return rc;
            for (int a = 0; a < N; a++)
                for (int q = 0; q < 15; q++) out[15 * (size_t)a + q] = h[(size_t)q * N + a];
            return BA_OK;
        }
        case BA_GET_POINTS:
        case BA_GET_POINTS_TEST: {
            if (n != 3 * (size_t)Ml) return BA_ERR_ARG;
            if ((rc = dl(d_pts[what == BA_GET_POINTS ? cur : 1 - cur].p, 3 * (size_t)Ml, h))) return rc;
            for (int j = 0; j < Ml; j++)
                for (int q = 0; q < 3; q++) out[3 * (size_t)j + q] = h[(size_t)q * Ml + j];
            return BA_OK;
        }
        }
        return BA_ERR_ARG;
    }

    int set_state(const double *cam15, const double *pts) override
    {
        HIPCHK(hipStreamSynchronize(st));
        if (cam15) {
            std::vector<T> h((size_t)15 * N);
            for (int a = 0; a < N; a++)
                for (int q = 0; q < 15; q++) h[(size_t)q * N + a] = (T)cam15[15 * (size_t)a + q];
            HIPCHK(hipMemcpy(d_cam[cur].p, h.data(), sizeof(T) * h.size(), hipMemcpyHostToDevice));
        }
        if (pts && Ml > 0) {
            std::vector<T> h((size_t)3 * Ml);
            for (int j = 0; j < Ml; j++)
                for (int q = 0; q < 3; q++) h[(size_t)q * Ml + j] = (T)pts[3 * (size_t)j + q];
            HIPCHK(hipMemcpy(d_pts[cur].p, h.data(), sizeof(T) * h.size(), hipMemcpyHostToDevice));
        }
        have_step = false;
        return BA_OK;
    }

    // The LM loop: BacktrackLevMarqQRChol.h:204-436 == BacktrackLevMarqCholesky.h:190-361 (Scalar arithmetic in T).
    int minimize(const ba_lm_params *lmp, ba_trial_cb cb, void *user, ba_result *out) override
    {
        ba_lm_params lm;
        if (lmp) lm = *lmp; else ba_lm_params_default(&lm);
        const auto tbeg = std::chrono::steady_clock::now();
        ba_timing tm0 = tm;
        if (lm.verbose && rank == 0) {
            // outputHeader / outputIterHeader, BacktrackLevMarqQRChol.h:65-82
            printf("############################## Backtrack LevMarq ###############################\n");
            printf("--------------------------------------------------------------------------------\n");
            printf(" Iter%15s%15s%15s%15s%15s\n", "Status", "f", "rho", "lambda", "Elapsed");
            printf("--------------------------------------------------------------------------------\n");
        }
        T lambda = (T)lm.lambda_init, lambda_inc = (T)lm.lambda_increase_base;
        const T lam_min = (T)lm.lambda_min, lam_max = (T)lm.lambda_max, tol_fun = (T)lm.tol_fun;
        T hist[2] = {0, 0}, energy = 0;
        int fun_evals = 0, iter = 0, trials = 0, status = BA_RUNNING, rc = BA_OK;
        bool stop = false, lin_pending = false;
        while (true) {
            iter++;
            if (iter > lm.max_iter) { status = BA_MAX_ITERS; break; }
            if (fun_evals > lm.max_fun_ev) { status = BA_TOO_MANY_FUN_EVALS; break; }
            double e = 0, dmax = 0;
            // after an accepted step of a single-shard run the linearisation is already in the stream (try_step_impl, speculate);
            // its energy (== the test energy of that step, evaluated by the same code at the same point) is read back together
            // with the next trial
            const bool lin_spec = iter > 1 && world == 1 && use_graph && !keep;
            // sharded run: no speculation (the trial is not replayed), but the linearisation and the all-reduce of its energy
            // are only enqueued as well -- the energy comes back with the scalars of the next trial, one host
            // synchronisation per LM iteration instead of two
            const bool lin_async = lin_spec || (iter > 1 && world > 1 && !keep);
            if (lin_spec) {
                // (already enqueued behind the accepted trial, speculatively, for the buffer that is now the current one)
            } else if (lin_async) {
                if ((rc = linearize_enqueue(false)) || (rc = allreduce(d_scal.p + SC_ENERGY, 1, 0))) break;
            } else {
                if ((rc = linearize(&e, iter == 1 ? &dmax : nullptr))) break;
                energy = (T)e;
            }
            lin_pending = lin_async;
            fun_evals++;
            if (iter == 1) // :278-280 / Cholesky.h:263-265; MOREQR: 1e-6 * max column norm (BacktrackLevMarqMore.h:272-284)
                lambda = kind == BA_MOREQR ? (T)(1e-6 * std::sqrt(dmax)) : (T)(1e-12 * dmax);
            while (true) {
                if (lm.max_trials > 0 && trials >= lm.max_trials) { stop = true; status = BA_RUNNING; break; }
                const auto t0 = std::chrono::steady_clock::now();
                double et = 0, rs = 0, dn = 0;
                if ((rc = try_step_impl((double)lambda, &et, &rs, &dn, use_graph, world == 1 && use_graph && !keep))) { stop = true; break; }
                if (lin_pending) { energy = h_scal[SC_ENERGY]; linearize_account(); lin_pending = false; }
                fun_evals++;
                const T e_test = (T)et;
                const double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
                trials++;
                if (e_test < energy) {
                    const T rho = (energy - e_test) / (T)rs;
                    const T tmv = (T)2.0 * rho - (T)1.0;
                    const T mul = (T)1.0 - tmv * tmv * tmv;
                    lambda *= std::max<T>((T)1.0 / (T)3.0, mul);
                    lambda = std::max<T>(lambda, lam_min);
                    if (cb) cb(user, iter, 1, (double)energy, (double)rho, (double)lambda, el);
                    if (lm.verbose && rank == 0)
                        printf("%5d%15s%15g%15g%15g%14gs\n", iter, "Accepted", (double)energy, (double)rho, (double)lambda, el);
                    lambda_inc = (T)lm.lambda_increase_base;
                    energy = e_test;
                    hist[iter % 2] = energy;
                    break;
                } else {
                    if (cb) cb(user, iter, 0, (double)energy, 0.0, (double)lambda, el);
                    if (lm.verbose && rank == 0)
                        printf("%5d%15s%15g%15g%15g%14gs\n", iter, "Rejected", (double)energy, 0.0, (double)lambda, el);
                    if (lambda > lam_max) { status = BA_EXCEEDED_LAMBDA_MAX; stop = true; break; }
                    lambda *= lambda_inc;
                    lambda_inc = std::pow(lambda_inc, (T)1.5);
                }
            }
            if (stop) break;
            if (iter > 2) {
                const T maxf = std::max(hist[0], hist[1]);
                if (std::fabs(energy - maxf) < tol_fun * energy) { status = BA_SUCCESS; break; } // before x = xTest (:419-428)
            }
            if ((rc = accept())) break;
        }
        if (lin_pending && !rc) { // stopped by max_trials right behind an enqueued linearisation
            if (!(rc = fetch_scalars())) { energy = h_scal[SC_ENERGY]; linearize_account(); }
        }
        if (lm.verbose && rank == 0) printf("--------------------------------------------------------------------------------\n");
        if (out) {
            out->status = status; out->iterations = iter; out->trials = trials; out->fun_evals = fun_evals;
            out->energy = (double)energy; out->lambda = (double)lambda;
            out->seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - tbeg).count();
            const long long nt = tm.n_trials - tm0.n_trials, nl = tm.n_linearize - tm0.n_linearize;
            const long long ng = tm.n_graph_trials - tm0.n_graph_trials;
            // graph replay: only the whole trial is timed (it includes the ~20 us test-energy evaluation)
            out->schur_ms = !nt ? 0.0 : ng == nt ? (tm.trial_ms - tm0.trial_ms) / nt
                                : ((tm.eliminate_ms - tm0.eliminate_ms) + (tm.schur_ms - tm0.schur_ms) + (tm.factor_ms - tm0.factor_ms) +
                                   (tm.backsub_ms - tm0.backsub_ms)) / (nt - ng > 0 ? nt - ng : 1);
            out->linearize_ms = nl ? (tm.linearize_ms - tm0.linearize_ms) / nl : 0.0;
        }
        return rc;
    }

    // which = 1: the backward sweep with the group at the head of its chain missing (and a short spin bound): the groups behind
    // it wait for unknowns that are never published -- the production path's reaction to that must be BA_ERR_HIP.
    int selftest(int which) override
    {
        if (which != 1) return BA_ERR_ARG;
        const int nblk = (D + NB - 1) / NB, groups = (nblk + 1) / 2;
        if (groups < 2 || groups > num_cus) return BA_ERR_ARG;
        hipLaunchKernelGGL((k_fill_sentinel<T>), dim3((Dp + 255) / 256), dim3(256), 0, st, Dp, d_dxc.p);
        hipLaunchKernelGGL((k_ldlt_backflow<T, NB>), dim3(groups), dim3(256), 0, st, D, ld, D, nblk, d_S.p, d_Winv.p, d_dxc.p,
                           d_scal.p + SC_ERR, 1 << 10, groups - 1);
        have_step = false;
        return fetch_scalars();
    }

    int time_phase(int phase, int reps, double lambda_d, double *ms) override
    {
        if (reps < 1 || !ms) return BA_ERR_ARG;
        double acc_ms = 0;
        int rcl = set_lambda((T)lambda_d);
        if (rcl) return rcl;
        HIPCHK(hipStreamSynchronize(st));
        HIPCHK(hipEventRecord(ev[EV_T0], st));
        for (int k = 0; k < reps; k++) {
            switch (phase) {
            case 0: launch_eval(false, cur); break;
            case 1: launch_eval(true, cur); launch_grad(cur); break;
            case 2: launch_eliminate(); break;
            case 3: launch_schur(); break;
            case 4:
                launch_schur(); // the factorisation is in place: rebuild S first (timed separately by phase 3)
                launch_post_reduce();
                launch_factor_solve();
                break;
            case 5: launch_backsub_retract(); break;
            case 6: // dense factorisation only (k_ldlt_panel + k_ldlt_step / k_ldlt_update): events around it, per rep
            case 7: // backward sweep only (k_ldlt_backstep)
                launch_schur();
                launch_post_reduce();
                if (phase == 6) HIPCHK(hipEventRecord(ev[EV_T2], st));
                launch_factor();
                if (phase == 6) HIPCHK(hipEventRecord(ev[EV_T3], st));
                if (phase == 7) HIPCHK(hipEventRecord(ev[EV_T2], st));
                launch_backsweep();
                if (phase == 7) HIPCHK(hipEventRecord(ev[EV_T3], st));
                HIPCHK(hipStreamSynchronize(st));
                acc_ms += ev_ms(EV_T2, EV_T3);
                break;
            default: return BA_ERR_ARG;
            }
        }
        HIPCHK(hipEventRecord(ev[EV_T1], st));
        HIPCHK(hipStreamSynchronize(st));
        HIPCHK(hipGetLastError());
        *ms = (phase >= 6 ? acc_ms : ev_ms(EV_T0, EV_T1)) / reps;
        have_step = false;
        return BA_OK;
    }
};

} // namespace

struct ba_solver {
    SolverBase *impl = nullptr;
};

extern "C" {

int ba_device_info(int device, char *name, size_t n, int *cus)
{
    int cnt = 0;
    if (hipGetDeviceCount(&cnt) != hipSuccess || cnt <= 0) return BA_ERR_HIP;
    if (device < 0) { if (hipGetDevice(&device) != hipSuccess) return BA_ERR_HIP; }
    if (device >= cnt) return BA_ERR_ARG;
    hipDeviceProp_t pr;
    if (hipGetDeviceProperties(&pr, device) != hipSuccess) return BA_ERR_HIP;
    if (name && n) { snprintf(name, n, "%s (%s)", pr.name, pr.gcnArchName); }
    if (cus) *cus = pr.multiProcessorCount;
    return BA_OK;
}

int ba_solver_create(const ba_problem *p, ba_solver_kind kind, ba_scalar scalar, int device, int shard_rank, int shard_world,
                     ba_solver **out)
{
    if (!p || !out || shard_world < 1 || shard_rank < 0 || shard_rank >= shard_world) return BA_ERR_ARG;
    if (kind != BA_QRKIT && kind != BA_QRCHOL && kind != BA_CHOLESKY && kind != BA_MOREQR) return BA_ERR_ARG;
    if (scalar != BA_F64 && scalar != BA_F32) return BA_ERR_ARG;
    *out = nullptr;
    int cnt = 0;
    if (hipGetDeviceCount(&cnt) != hipSuccess || cnt <= 0) {
        fprintf(stderr, "ba_mi355x: no HIP device -- the solver has no CPU path\n");
        return BA_ERR_HIP;
    }
    if (device >= 0) {
        if (device >= cnt) return BA_ERR_ARG;
        HIPCHK(hipSetDevice(device));
    }
    SolverBase *impl = (scalar == BA_F64) ? (SolverBase *)new (std::nothrow) Solver<double>() : (SolverBase *)new (std::nothrow) Solver<float>();
    if (!impl) return BA_ERR_NOMEM;
    int rc = impl->init(p, kind, shard_rank, shard_world);
    if (rc) { delete impl; return rc; }
    ba_solver *s = new (std::nothrow) ba_solver;
    if (!s) { delete impl; return BA_ERR_NOMEM; }
    s->impl = impl;
    *out = s;
    return BA_OK;
}

void ba_solver_free(ba_solver *s)
{
    if (!s) return;
    delete s->impl;
    delete s;
}

int ba_solver_set_allreduce(ba_solver *s, ba_allreduce_fn fn, void *user)
{
    if (!s) return BA_ERR_ARG;
    s->impl->ar_fn = fn; s->impl->ar_user = user;
    return BA_OK;
}

int ba_solver_set_stream(ba_solver *s, void *hip_stream)
{
    if (!s) return BA_ERR_ARG;
    if (s->impl->own_stream && s->impl->st) { (void)hipStreamSynchronize(s->impl->st); (void)hipStreamDestroy(s->impl->st); }
    s->impl->st = (hipStream_t)hip_stream;
    s->impl->own_stream = false;
    return BA_OK;
}

int ba_solver_keep_intermediates(ba_solver *s, int on)
{
    if (!s) return BA_ERR_ARG;
    s->impl->keep = on != 0;
    return BA_OK;
}

int ba_solver_shard(const ba_solver *s, int *p0, int *p1, int *o0, int *o1)
{
    if (!s) return BA_ERR_ARG;
    if (p0) *p0 = s->impl->sx.p0;
    if (p1) *p1 = s->impl->sx.p1;
    if (o0) *o0 = s->impl->sx.o0;
    if (o1) *o1 = s->impl->sx.o1;
    return BA_OK;
}

int ba_minimize(ba_solver *s, const ba_lm_params *lm, ba_trial_cb cb, void *user, ba_result *out)
{
    return s ? s->impl->minimize(lm, cb, user, out) : BA_ERR_ARG;
}
int ba_solver_linearize(ba_solver *s, double *energy, double *diag_max) { return s ? s->impl->linearize(energy, diag_max) : BA_ERR_ARG; }
int ba_solver_try_step(ba_solver *s, double lambda, double *energy_test, double *rho_scale, double *dx_norm)
{
    return s ? s->impl->try_step(lambda, energy_test, rho_scale, dx_norm) : BA_ERR_ARG;
}
int ba_solver_accept(ba_solver *s) { return s ? s->impl->accept() : BA_ERR_ARG; }
int ba_solver_stats(ba_solver *s, double *out4) { return (s && out4) ? s->impl->stats(out4) : BA_ERR_ARG; }
int ba_solver_get(ba_solver *s, int what, double *out, size_t n) { return (s && out) ? s->impl->get(what, out, n) : BA_ERR_ARG; }
int ba_solver_set_state(ba_solver *s, const double *cam15, const double *pts) { return s ? s->impl->set_state(cam15, pts) : BA_ERR_ARG; }

int ba_solver_timing(ba_solver *s, ba_timing *out, int reset)
{
    if (!s) return BA_ERR_ARG;
    if (out) *out = s->impl->tm;
    if (reset) s->impl->tm = ba_timing{};
    return BA_OK;
}

int ba_solver_time_phase(ba_solver *s, int phase, int reps, double lambda, double *ms_per_launch)
{
    return s ? s->impl->time_phase(phase, reps, lambda, ms_per_launch) : BA_ERR_ARG;
}

int ba_solver_selftest(ba_solver *s, int which) { return s ? s->impl->selftest(which) : BA_ERR_ARG; }

} // extern "C"
