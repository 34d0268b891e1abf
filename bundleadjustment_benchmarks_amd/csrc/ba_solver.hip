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
#include "ba_qr.hip.h"

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
constexpr int NSCAL = 24;    // device scalar slots
// SC_ETEST .. SC_ERR are contiguous: the ONE scalar all-reduce behind a sharded trial sums these five
enum { SC_ENERGY = 0, SC_DMAX_P = 1, SC_ETEST = 2, SC_RHO_P = 3, SC_DN_P = 4,
       SC_GUARD = 5 /* the previous trial's decision (accepted + 2 stop) of this shard: summed, it must be world x the same (k_lm_control) */,
       SC_ERR = 6 /* device error word: a BA_DEVERR_* code written by a kernel whose in-launch hand-off wait ran out (sharded: summed, so
                     that every shard sees a time-out of any shard on the same trial) */,
       SC_DMAX_C = 7, SC_ST0 = 8 /* ..11 stats */, SC_LAMBDA = 12 /* lambda of the current trial, read by the kernels */,
       SC_ZERO = 13 /* always 0: the 'lambda' of MOREQR's outer factorisation */,
       SC_ELOC = 15 /* sharded: this shard's part of the energy of the latest linearisation (rides on the next packed all-reduce) */,
       SC_RHO_C = 16, SC_DN_C = 17 };
constexpr int N_STEP_SCALARS = 5; // SC_ETEST, SC_RHO_P, SC_DN_P, SC_GUARD, SC_ERR
enum { EV_T0 = 0, EV_T1, EV_T2, EV_T3, EV_T4, EV_T5, EV_T6, EV_L0, EV_L1, EV_N };
constexpr int RING_EV = 8; // event slots of the trials in flight under ba_minimize (LM_DEPTH + the ones not yet harvested)
constexpr int RING_NE = 6; // per slot: trial start | before / after the matrix all-reduce | before / after the scalar all-reduce | after control + linearisation

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
    bool poisoned = false; // the watchdog gave up on a launch that never finished: every later call fails, nothing is freed
    int recoveries = 0;    // trials repeated through the launch-per-step factorisation after a hand-off time-out
    ba_allreduce_fn ar_fn = nullptr;
    void *ar_user = nullptr;
    void *comm = nullptr; // ncclComm_t of the shard group (ba_solver_comm_init)
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
        d_dchunk_ptr, d_cam_dchunk_ptr, d_cam_obs, d_qr_pts, d_flags, d_pperm /* column permutation of every point's 3x3 block */;
    DevBuf<int4> d_chunk_info; // per chunk of the pair kernel: first entry, count | BA_CHUNK_SINGLE, cameras hi | lo << 16, chunk id
    DevBuf<int2> d_ent;        // per entry: row observation, column observation (~point for a self entry)
    DevBuf<int> d_wave_ptr;    // per wavefront of the pair kernel: its range of chunk descriptors
    DevBuf<int> d_red_pairs;   // pairs k_schur_reduce writes: those without entries and those with several chunks
    int schur_window = 0;
    int nred = 0;
    const bool no_fold = getenv("BA_NO_FOLD") != nullptr; // dev switch: every launch of its own again (A/B timing on one box)
    int schur_grid = 1, schur_wgs = 4 /* workgroups of k_schur_pairs per CU */, schur_bands = 8, schur_nband = 1, schur_gb = 2 /* groups of four entries per batch */;
    // state: x = d_cam[0], d_pts[0]; xTest = d_cam[1], d_pts[1] (x = xTest is a device-side copy, k_commit)
    // linearisation at x (r, J, J^T r, block diagonals, MOREQR's outer factors): one set
    DevBuf<T> d_r, d_Jc, d_Jp, d_JcA, d_U0, d_gp, d_V, d_gc, d_rec0, d_dinv0, d_tvec0, d_tri0;
    DevBuf<T> d_cam[2], d_pts[2], d_meas, d_gcg, d_dslab, d_rec, d_dinv, d_tvec, d_tri,
        d_slab, d_S, d_pack, d_Skeep, d_Wp, d_Winv, d_dxc, d_dxp, d_part_e, d_part_pm, d_part_bs, d_part_st, d_scal;
    DevBuf<ba_lm_dev<T>> d_lm; // LM state of the device-side step control (k_lm_control)
    // QRKIT (single shard): dense J2bot (+ rhs column) for the Householder QR of the right block, its reflector scalars, the thin Q rows
    DevBuf<T> d_qA, d_qtau, d_q1obs, d_q1lam;
    size_t q_lda = 0, q_tau_stride = 0;
    hipStream_t st_qr = nullptr;   // second stream of the dense QR: trailing updates beside the panel's chunk chain (ba_qr_solve)
    hipStream_t st_qr3 = nullptr;  // third one: the look-ahead updates (the next panel's columns)
    hipEvent_t ev_qr[3] = {nullptr, nullptr, nullptr};
    ba_qr_side qr_side() const
    {
        ba_qr_side sd;
        sd.st2 = st_qr; sd.st3 = st_qr3; sd.ev_chunk = ev_qr[0]; sd.ev_apply = ev_qr[1]; sd.ev_next = ev_qr[2];
        return sd;
    }
    int q_rows = 0;
    // QRKIT / QRSPQR always run the dense QR of J2bot -- sharded too (distributed TSQR: launch_qr_stack), never QRCHOL's normal
    // equations under another name
    // (round 4) ... and so does MOREQR: its right block is the dense QR of [rows left by the per-point QRs ; R22 ; sqrt(lambda) I]
    // (BacktrackLevMarqMore.h:297-345), R22 from one dense QR of J2bot(lambda = 0) per outer iteration (:288) -- no S, no LDL^T, sharded
    // through the same TSQR stack.  BA_MOREQR_QR=0 (read at solver creation) selects rounds 1 - 3's variant instead, which eliminates the
    // points by QR and then factors S = (Jc'Jc + lambda I) - sum Z Z' by LDL^T: 20x faster, but normal equations.
    bool more_qr_on = true;
    bool dense_qr() const { return kind == BA_QRKIT || kind == BA_QRSPQR || more_qr(); }
    bool more_qr() const { return kind == BA_MOREQR && more_qr_on; }
    DevBuf<T> d_dbg; // diagnostic buffer (BA_DBG_ATB)
    DevBuf<T> d_mQl, d_mQR, d_R22; // MOREQR: the inner point blocks' thin Q (lambda rows, R1 rows: [Ml][9] each); R22 | c2 of the outer QR, D x (D + 1)
    int outer_rows() const { return 2 * Kl + 3 * Ml + D; }               // J2bot (QRKIT / QRSPQR per trial; MOREQR per outer iteration, lambda = 0)
    int inner_rows() const { return more_qr() ? 6 * Ml + 2 * D : outer_rows(); } // the matrix a TRIAL factors
    DevBuf<T> d_qB; // sharded: the stack of the shards' R factors (+ rhs column), behind it g_c and the energy (one all-reduce)
    size_t qb_ld() const { return (size_t)world * D + 64; }
    size_t qb_nmat() const { return qb_ld() * (size_t)(D + 1); }
    size_t qb_count() const { return qb_nmat() + (size_t)D + 1; }
    ba_lm_host *h_log = nullptr, *d_log = nullptr; // table rows + progress counter in pinned host memory (host / device address)
    T h_scal[NSCAL];
    T *h_lam = nullptr; // pinned staging word for lambda
    // one LM iteration as hipGraphs: world == 1: g_trial = elimination ... test energy, control, x = xTest, linearisation in ONE graph
    // (two graphs with events between them left 18 + 27 us of idle GPU per iteration at config 4: 0.872 -> 0.852 ms);
    // sharded: g_a (elimination, assembly, pack) | all-reduce | g_b (unpack ... test energy) | all-reduce | g_ctl
    hipGraphExec_t g_trial = nullptr, g_a = nullptr, g_b = nullptr, g_ctl = nullptr;
    bool use_graph = true;
    hipEvent_t ev[EV_N] = {};
    struct EvSlot { hipEvent_t e[RING_NE]; };
    EvSlot ring[RING_EV] = {};
    int gK = 0, gM = 0, gB = 0; // grids (observations, points, points x 8 lanes)
    // Workgroups of k_eval: point-aligned observation ranges (at most 256, whole points) whenever no point has more than 256
    // observations -- for the residual-only evaluation too, so that an energy is the same bits whichever instantiation sums it.
    // fuse: the linearisation behind an accepted step then also does the point part of the gradient and (CHOLESKY) the elimination
    // of the trial that follows (k_eval<T, true, FUSE>); BA_NO_FUSE=1 keeps the separate launches on the same ranges (A/B, bit-equal).
    int gE = 0;
    bool fuse = false;
    DevBuf<int> d_eb;
    bool have_step = false;
    int num_cus = 256; // of the device the solver lives on
    double wall_khz = 1e5;
    // A hand-off between workgroups of one launch that timed out (device error word) is survivable: the trial is repeated with the
    // factorisation and the back sweep as one launch per step (nobody waits for anybody), and the solver stays in that mode.
    bool safe_factor = false;
    int fault_rowflag = 0;    // self-test (2): the fused steps' row workgroups stay silent, the panel's wait is short
    double spin_next_s = 0;   // self-test (3): a kernel of that many seconds in front of the next trial
    double watchdog_s = 600;  // no LM row for that long = a hung launch (BA_WATCHDOG_S)

    ~Solver() override
    {
        if (poisoned) return; // (ba_solver_free does not even get here: a HIP call could block on the launch that never ended)
        for (auto &e : ev)
            if (e) (void)hipEventDestroy(e);
        for (auto &sl : ring)
            for (auto &e : sl.e)
                if (e) (void)hipEventDestroy(e);
        for (hipGraphExec_t g : {g_trial, g_a, g_b, g_ctl})
            if (g) (void)hipGraphExecDestroy(g);
        if (h_lam) (void)hipHostFree(h_lam);
        if (h_log) (void)hipHostFree((void *)h_log);
        if (comm) ba_rccl_destroy(comm);
        if (own_stream && st) (void)hipStreamDestroy(st);
        if (st_qr) (void)hipStreamDestroy(st_qr);
        if (st_qr3) (void)hipStreamDestroy(st_qr3);
        for (hipEvent_t e : ev_qr) if (e) (void)hipEventDestroy(e);
    }

    int init(const ba_problem *p, ba_solver_kind k, int rk, int wd) override
    {
        kind = k; rank = rk; world = wd;
        more_qr_on = !(getenv("BA_MOREQR_QR") != nullptr && atoi(getenv("BA_MOREQR_QR")) == 0);
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
        gE = gK;
        if (sx.kmax <= 256 && Kl > 0) {
            std::vector<int> eb(1, 0);
            int fill = 0; // observations in the current range
            for (int j = 0; j < Ml; j++) {
                const int k = sx.pt_ptr[j + 1] - sx.pt_ptr[j];
                if (fill + k > 256) { eb.push_back(sx.pt_ptr[j]); fill = 0; }
                fill += k;
            }
            eb.push_back(Kl);
            gE = (int)eb.size() - 1;
            if ((rc = d_eb.upload(eb))) return rc;
            fuse = getenv("BA_NO_FUSE") == nullptr;
        }
        if (kind != BA_CHOLESKY && sx.kmax > 1024) return BA_ERR_ARG; // more than 1024 observations of one point: not supported by k_elim_qr
        if (!st) { HIPCHK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking)); own_stream = true; }
        for (auto &e : ev) HIPCHK(hipEventCreate(&e));
        for (auto &sl : ring)
            for (auto &e : sl.e) HIPCHK(hipEventCreate(&e));
        HIPCHK(hipHostMalloc((void **)&h_lam, sizeof(T)));
        // the LM table rows and the progress counter the control kernel writes: pinned, mapped, coherent host memory
        HIPCHK(hipHostMalloc((void **)&h_log, sizeof(ba_lm_host), hipHostMallocMapped | hipHostMallocCoherent));
        memset((void *)h_log, 0, sizeof(ba_lm_host));
        HIPCHK(hipHostGetDevicePointer((void **)&d_log, (void *)h_log, 0));
        use_graph = getenv("BA_NO_GRAPH") == nullptr;
        dist_factor = getenv("BA_DIST_FACTOR") != nullptr && atoi(getenv("BA_DIST_FACTOR")) != 0;
        if (const char *wd = getenv("BA_WATCHDOG_S")) { const double v = atof(wd); if (v > 0) watchdog_s = v; }
        {
            int dev = 0;
            hipDeviceProp_t pr;
            if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess && pr.multiProcessorCount > 0)
                num_cus = pr.multiProcessorCount;
            int khz = 0; // rate of wall_clock64(), the constant-frequency counter behind the per-trial device times (100 MHz on gfx950)
            if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, dev) == hipSuccess && khz > 0) wall_khz = khz;
        }
#define UP(buf, vec) if ((rc = buf.upload(vec))) return rc
        UP(d_obs_cam, sx.obs_cam); UP(d_obs_pt, sx.obs_pt); UP(d_pt_ptr, sx.pt_ptr); UP(d_pair_hi, sx.pair_hi);
        UP(d_pair_lo, sx.pair_lo);
        UP(d_pair_chunk_ptr, sx.pair_chunk_ptr); UP(d_dchunk_ptr, sx.dchunk_ptr); UP(d_cam_dchunk_ptr, sx.cam_dchunk_ptr);
        UP(d_cam_obs, sx.cam_obs); UP(d_qr_pts, sx.qr_pts);
#undef UP
        {
            // Records beyond the Infinity Cache (256 MB; config 5: 1 GB): the kernel is bound by the traffic between the L2s and
            // memory, and fewer wavefronts in flight walking the pair list side by side leave more of a row camera's records in the
            // L2 (2.23 -> 2.07 ms at config 5; nothing either way at config 4, whose 58 MB of records stay in the Infinity Cache).
            if ((unsigned long long)Kl * BA_REC * sizeof(T) > (256ull << 20)) { schur_wgs = 2; schur_window = 1; }
            if (const char *ev = getenv("BA_SCHUR_GB")) schur_gb = atoi(ev); // (A/B only: deeper batches measured SLOWER at configs 4 and 5, profiles/EXPERIMENTS.md 4)
            if (const char *ev = getenv("BA_SCHUR_WGS")) schur_wgs = std::max(1, std::min(8, atoi(ev)));
            if (const char *ev = getenv("BA_SCHUR_BANDS")) schur_bands = atoi(ev);
            // Chunks dealt to the wavefronts of the persistent pair kernel, longest first, always to the least loaded wavefront
            // (cost = batches of eight entries + a constant per chunk); a wavefront's list is then walked in chunk order.
            if (N > 65535) return BA_ERR_ARG; // (hi | lo << 16 in the chunk descriptor; a reduced matrix of that size would not fit anyway)
            if ((unsigned long long)Kl * BA_REC * sizeof(T) >= 0xf0000000ull) return BA_ERR_ARG; // 32-bit record offsets: <= 15 M observations per shard (fp64)
            {
                // every wavefront of the persistent grid must be resident at once (the dealing assumes they run side by side)
                int nb = 0;
                const hipError_t oe = kind == BA_CHOLESKY ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_schur_pairs<T, true, 0, 2>, 256, 0)
                                                          : hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_schur_pairs<T, false, 0, 2>, 256, 0);
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
            // The dealing goes window by window through the band (schur_window chunks per wavefront and window; 0 = the whole band
            // is one window): the loads carry over, so the balance is the same, but a wavefront's list now holds a few chunks of
            // EVERY window, i.e. all wavefronts of the band walk through the (row camera, column camera) order side by side and the
            // records of one row camera (its observations: ~1 MB at config 5) are in the XCD's L2 when the next column camera's
            // chunk asks for them.
            if (const char *ev = getenv("BA_SCHUR_WINDOW")) schur_window = std::max(0, atoi(ev));
            for (int b = 0; b < nband; b++) {
                typedef std::pair<long long, int> load_t; // (load, wavefront): min-heap
                std::priority_queue<load_t, std::vector<load_t>, std::greater<load_t>> heap;
                for (int w = 0; w < Wb; w++) heap.push(load_t(0, b * Wb + w));
                const int nb_ = bptr[b + 1] - bptr[b];
                const long long win = schur_window > 0 ? (long long)schur_window * Wb : (long long)std::max(nb_, 1);
                for (long long w0 = bptr[b]; w0 < bptr[b + 1]; w0 += win) {
                    const int w1 = (int)std::min<long long>(w0 + win, bptr[b + 1]);
                    order.assign(w1 - (int)w0, 0);
                    std::iota(order.begin(), order.end(), (int)w0);
                    std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return cost(x) > cost(y); });
                    for (int c : order) {
                        load_t t = heap.top();
                        heap.pop();
                        owner[c] = t.second;
                        wptr[t.second + 1]++;
                        heap.push(load_t(t.first + cost(c), t.second));
                    }
                }
            }
            schur_nband = nband;
            for (int w = 0; w < W; w++) wptr[w + 1] += wptr[w];
            std::vector<int4> ci((size_t)sx.nchunks);
            {
                std::vector<int> cur(wptr.begin(), wptr.end() - 1);
                for (int c = 0; c < sx.nchunks; c++) { // increasing chunk id inside every wavefront's list
                    const int q = sx.chunk_pair[c];
                    // (a diagonal pair always goes through k_schur_reduce, which puts lambda on the diagonal when k_post_reduce is folded away)
                    const bool single = sx.pair_chunk_ptr[q + 1] - sx.pair_chunk_ptr[q] == 1 && sx.pair_hi[q] != sx.pair_lo[q];
                    ci[cur[owner[c]]++] = make_int4(sx.chunk_ptr[c], (sx.chunk_ptr[c + 1] - sx.chunk_ptr[c]) | (single ? BA_CHUNK_SINGLE : 0),
                                                    sx.pair_hi[q] | (sx.pair_lo[q] << 16), c);
                }
            }
            if ((rc = d_wave_ptr.upload(wptr))) return rc;
            std::vector<int> red;
            for (int q = 0; q < sx.npairs; q++)
                if (sx.pair_chunk_ptr[q + 1] - sx.pair_chunk_ptr[q] != 1 || sx.pair_hi[q] == sx.pair_lo[q]) red.push_back(q);
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
        AL(d_r, 2 * K1); if (kind != BA_CHOLESKY) AL(d_Jc, 18 * K1); /* CHOLESKY: JcA alone */ AL(d_JcA, 20 * K1); AL(d_Jp, 6 * K1); AL(d_U0, 6 * M1); AL(d_gp, 3 * M1);
        AL(d_V, (size_t)81 * N); AL(d_gc, (size_t)D);
        if (kind == BA_MOREQR) { AL(d_rec0, (size_t)BA_REC * K1); AL(d_dinv0, 3 * M1); AL(d_tvec0, 3 * M1); AL(d_tri0, 6 * M1); }
        if ((rc = d_lm.alloc(1))) return rc;
        if ((rc = d_pperm.upload(std::vector<int>(M1, 0 | (1 << 2) | (2 << 4))))) return rc; // identity (CHOLESKY never pivots)
        if (dense_qr()) {
            if (const char *ev = getenv("BA_QR_DBG")) { // diagnostic bits (ba_qr.hip.h: ba_qr_dbg_flag)
                const int bits = atoi(ev);
                HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(ba_qr_dbg_flag), &bits, sizeof(int)));
            }
            if (const char *ev = getenv("BA_QR_HW_SQRT")) { // diagnostic switch (ba_qr.hip.h: ba_qr_sqrt)
                const int on = atoi(ev); // 0 (default): v_sqrt_f32 + one Newton step, 1: the bare instruction, 2: sqrtf
                HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(ba_qr_hw_sqrt_flag), &on, sizeof(int)));
            }
            // J2bot is dense: (2K + 3M + D) x (D + 1) scalars (config 3: 256 MB in fp32; a problem whose J2bot does not fit is refused)
            // (a camera may see a point more than once: k_qrkit_build / k_more_build add such observations' blocks up -- rounds 2 - 3 refused them)
            q_rows = std::max(outer_rows(), inner_rows());
            if (more_qr()) { AL(d_mQl, 9 * M1); AL(d_mQR, 9 * M1); AL(d_R22, (size_t)D * (D + 1)); AL(d_dbg, (size_t)D); }
            if (!getenv("BA_QR_ONE_STREAM")) {
                HIPCHK(hipStreamCreateWithFlags(&st_qr, hipStreamNonBlocking));
                // (look-ahead on a third stream: measured, not the default -- 2.27 - 2.34 against 2.38 ms stand-alone, but 375 against
                // 418 LM it/s inside the captured iteration graph, profiles/EXPERIMENTS.md 6.2)
                if (getenv("BA_QR_LOOKAHEAD")) HIPCHK(hipStreamCreateWithFlags(&st_qr3, hipStreamNonBlocking));
                for (auto &e : ev_qr) HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
            }
            q_lda = (size_t)q_rows + 64;
            q_tau_stride = (size_t)((std::max(q_rows, world * D) + ba_qr_cfg<T>::CH - 1) / ba_qr_cfg<T>::CH + 2) * BA_QR_PB * BA_QR_PB; // a 32 x 32 T factor per chunk (of J2bot's rows or, sharded, of the stack's)
            AL(d_qA, q_lda * (size_t)(D + 1)); AL(d_qtau, (st_qr3 ? 2 : 1) * BA_QR_TAU_LEVELS * q_tau_stride); AL(d_q1obs, 6 * K1); AL(d_q1lam, 9 * M1);
            if (world > 1) AL(d_qB, qb_count());
        }
        AL(d_gcg, (size_t)D); AL(d_dslab, (size_t)BA_SLAB * (sx.ndchunks > 0 ? sx.ndchunks : 1));
        AL(d_rec, (size_t)BA_REC * K1); AL(d_dinv, 3 * M1); AL(d_tvec, 3 * M1); AL(d_tri, 6 * M1);
        AL(d_slab, (size_t)BA_SLAB * (sx.nchunks > 0 ? sx.nchunks : 1));
        AL(d_S, (size_t)ld * (Dp + 64)); AL(d_Wp, (size_t)4 * ld * NB); AL(d_Winv, (size_t)((D + NB - 1) / NB) * NB * NB); AL(d_dxc, (size_t)2 * Dp + 2 * NB); /* the solution + the back sweep's hand-over vector */ AL(d_dxp, 3 * M1);
        if ((rc = d_flags.alloc((size_t)Dp / NB + 2))) return rc;
        AL(d_part_e, (size_t)gE); AL(d_part_pm, (size_t)gM);
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

    // Sum / max over the ranks that shard the problem, in place, ordered on the solver's stream: RCCL inside the library
    // (ba_solver_comm_init) or the host layer's callback (ba_solver_set_allreduce; the gloo tests).
    int allreduce(void *buf, size_t count, int op)
    {
        if (world <= 1 && !comm) return BA_OK;
        if (comm) return ba_rccl_allreduce(comm, buf, count, sizeof(T) == 8, op, (void *)st);
        if (!ar_fn) return BA_ERR_COMM;
        return ar_fn(ar_user, buf, count, sizeof(T) == 8 ? BA_F64 : BA_F32, op, (void *)st) ? BA_ERR_COMM : BA_OK;
    }
    bool sharded() const { return world > 1 || comm != nullptr; }
    // (round 4) the distributed factor's collectives: a broadcast from the owner of a block column, a reduce-scatter to the owners
    int bcast(void *buf, size_t count, int root)
    {
        if (world <= 1 && !comm) return BA_OK;
        if (comm) return ba_rccl_broadcast(comm, buf, count, sizeof(T) == 8, root, (void *)st);
        if (!ar_fn) return BA_ERR_COMM;
        return ar_fn(ar_user, buf, count, sizeof(T) == 8 ? BA_F64 : BA_F32, BA_OP_BCAST | (root << 8), (void *)st) ? BA_ERR_COMM : BA_OK;
    }
    int reduce_scatter(void *buf, size_t count_per_rank)
    {
        if (world <= 1 && !comm) return BA_OK;
        if (comm) return ba_rccl_reduce_scatter(comm, buf, count_per_rank, sizeof(T) == 8, rank, (void *)st);
        if (!ar_fn) return BA_ERR_COMM;
        return ar_fn(ar_user, buf, count_per_rank, sizeof(T) == 8 ? BA_F64 : BA_F32, BA_OP_REDUCE_SCATTER, (void *)st) ? BA_ERR_COMM : BA_OK;
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
        report_device_error((int)h_scal[SC_ERR]);
        h_scal[SC_ERR] = 0;
        (void)hipMemsetAsync(d_scal.p + SC_ERR, 0, sizeof(T), st);
        return BA_ERR_HIP;
    }
    static void report_device_error(int code)
    {
        fprintf(stderr, "ba_mi355x: device error %d: %s\n", code,
                code == BA_DEVERR_DIVERGED ? "k_lm_control: the shards did not take the same accept / stop decision"
                : code >= BA_DEVERR_SWEEP  ? "k_ldlt_backflow: an unknown of the backward sweep was never published"
                                           : "k_ldlt_step: the look-ahead update of a row block was never announced");
    }

    double ev_ms(hipEvent_t a, hipEvent_t b)
    {
        float ms = 0;
        if (hipEventElapsedTime(&ms, a, b) != hipSuccess) return 0;
        return ms;
    }
    double ev_ms(int a, int b) { return ev_ms(ev[a], ev[b]); }

    // which: 0 = x, 1 = xTest
    // commit (with jac, which = 1): the linearisation AT xTest also performs x = xTest (k_commit's copy rides on k_eval)
    // fused (with jac): the point part of the gradient and, CHOLESKY, the next trial's elimination in the same pass (k_eval<T, true, FUSE>)
    void launch_eval(bool jac, int which, const int *go = nullptr, bool commit = false, bool fused = false)
    {
        const T tau2 = tau * tau;
        ba_fuse_args<T> fa{};
        fa.eb = d_eb.p; // (nullptr when a point has more than 256 observations: plain runs of 256)
        fa.pt_ptr = d_pt_ptr.p; fa.lam = d_scal.p + SC_LAMBDA; fa.U0 = d_U0.p; fa.gp = d_gp.p;
        fa.rec = d_rec.p; fa.dinv = d_dinv.p; fa.tvec = d_tvec.p; fa.tri = d_tri.p; fa.fresh = &d_lm.p->rec_fresh;
#define BA_EVAL(J, F, ...) hipLaunchKernelGGL((k_eval<T, J, F, ##__VA_ARGS__>), dim3(gE), dim3(256), 0, st, Kl, N, Ml, d_cam[which].p, d_pts[which].p, d_obs_cam.p, d_obs_pt.p, \
                                         d_meas.p, tau2, d_r.p, d_Jc.p, d_Jp.p, d_JcA.p, d_part_e.p, go, commit ? d_cam[0].p : (T *)nullptr,              \
                                         commit ? d_pts[0].p : (T *)nullptr, fa)
        // (CHOLESKY keeps the camera blocks in the AoS records alone: SOA = false, d_Jc is not even allocated)
        if (!jac) BA_EVAL(false, 0);
        else if (kind == BA_CHOLESKY) { if (fused && fuse) BA_EVAL(true, 2, false); else BA_EVAL(true, 0, false); }
        else if (fused && fuse) BA_EVAL(true, 1);
        else BA_EVAL(true, 0);
#undef BA_EVAL
    }

    // tail: the energy reduction that closes the linearisation, as one more block of the last launch (+ the control segment's end stamp)
    void launch_grad(const int *go = nullptr, const ba_red_job *tail = nullptr, bool points = true /* false: the fused k_eval has done them */)
    {
        if (!points) {
            if (sx.ndchunks > 0)
                hipLaunchKernelGGL((k_cam_gram<T>), dim3((sx.ndchunks + 7) / 8), dim3(256), 0, st, sx.ndchunks, Kl,
                                   d_dchunk_ptr.p, d_cam_obs.p, d_JcA.p, d_dslab.p, go);
        } else if (sx.ndchunks > 0 && gM <= 2 * num_cus && !no_fold) // a point part of one round of workgroups: both in one launch (k_grad_prep)
            hipLaunchKernelGGL((k_grad_prep<T>), dim3(gM + (sx.ndchunks + 7) / 8), dim3(256), 0, st, gM, Ml, Kl, d_pt_ptr.p, d_Jp.p, d_r.p, d_U0.p, d_gp.p,
                               d_part_pm.p, sx.ndchunks, d_dchunk_ptr.p, d_cam_obs.p, d_JcA.p, d_dslab.p, go);
        else {
            hipLaunchKernelGGL((k_point_prep<T>), dim3(gM), dim3(256), 0, st, Ml, Kl, d_pt_ptr.p, d_Jp.p, d_r.p, d_U0.p, d_gp.p, d_part_pm.p, go);
            if (sx.ndchunks > 0)
                hipLaunchKernelGGL((k_cam_gram<T>), dim3((sx.ndchunks + 7) / 8), dim3(256), 0, st, sx.ndchunks, Kl,
                                   d_dchunk_ptr.p, d_cam_obs.p, d_JcA.p, d_dslab.p, go);
        }
        hipLaunchKernelGGL((k_cam_gram_reduce<T>), dim3((N * BA_SLAB + 255) / 256 + (tail ? 1 : 0)), dim3(256), 0, st, N, d_cam_dchunk_ptr.p,
                           d_dslab.p, d_V.p, d_gc.p, go, tail ? *tail : ba_red_job{nullptr, 0, 0, 0}, d_scal.p, tail ? &d_lm.p->t_end : (long long *)nullptr);
    }

    // m_functor(x, r); energy; m_functor.df(x, J); JtRes; column norms (BacktrackLevMarqQRChol.h:257-280), host-synchronous
    int linearize(double *energy, double *diag_max) override
    {
        int rc;
        if ((rc = linearize_enqueue(diag_max != nullptr, nullptr))) return rc;
        if (more_qr() && (rc = more_outer_finish(nullptr))) return rc;
        if (sharded()) { // the kernels left this shard's part in SC_ELOC (it also rides on the next trial's packed all-reduce)
            HIPCHK(hipMemcpyAsync(d_scal.p + SC_ENERGY, d_scal.p + SC_ELOC, sizeof(T), hipMemcpyDeviceToDevice, st));
            if ((rc = allreduce(d_scal.p + SC_ENERGY, 1, 0))) return rc;
        }
        if (diag_max && (rc = allreduce(d_scal.p + SC_DMAX_P, 1, 1))) return rc;
        if ((rc = fetch_scalars())) return rc;
        HIPCHK(hipGetLastError());
        tm.linearize_ms += ev_ms(EV_L0, EV_L1);
        tm.n_linearize++;
        if (energy) *energy = (double)h_scal[SC_ENERGY];
        if (diag_max) *diag_max = std::max((double)h_scal[SC_DMAX_P], (double)h_scal[SC_DMAX_C]);
        return BA_OK;
    }

    // The launches of the linearisation at x.  go != nullptr: conditional on the device-side step control (the kernels return
    // at once unless the trial in front of them was accepted).  The energy lands in SC_ENERGY, or -- sharded -- this shard's part
    // in SC_ELOC.
    int linearize_enqueue(bool want_dmax, const int *go)
    {
        int rc;
        if (!go) HIPCHK(hipEventRecord(ev[EV_L0], st));
        // behind a trial (go != nullptr): the linearisation is AT xTest and carries x = xTest along (no k_commit launch), and the
        // energy sum rides on the last launch of launch_grad (no k_reduce_scalars launch) unless MOREQR's outer QR follows it
        ba_red_jobs jobs{};
        int nj = 0;
        jobs.j[nj++] = {d_part_e.p, gE, 0, sharded() ? SC_ELOC : SC_ENERGY};
        const bool tail = go != nullptr && !want_dmax && kind != BA_MOREQR;
        // behind a trial the point part of J^T r / J^T J and (CHOLESKY) the elimination of the next trial are part of the k_eval launch;
        // the first, host-synchronous linearisation (lambda0 is not known yet, max diag J^T J is wanted) keeps the separate launches
        const bool fz = fuse && go != nullptr && !want_dmax;
        launch_eval(true, go ? 1 : 0, go, go != nullptr, fz);
        launch_grad(go, tail ? &jobs.j[0] : nullptr, !fz);
        if (kind == BA_MOREQR) { // m_solver.compute(J) + Q^T r, once per outer iteration (BacktrackLevMarqMore.h:288-291): the point blocks ...
            launch_elim_qr(d_scal.p + SC_ZERO, d_rec0.p, d_dinv0.p, d_tvec0.p, d_tri0.p, go, /*thin Q for J2bot*/ more_qr());
            if (more_qr() && (rc = launch_more_outer(go))) return rc; // ... and the dense QR of J2bot(lambda = 0); its part 2 follows where a collective may stand
        }
        if (tail) { have_step = false; return BA_OK; }
        if (want_dmax) {
            // max diag(J^T J): point part per shard, camera part from the (summed over shards) diagonal of J_c^T J_c
            T *tmp = d_dxc.p;
            hipLaunchKernelGGL((k_vdiag<T>), dim3((D + 255) / 256), dim3(256), 0, st, N, d_V.p, tmp);
            if ((rc = allreduce(tmp, (size_t)D, 0))) return rc;
            jobs.j[nj++] = {d_part_pm.p, gM, 1, SC_DMAX_P};
            jobs.j[nj++] = {tmp, D, 1, SC_DMAX_C};
        }
        hipLaunchKernelGGL((k_reduce_scalars<T>), dim3(nj), dim3(256), 0, st, jobs, d_scal.p, go, &d_lm.p->t_end);
        if (!go) HIPCHK(hipEventRecord(ev[EV_L1], st));
        have_step = false;
        return BA_OK;
    }

    void launch_eliminate()
    {
        if (kind == BA_CHOLESKY) {
            hipLaunchKernelGGL((k_elim_chol<T>), dim3(gK), dim3(256), 0, st, Kl, Ml, d_obs_pt.p, d_pt_ptr.p, d_JcA.p, d_Jp.p,
                               d_U0.p, d_gp.p, d_scal.p + SC_LAMBDA, d_rec.p, d_dinv.p, d_tvec.p, d_tri.p,
                               fuse ? (const int *)&d_lm.p->rec_fresh : (const int *)nullptr);
        } else if (kind == BA_MOREQR) {
            if (Kl > 0) // BacktrackLevMarqMore.h:297-345, the per-trial QR of [R ; sqrt(lambda) I]
                hipLaunchKernelGGL((k_more_trial<T>), dim3(gK), dim3(256), 0, st, Kl, Ml, d_obs_pt.p, d_pt_ptr.p, d_scal.p + SC_LAMBDA,
                                   d_rec0.p, d_tri0.p, d_tvec0.p, d_rec.p, d_dinv.p, d_tvec.p, d_tri.p, d_mQl.p, d_mQR.p);
        } else {
            launch_elim_qr(d_scal.p + SC_LAMBDA, d_rec.p, d_dinv.p, d_tvec.p, d_tri.p, nullptr, dense_qr());
        }
    }

    // per-point QR, one launch per non-empty track-length bucket (ba_structure: lanes per point x observations per lane)
    void launch_elim_qr(const T *lam, T *rec, T *dinv, T *tvec, T *tri, const int *go, bool thin_q = false)
    {
        T *qo = thin_q ? d_q1obs.p : nullptr, *ql = thin_q ? d_q1lam.p : nullptr;
#define BA_QR(B, L, SLOTS)                                                                                                       \
        if (sx.qr_bucket_ptr[B + 1] > sx.qr_bucket_ptr[B]) {                                                                   \
            const int np_ = sx.qr_bucket_ptr[B + 1] - sx.qr_bucket_ptr[B];                                                       \
            hipLaunchKernelGGL((k_elim_qr<T, L, SLOTS>), dim3(((size_t)np_ * L + 255) / 256), dim3(256), 0, st, np_,           \
                               d_qr_pts.p + sx.qr_bucket_ptr[B], Ml, Kl, d_pt_ptr.p, d_Jc.p, d_Jp.p, d_r.p, lam, rec, dinv, tvec, tri, go, qo, ql, d_pperm.p); \
        }
        BA_QR(0, 8, 4)
        BA_QR(1, 16, 4)
        BA_QR(2, 32, 4)
        BA_QR(3, 64, 4)
        BA_QR(4, 64, 16)
#undef BA_QR
    }

    // single shard: k_post_reduce's work rides on the two Schur launches (see k_schur_reduce)
    bool post_folded() const { return !sharded() && nred > 0 && !no_fold; }
    bool ctl_reduces() const { return !sharded() && !no_fold; }

    void launch_schur()
    {
        const T *lamf = post_folded() ? d_scal.p + SC_LAMBDA : nullptr;
        if (sx.nchunks > 0) {
            // persistent: schur_wgs workgroups per CU, every wavefront walks its own balanced list of chunks (see k_schur_pairs)
            const dim3 gp(schur_grid);
#define BA_PAIRS(SC, KO_, GB_) hipLaunchKernelGGL((k_schur_pairs<T, SC, KO_, GB_>), gp, dim3(256), 0, st, d_wave_ptr.p, schur_nband, d_chunk_info.p, d_ent.p, d_rec.p, \
                                                  (unsigned)(sizeof(T) * d_rec.n), d_tvec.p, Ml, d_slab.p, d_V.p, d_gc.p, D, ld, d_S.p)
            // SCALED: CHOLESKY is the only symbol whose point blocks carry a diagonal D (dinv != 1)
            static const int ko = getenv("BA_SCHUR_KNOCKOUT") ? atoi(getenv("BA_SCHUR_KNOCKOUT")) : 0; // (experiment: wrong results, see k_schur_pairs)
            if (ko == 1) BA_PAIRS(false, 1, 2);
            else if (ko == 2) BA_PAIRS(false, 2, 2);
            else if (ko == 3) BA_PAIRS(false, 3, 2);
            else if (schur_gb == 8) { if (kind == BA_CHOLESKY) BA_PAIRS(true, 0, 8); else BA_PAIRS(false, 0, 8); }
            else if (schur_gb == 4) { if (kind == BA_CHOLESKY) BA_PAIRS(true, 0, 4); else BA_PAIRS(false, 0, 4); }
            else if (kind == BA_CHOLESKY) BA_PAIRS(true, 0, 2); else BA_PAIRS(false, 0, 2);
#undef BA_PAIRS
        }
        const int post_blocks = lamf ? (Dp + 2) / 3 : 0;
        const long long nthr = (long long)nred * BA_SLAB;
        if (nred > 0)
            hipLaunchKernelGGL((k_schur_reduce<T>), dim3((unsigned)((nthr + 191) / 192) + post_blocks), dim3(192), 0, st, nred, d_red_pairs.p, D, ld,
                               d_pair_hi.p, d_pair_lo.p, d_pair_chunk_ptr.p, d_slab.p, d_V.p, d_gc.p, d_S.p, lamf, post_blocks, Dp, d_gcg.p, d_dxc.p);
    }

    // ---- distributed factor (BA_DIST_FACTOR=1, sharded LDL^T symbols; SURVEY 8e "consider distributing K6" -- FUNCTIONAL, unmeasured: this
    // pool gives one GPU).  1-D block-cyclic over the ranks: rank p % world owns block column p.
    //   exchange (round 4): every rank only needs the SUM of the block columns it owns, so the shards' partial
    //   systems meet in a REDUCE-SCATTER -- the trapezoid packed owner by owner into world chunks of equal length (k_pack_owner), half
    //   the bytes of the all-reduce per rank -- and the camera gradient + energy (D + 1 scalars, needed everywhere) in a small all-reduce;
    //   factor: per block column the owner factors it (k_ldlt_panel: L into S, Y = L D into Wp, the block's inverse into Winv), the
    //   three pieces travel in ONE broadcast from the owner (staged contiguously, k_panel_stage; round 3: three sum all-reduces of zeroed
    //   copies), and every rank applies the panel to the block columns IT owns (k_ldlt_update with the owner filter).
    // Behind the last column every rank holds the whole L and runs the back sweep redundantly.  The launch-per-step kernels: nobody
    // waits for anybody inside a launch; one collective per block column on the factor's critical path -- a design to measure once
    // there is a node, not a speed claim (at D = 2313 the factor is latency-bound and every hop is on its critical path).
    bool dist_factor = false;
    DevBuf<T> d_stage;       // one block column's [L | Y | W] for the broadcast
    DevBuf<int> d_own_off;   // per block column: offset of its first scalar in the owner-packed buffer, as a count of 64-scalar units (fits 32 bits)
    size_t own_chunk = 0;    // scalars per rank in the owner-packed buffer
    bool dist_on() const { return dist_factor && sharded() && !dense_qr(); }
    int dist_setup()
    {
        if (d_own_off.p) return BA_OK;
        const int nbc = Dp / NB;
        std::vector<size_t> fill((size_t)world, 0);
        std::vector<int> off((size_t)nbc);
        for (int p = 0; p < nbc; p++) { off[p] = (int)(fill[p % world] / 64); fill[p % world] += (size_t)64 * (size_t)(Dp - 64 * p); }
        own_chunk = 0;
        for (size_t f : fill) own_chunk = std::max(own_chunk, f);
        for (int p = 0; p < nbc; p++) off[p] += (int)((size_t)(p % world) * own_chunk / 64);
        int rc;
        if ((rc = d_own_off.upload(off))) return rc;
        if ((rc = d_pack.alloc((size_t)world * own_chunk + (size_t)D + 2))) return rc; // behind the chunks: g_c (D) + the energy
        return d_stage.alloc((size_t)(2 * (size_t)ld + NB) * NB);
    }
    size_t dist_small_off() const { return (size_t)world * own_chunk; }
    // segment A's tail / segment B's head in this mode
    int launch_dist_pack()
    {
        int rc;
        if ((rc = dist_setup())) return rc;
        HIPCHK(hipMemsetAsync(d_pack.p, 0, sizeof(T) * (size_t)world * own_chunk, st)); // (the padding behind the shorter chunks)
        const dim3 g((Dp + 255) / 256 > 8 ? 8 : (Dp + 255) / 256, Dp);
        hipLaunchKernelGGL((k_pack_owner<T, false>), g, dim3(256), 0, st, Dp, D, ld, d_S.p, d_pack.p, d_own_off.p, world, rank, dist_small_off(), d_scal.p,
                           (int)SC_ELOC, (int)SC_ENERGY);
        return BA_OK;
    }
    int launch_dist_unpack()
    {
        const dim3 g((Dp + 255) / 256 > 8 ? 8 : (Dp + 255) / 256, Dp);
        hipLaunchKernelGGL((k_pack_owner<T, true>), g, dim3(256), 0, st, Dp, D, ld, d_S.p, d_pack.p, d_own_off.p, world, rank, dist_small_off(), d_scal.p,
                           (int)SC_ELOC, (int)SC_ENERGY);
        return BA_OK;
    }
    int dist_exchange()
    {
        int rc;
        if ((rc = reduce_scatter(d_pack.p, own_chunk))) return rc;
        return allreduce(d_pack.p + dist_small_off(), (size_t)D + 1, 0);
    }
    int launch_factor_solve_dist()
    {
        const int nrows = D + 1, ncols = D, nblk = (ncols + NB - 1) / NB;
        int rc;
        for (int p = 0; p < nblk; p++) {
            const int p0 = p * NB, below = nrows - (p0 + NB), npanel = below > 0 ? (below + 63) / 64 : 1, owner = p % world;
            const int h = Dp - p0; // rows of the block column from its diagonal block down (the rhs row D among them)
            T *wcol = d_Wp.p, *winv = d_Winv.p + (size_t)p * NB * NB;
            const dim3 gs((h + 255) / 256, 2 * NB + 1);
            if (owner == rank) {
                hipLaunchKernelGGL((k_ldlt_panel<T, NB>), dim3(npanel), dim3(256), 0, st, nrows, ncols, ld, p0, d_S.p, wcol, winv, (int *)nullptr, 0);
                hipLaunchKernelGGL((k_panel_stage<T, NB, false>), gs, dim3(256), 0, st, h, p0, ld, d_S.p, wcol, winv, d_stage.p);
            }
            if ((rc = bcast(d_stage.p, (size_t)(2 * h + NB) * NB, owner))) return rc;
            if (owner != rank) hipLaunchKernelGGL((k_panel_stage<T, NB, true>), gs, dim3(256), 0, st, h, p0, ld, d_S.p, wcol, winv, d_stage.p);
            const int p1 = p0 + NB;
            if (p1 < ncols) {
                const int nti = (nrows - p1 + 63) / 64, ntj = (ncols - p1 + 63) / 64;
                hipLaunchKernelGGL((k_ldlt_update<T, NB>), dim3(ntj, nti), dim3(256), 0, st, nrows, ncols, ld, p0, d_S.p, (const T *)wcol, world, rank);
            }
        }
        ba_ldlt_backsweep<T, NB>(st, D, ld, D, d_S.p, d_Winv.p, d_dxc.p, d_dxc.p + Dp, /*armed by k_post_reduce*/ true, num_cus, d_scal.p + SC_ERR, /*safe*/ true);
        return BA_OK;
    }
    void launch_factor_solve() { launch_factor(); launch_backsweep(); }

    // QRKIT's right block (BAFunctor.h:101): J2bot built densely, Householder QR (ba_qr.hip.h), dx_c from R y = -Q^T qtb2
    void launch_qrkit_build()
    {
        if (more_qr()) { launch_more_build(); return; }
        (void)hipMemsetAsync(d_qA.p, 0, sizeof(T) * d_qA.n, st);
        const int nthr = std::max(Kl, D);
        hipLaunchKernelGGL((k_qrkit_build<T>), dim3((nthr + 255) / 256), dim3(256), 0, st, Kl, Ml, D, d_obs_cam.p, d_obs_pt.p, d_pt_ptr.p, d_Jc.p, d_r.p,
                           d_rec.p, d_q1obs.p, d_q1lam.p, d_tvec.p, d_scal.p + SC_LAMBDA, d_qA.p, q_lda, rank == 0 ? 1 : 0);
    }
    // MOREQR per trial (behind k_more_trial): the dense matrix the inner QR factors (k_more_build, k_more_tail)
    void launch_more_build()
    {
        (void)hipMemsetAsync(d_qA.p, 0, sizeof(T) * d_qA.n, st);
        if (Kl > 0)
            hipLaunchKernelGGL((k_more_build<T>), dim3(gK), dim3(256), 0, st, Kl, Ml, D, d_obs_cam.p, d_obs_pt.p, d_pt_ptr.p, d_rec0.p, d_rec.p, d_mQl.p, d_mQR.p,
                               d_tvec0.p, d_tvec.p, d_qA.p, q_lda);
        static const double dbg = getenv("BA_DBG_TAIL") ? (atof(getenv("BA_DBG_TAIL")) == -1.0 ? -1.0 : 1.0 + atof(getenv("BA_DBG_TAIL"))) : 1.0;
        if (rank == 0) hipLaunchKernelGGL((k_more_tail<T>), dim3(D + 1), dim3(256), 0, st, Ml, D, d_R22.p, d_scal.p + SC_LAMBDA, d_qA.p, q_lda, (T)dbg);
        static const bool dbg_atb = getenv("BA_DBG_ATB") != nullptr;
        if (dbg_atb) { // A^T b of the matrix as built, into the spare half of the step vector (getter 13)
            hipLaunchKernelGGL((k_dbg_atb<T>), dim3(D), dim3(256), 0, st, inner_rows(), D, (const T *)d_qA.p, q_lda, d_dbg.p);
        }
    }
    // MOREQR per outer iteration (behind k_elim_qr with lambda = 0; BacktrackLevMarqMore.h:288-291): the dense QR of J2bot(lambda = 0).
    // Part 1 (capturable): build + this shard's factorisation (+ sharded: its R | c2 into the zeroed stack).  Part 2: sharded -- the
    // all-reduce of the stack and its QR (distributed TSQR, like QRKIT's trial) --, then R22 | c2 out of the factored matrix.  Every
    // kernel is conditional on the step control (go): behind a rejected trial R22 stays what it was.
    int launch_more_outer(const int *go)
    {
        (void)hipMemsetAsync(d_qA.p, 0, sizeof(T) * d_qA.n, st);
        if (Kl > 0)
            hipLaunchKernelGGL((k_qrkit_build<T>), dim3((std::max(Kl, D) + 255) / 256), dim3(256), 0, st, Kl, Ml, D, d_obs_cam.p, d_obs_pt.p, d_pt_ptr.p, d_Jc.p, d_r.p,
                               d_rec0.p, d_q1obs.p, d_q1lam.p, d_tvec0.p, d_scal.p + SC_ZERO, d_qA.p, q_lda, 0, go);
        ba_qr_side sd = qr_side();
        sd.go = go;
        ba_qr_factor<T>(st, d_qA.p, q_lda, outer_rows(), D, d_qtau.p, q_tau_stride, sd);
        if (sharded()) {
            int rc;
            if (!d_qB.p && (rc = d_qB.alloc(qb_count()))) return rc;
            HIPCHK(hipMemsetAsync(d_qB.p, 0, sizeof(T) * qb_nmat(), st));
            hipLaunchKernelGGL((k_qr_stack_pack<T>), dim3(D + 2), dim3(256), 0, st, (const T *)d_qA.p, q_lda, D, rank, d_qB.p, qb_ld(), qb_nmat(), (const T *)d_gc.p,
                               (const T *)d_scal.p, (int)SC_ELOC);
        }
        return BA_OK;
    }
    int more_outer_finish(const int *go)
    {
        if (sharded()) {
            int rc;
            if ((rc = allreduce(d_qB.p, qb_count(), 0))) return rc;
            ba_qr_side sd = qr_side();
            sd.go = go;
            ba_qr_factor<T>(st, d_qB.p, qb_ld(), world * D, D, d_qtau.p, q_tau_stride, sd);
            hipLaunchKernelGGL((k_copy_r22<T>), dim3(D + 1), dim3(256), 0, st, D, (const T *)d_qB.p, qb_ld(), d_R22.p, go);
        } else
            hipLaunchKernelGGL((k_copy_r22<T>), dim3(D + 1), dim3(256), 0, st, D, (const T *)d_qA.p, q_lda, d_R22.p, go);
        return BA_OK;
    }
    DevBuf<T> d_qAcopy, d_dbgr, d_dbg2; // BA_DBG_QRCHECK
    void launch_qrkit_solve()
    {
        // BA_DBG_QRCHECK (diagnostic, step-level seam only): a copy of the matrix, and behind the solve its normal-equation residual (getter 14)
        bool chk = getenv("BA_DBG_QRCHECK") != nullptr;
        if (chk) {
            hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
            chk = hipStreamIsCapturing(st, &cs) == hipSuccess && cs == hipStreamCaptureStatusNone;
        }
        if (chk) {
            if (!d_qAcopy.p) { (void)d_qAcopy.alloc(d_qA.n); (void)d_dbgr.alloc(q_lda); (void)d_dbg2.alloc((size_t)2 * D); }
            (void)hipMemcpyAsync(d_qAcopy.p, d_qA.p, sizeof(T) * d_qA.n, hipMemcpyDeviceToDevice, st);
        }
        ba_qr_solve<T>(st, d_qA.p, q_lda, inner_rows(), D, d_qtau.p, q_tau_stride, d_dxc.p, qr_side());
        if (chk) {
            const int rows = inner_rows();
            hipLaunchKernelGGL((k_dbg_resid<T>), dim3((rows + 255) / 256), dim3(256), 0, st, rows, D, (const T *)d_qAcopy.p, q_lda, (const T *)d_dxc.p, d_dbgr.p);
            hipLaunchKernelGGL((k_dbg_atv<T>), dim3(D), dim3(256), 0, st, rows, D, (const T *)d_qAcopy.p, q_lda, (const T *)d_dbgr.p, d_dbg2.p);
            hipLaunchKernelGGL((k_dbg_atv<T>), dim3(D), dim3(256), 0, st, rows, D, (const T *)d_qAcopy.p, q_lda, (const T *)(d_qAcopy.p + (size_t)D * q_lda), d_dbg2.p + D);
        }
        (void)hipMemcpyAsync(d_gcg.p, d_gc.p, sizeof(T) * (size_t)D, hipMemcpyDeviceToDevice, st); // the camera gradient of the rho denominator
    }
    // Sharded QRKIT (distributed TSQR, ba_qr.hip.h): the QR of this shard's rows, its R + rhs head into the zeroed stack ...
    int launch_qr_stack_pack()
    {
        int rc;
        if (!d_qB.p && (rc = d_qB.alloc(qb_count()))) return rc;
        ba_qr_factor<T>(st, d_qA.p, q_lda, inner_rows(), D, d_qtau.p, q_tau_stride, qr_side());
        HIPCHK(hipMemsetAsync(d_qB.p, 0, sizeof(T) * qb_nmat(), st));
        hipLaunchKernelGGL((k_qr_stack_pack<T>), dim3(D + 2), dim3(256), 0, st, (const T *)d_qA.p, q_lda, D, rank, d_qB.p, qb_ld(), qb_nmat(), (const T *)d_gc.p,
                           (const T *)d_scal.p, (int)SC_ELOC);
        return BA_OK;
    }
    // ... and, behind the all-reduce, the QR of the stack and the camera step (redundantly on every shard)
    void launch_qr_stack_solve()
    {
        hipLaunchKernelGGL((k_qr_stack_unpack<T>), dim3((D + 255) / 256), dim3(256), 0, st, (const T *)d_qB.p, qb_nmat(), D, d_gcg.p, d_scal.p, (int)SC_ENERGY);
        ba_qr_solve<T>(st, d_qB.p, qb_ld(), world * D, D, d_qtau.p, q_tau_stride, d_dxc.p, qr_side());
    }
    // what the exchange step of a sharded trial sums: the packed reduced camera system, or (QRKIT / QRSPQR) the stack of R factors
    T *xchg_ptr() { return dense_qr() ? d_qB.p : d_pack.p; }
    size_t xchg_count() const { return dense_qr() ? qb_count() : pack_count() + 1; }

    void launch_factor()
    {
        ba_ldlt_factor<T, NB>(st, D + 1, D, ld, d_S.p, d_Wp.p, d_Winv.p, d_flags.p, (int)d_flags.n, d_scal.p + SC_ERR, safe_factor,
                              safe_factor ? 0 : fault_rowflag);
    }

    // backward sweep: one data-flow launch (k_ldlt_backflow) while its groups are certainly co-resident
    void launch_backsweep()
    {
        ba_ldlt_backsweep<T, NB>(st, D, ld, D, d_S.p, d_Winv.p, d_dxc.p, d_dxc.p + Dp, /*armed by k_post_reduce*/ true, num_cus, d_scal.p + SC_ERR,
                                 safe_factor);
    }

    void launch_post_reduce()
    {
        if (post_folded()) return;
        hipLaunchKernelGGL((k_post_reduce<T>), dim3((Dp + 3) / 4), dim3(256), 0, st, D, Dp, ld, d_scal.p + SC_LAMBDA, d_S.p, d_gcg.p, d_dxc.p);
    }

    int set_lambda(T lambda)
    {
        *h_lam = lambda;
        HIPCHK(hipMemcpyAsync(d_scal.p + SC_LAMBDA, h_lam, sizeof(T), hipMemcpyHostToDevice, st));
        // a lambda from the host: whatever records a fused linearisation left behind were for another one
        HIPCHK(hipMemsetAsync(&d_lm.p->rec_fresh, 0, sizeof(int), st));
        return BA_OK;
    }

    void launch_backsub_retract()
    {
        if (Ml > 0) { // the camera retraction rides as one more block at the end of the grid
            const ba_cam_retract_args cr{N, d_cam[0].p, d_dxc.p, d_gcg.p, d_cam[1].p, d_scal.p, (int)SC_RHO_C};
            hipLaunchKernelGGL((k_backsub<T, 8>), dim3(gB + 1), dim3(256), 0, st, Ml, d_pt_ptr.p, d_obs_cam.p, d_rec.p, d_dinv.p, d_tvec.p,
                               d_tri.p, d_dxc.p, d_gp.p, d_pts[0].p, d_scal.p + SC_LAMBDA, d_dxp.p, d_pts[1].p, d_part_bs.p, d_pperm.p, gB, cr);
        } else // (an empty shard keeps the zero partial sums written at creation)
            hipLaunchKernelGGL((k_retract_cams<T>), dim3(1), dim3(256), 0, st, N, d_cam[0].p, d_dxc.p, d_gcg.p, d_scal.p + SC_LAMBDA,
                               d_cam[1].p, d_scal.p, (int)SC_RHO_C);
    }

    ba_red_jobs test_energy_jobs() const
    {
        ba_red_jobs jobs{};
        jobs.j[0] = {d_part_e.p, gE, 0, SC_ETEST};
        jobs.j[1] = {d_part_bs.p, gB, 0, SC_RHO_P};
        jobs.j[2] = {d_part_bs.p + gB, gB, 0, SC_DN_P};
        return jobs;
    }
    // reduce: false when the control kernel behind this trial sums the partials itself (launch_seg_ctl, single shard)
    void launch_test_energy(bool reduce = true)
    {
        launch_eval(false, 1);
        if (reduce) hipLaunchKernelGGL((k_reduce_scalars<T>), dim3(3), dim3(256), 0, st, test_energy_jobs(), d_scal.p, (const int *)nullptr);
    }

    // sharded: the block-lower trapezoid of S (matrix + rhs row + g_c row: half the bytes of the full buffer) + one tail scalar
    size_t pack_count() const
    {
        const int nbc = Dp / NB;
        return (size_t)64 * ((size_t)nbc * Dp - (size_t)32 * nbc * (nbc - 1));
    }
    int launch_pack(bool unpack)
    {
        const size_t npk = pack_count();
        int rc;
        if (!d_pack.p && (rc = d_pack.alloc(npk + 1))) return rc;
        const dim3 gpk((Dp + 255) / 256 > 8 ? 8 : (Dp + 255) / 256, Dp);
        if (unpack) hipLaunchKernelGGL((k_pack_lower<T, true>), gpk, dim3(256), 0, st, Dp, ld, d_S.p, d_pack.p, npk, d_scal.p, (int)SC_ELOC, (int)SC_ENERGY);
        else hipLaunchKernelGGL((k_pack_lower<T, false>), gpk, dim3(256), 0, st, Dp, ld, d_S.p, d_pack.p, npk, d_scal.p, (int)SC_ELOC, (int)SC_ENERGY);
        return BA_OK;
    }

    // The segments of one trial (everything behind lambda in SC_LAMBDA).  Single shard: A + B back to back; sharded: an
    // all-reduce of the packed system between A and B and one of the three step scalars behind B.
    int launch_seg_a()
    {
        launch_eliminate();
        if (dense_qr()) { launch_qrkit_build(); return sharded() ? launch_qr_stack_pack() : BA_OK; }
        launch_schur();
        if (dist_on()) return launch_dist_pack();
        return sharded() ? launch_pack(false) : BA_OK;
    }
    int launch_seg_b()
    {
        int rc;
        if (dist_on()) { if ((rc = launch_dist_unpack())) return rc; }
        else if (sharded() && !dense_qr() && (rc = launch_pack(true))) return rc;
        if (dense_qr()) { if (sharded()) launch_qr_stack_solve(); else launch_qrkit_solve(); }
        else {
            launch_post_reduce();
            if (dist_factor && sharded()) { if ((rc = launch_factor_solve_dist())) return rc; }
            else launch_factor_solve();
        }
        launch_backsub_retract();
        launch_test_energy(!ctl_reduces()); // (single shard: k_lm_control sums the three partial arrays at its head)
        return BA_OK;
    }
    // step control on the device, x = xTest and the linearisation of the next outer iteration, the latter two conditional
    int launch_seg_ctl()
    {
        ba_lm_slots sl{SC_ENERGY, SC_ETEST, SC_RHO_P, SC_RHO_C, SC_DN_P, SC_DN_C, SC_LAMBDA, SC_ERR, SC_GUARD, world};
        hipLaunchKernelGGL((k_lm_control<T>), dim3(1), dim3(256 * BA_LM_JOBS), 0, st, d_scal.p, d_lm.p, d_log, sl, test_energy_jobs(), ctl_reduces() ? 3 : 0);
        int rc = linearize_enqueue(false, &d_lm.p->go); // (x = xTest happens inside its first kernel)
        if (!rc && more_qr() && !sharded()) rc = more_outer_finish(&d_lm.p->go); // (sharded: behind the segment, it holds an all-reduce -- enqueue_trial)
        return rc;
    }

    // m_solver.compute .. dx; xTest = x (+) dx; m_functor(xTest); rhoScale (BacktrackLevMarqQRChol.h:291-375): the step-level
    // seam, host-synchronous, with per-phase events (ba_minimize runs the same launches as graphs under device-side control).
    int try_step(double lambda_d, double *e_test, double *rho_scale, double *dx_norm) override
    {
        int rc;
        if ((rc = set_lambda((T)lambda_d))) return rc;
        HIPCHK(hipEventRecord(ev[EV_T0], st));
        launch_eliminate();
        HIPCHK(hipEventRecord(ev[EV_T1], st));
        if (dense_qr()) launch_qrkit_build(); else launch_schur();
        HIPCHK(hipEventRecord(ev[EV_T2], st));
        if (sharded()) {
            if (dense_qr()) { if ((rc = launch_qr_stack_pack()) || (rc = allreduce(d_qB.p, qb_count(), 0))) return rc; }
            else if (dist_on()) { if ((rc = launch_dist_pack()) || (rc = dist_exchange()) || (rc = launch_dist_unpack())) return rc; }
            else if ((rc = launch_pack(false)) || (rc = allreduce(d_pack.p, pack_count() + 1, 0)) || (rc = launch_pack(true))) return rc;
        }
        HIPCHK(hipEventRecord(ev[EV_T3], st));
        if (dense_qr()) { if (sharded()) launch_qr_stack_solve(); else launch_qrkit_solve(); }
        else {
            launch_post_reduce();
            if (keep) {
                if (!d_Skeep.p && (rc = d_Skeep.alloc(d_S.n))) return rc;
                HIPCHK(hipMemcpyAsync(d_Skeep.p, d_S.p, sizeof(T) * d_S.n, hipMemcpyDeviceToDevice, st));
            }
            if (dist_factor && sharded()) { if ((rc = launch_factor_solve_dist())) return rc; }
            else launch_factor_solve();
        }
        HIPCHK(hipEventRecord(ev[EV_T4], st));
        launch_backsub_retract();
        HIPCHK(hipEventRecord(ev[EV_T5], st));
        launch_test_energy();
        HIPCHK(hipEventRecord(ev[EV_T6], st));
        if ((rc = allreduce(d_scal.p + SC_ETEST, N_STEP_SCALARS, 0))) return rc; // (the error word too: a time-out of one shard fails the step on all)
        if ((rc = fetch_scalars())) return rc;
        HIPCHK(hipGetLastError());
        tm.eliminate_ms += ev_ms(EV_T0, EV_T1);
        tm.schur_ms += ev_ms(EV_T1, EV_T2);
        if (sharded()) tm.comm_ms += ev_ms(EV_T2, EV_T3);
        tm.factor_ms += ev_ms(EV_T3, EV_T4);
        tm.backsub_ms += ev_ms(EV_T4, EV_T5);
        tm.test_eval_ms += ev_ms(EV_T5, EV_T6);
        tm.trial_ms += ev_ms(EV_T0, EV_T6);
        tm.n_trials++;
        if (e_test) *e_test = (double)h_scal[SC_ETEST];
        if (rho_scale) *rho_scale = (double)(h_scal[SC_RHO_P] + h_scal[SC_RHO_C]);
        if (dx_norm) *dx_norm = std::sqrt((double)(h_scal[SC_DN_P] + h_scal[SC_DN_C]));
        have_step = true;
        return BA_OK;
    }

    // x = xTest (BacktrackLevMarqQRChol.h:428)
    int accept() override
    {
        if (!have_step) return BA_ERR_ARG;
        const int ncam = 15 * N, npts = 3 * Ml;
        hipLaunchKernelGGL((k_commit<T>), dim3((ncam + npts + 255) / 256), dim3(256), 0, st, ncam, npts, d_cam[1].p, d_pts[1].p, d_cam[0].p, d_pts[0].p,
                           (const int *)nullptr);
        HIPCHK(hipStreamSynchronize(st));
        have_step = false;
        return BA_OK;
    }

    int stats(double *out4) override
    {
        hipLaunchKernelGGL((k_stats<T>), dim3(gK), dim3(256), 0, st, Kl, N, Ml, d_cam[0].p, d_pts[0].p, d_obs_cam.p,
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
            if ((rc = dl(d_r.p, 2 * (size_t)Kl, h))) return rc;
            // file order inside the shard when the input was sorted; sorted order otherwise (perm documents it)
            for (int i = 0; i < Kl; i++) { out[2 * (size_t)i] = h[i]; out[2 * (size_t)i + 1] = h[(size_t)Kl + i]; }
            return BA_OK;
        }
        case BA_GET_JC: {
            if (n != 18 * (size_t)Kl) return BA_ERR_ARG;
            if (kind == BA_CHOLESKY) { // the AoS records [Kl][20] are the only copy
                if ((rc = dl(d_JcA.p, 20 * (size_t)Kl, h))) return rc;
                for (int i = 0; i < Kl; i++)
                    for (int q = 0; q < 18; q++) out[18 * (size_t)i + q] = h[20 * (size_t)i + q];
                return BA_OK;
            }
            if ((rc = dl(d_Jc.p, 18 * (size_t)Kl, h))) return rc;
            for (int i = 0; i < Kl; i++)
                for (int q = 0; q < 18; q++) out[18 * (size_t)i + q] = h[(size_t)q * Kl + i];
            return BA_OK;
        }
        case BA_GET_JP: {
            if (n != 6 * (size_t)Kl) return BA_ERR_ARG;
            if ((rc = dl(d_Jp.p, 6 * (size_t)Kl, h))) return rc;
            for (int i = 0; i < Kl; i++)
                for (int q = 0; q < 6; q++) out[6 * (size_t)i + q] = h[(size_t)q * Kl + i];
            return BA_OK;
        }
        case BA_GET_GRAD: {
            if (n != 3 * (size_t)Ml + D) return BA_ERR_ARG;
            if ((rc = dl(d_gp.p, 3 * (size_t)Ml, h)) || (rc = dl(d_gc.p, D, h2))) return rc;
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
        case 11: { // diagnostic (QRKIT / QRSPQR, single shard, behind a step): the diagonal of R of the dense QR of J2bot (= the betas of the top TSQR level)
            if (!dense_qr() || n != (size_t)D || !d_qA.p) return BA_ERR_ARG;
            HIPCHK(hipStreamSynchronize(st));
            std::vector<T> col(1);
            for (int c = 0; c < D; c++) {
                HIPCHK(hipMemcpy(col.data(), d_qA.p + (size_t)c * q_lda + c, sizeof(T), hipMemcpyDeviceToHost));
                out[c] = (double)col[0];
            }
            return BA_OK;
        }
        case 14: { // diagnostic (BA_DBG_QRCHECK): [A^T (b - A y) | A^T b] of the last dense least-squares solve, 2 D values
            if (!dense_qr() || n != (size_t)2 * D || !d_dbg2.p) return BA_ERR_ARG;
            std::vector<T> hh;
            int rcg;
            if ((rcg = dl(d_dbg2.p, (size_t)2 * D, hh))) return rcg;
            for (int c = 0; c < 2 * D; c++) out[c] = (double)hh[c];
            return BA_OK;
        }
        case 15:   // diagnostic (BA_DBG_QRCHECK): the matrix of the last dense solve as built, inner_rows() x (D + 1) column-major
        case 16: { // diagnostic: the same matrix as the factorisation left it (R, reflectors in place, Q^T rhs in column D)
            const size_t rows = (size_t)inner_rows();
            const T *src = what == 15 ? d_qAcopy.p : d_qA.p;
            if (!dense_qr() || n != rows * (D + 1) || !src) return BA_ERR_ARG;
            HIPCHK(hipStreamSynchronize(st));
            std::vector<T> col(rows);
            for (int c = 0; c <= D; c++) {
                HIPCHK(hipMemcpy(col.data(), src + (size_t)c * q_lda, sizeof(T) * rows, hipMemcpyDeviceToHost));
                for (size_t i = 0; i < rows; i++) out[(size_t)c * rows + i] = (double)col[i];
            }
            return BA_OK;
        }
        case 17: { // diagnostic: the T factors of the LAST panel factored (BA_QR_TAU_LEVELS level slots of q_tau_stride scalars)
            if (!dense_qr() || n != (size_t)BA_QR_TAU_LEVELS * q_tau_stride || !d_qtau.p) return BA_ERR_ARG;
            std::vector<T> hh;
            int rcg;
            if ((rcg = dl(d_qtau.p, n, hh))) return rcg;
            for (size_t c = 0; c < n; c++) out[c] = (double)hh[c];
            return BA_OK;
        }
        case 13: { // diagnostic (BA_DBG_ATB): A^T b of MOREQR's inner matrix as built
            if (!more_qr() || n != (size_t)D) return BA_ERR_ARG;
            std::vector<T> hh;
            int rcg;
            if ((rcg = dl(d_dbg.p, (size_t)D, hh))) return rcg;
            for (int c = 0; c < D; c++) out[c] = (double)hh[c];
            return BA_OK;
        }
        case 12: { // diagnostic: the first D rows of the factored matrix, all D + 1 columns (R | the head of Q^T rhs), column-major D x (D + 1)
            if (!dense_qr() || n != (size_t)D * (D + 1) || !d_qA.p) return BA_ERR_ARG;
            HIPCHK(hipStreamSynchronize(st));
            std::vector<T> col((size_t)D);
            for (int c = 0; c <= D; c++) {
                HIPCHK(hipMemcpy(col.data(), d_qA.p + (size_t)c * q_lda, sizeof(T) * (size_t)D, hipMemcpyDeviceToHost));
                for (int i = 0; i < D; i++) out[(size_t)c * D + i] = (c < D && i > c) ? 0.0 : (double)col[i];
            }
            return BA_OK;
        }
        case BA_GET_CAMS:
        case BA_GET_CAMS_TEST: {
            if (n != 15 * (size_t)N) return BA_ERR_ARG;
            if ((rc = dl(d_cam[what == BA_GET_CAMS ? 0 : 1].p, 15 * (size_t)N, h))) return rc;
            for (int a = 0; a < N; a++)
                for (int q = 0; q < 15; q++) out[15 * (size_t)a + q] = h[(size_t)q * N + a];
            return BA_OK;
        }
        case BA_GET_POINTS:
        case BA_GET_POINTS_TEST: {
            if (n != 3 * (size_t)Ml) return BA_ERR_ARG;
            if ((rc = dl(d_pts[what == BA_GET_POINTS ? 0 : 1].p, 3 * (size_t)Ml, h))) return rc;
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
            HIPCHK(hipMemcpy(d_cam[0].p, h.data(), sizeof(T) * h.size(), hipMemcpyHostToDevice));
        }
        if (pts && Ml > 0) {
            std::vector<T> h((size_t)3 * Ml);
            for (int j = 0; j < Ml; j++)
                for (int q = 0; q < 3; q++) h[(size_t)q * Ml + j] = (T)pts[3 * (size_t)j + q];
            HIPCHK(hipMemcpy(d_pts[0].p, h.data(), sizeof(T) * h.size(), hipMemcpyHostToDevice));
        }
        have_step = false;
        return BA_OK;
    }

    // ---- the LM loop: BacktrackLevMarqQRChol.h:204-436 == BacktrackLevMarqCholesky.h:190-361 (Scalar arithmetic in T) ------------
    // The first outer iteration is linearised host-synchronously (lambda0 needs max diag J'J); from then on the host only ENQUEUES:
    // every trial is the captured segments + k_lm_control, which decides accept / rho / lambda / flat-line on the device and gates
    // x = xTest and the next linearisation.  The host keeps at most LM_DEPTH trials in the stream, watches the progress counter
    // the control kernel bumps in pinned memory, and prints / forwards the table rows as they appear there.
    static constexpr int LM_DEPTH = 3;

    int capture(hipGraphExec_t *g, int (Solver::*seg)())
    {
        if (*g) return BA_OK;
        hipGraph_t gr = nullptr;
        HIPCHK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
        const int rc = (this->*seg)();
        const hipError_t e = hipStreamEndCapture(st, &gr); // (also on the error path: never leave the stream capturing)
        if (rc) { if (gr) (void)hipGraphDestroy(gr); return rc; }
        if (e != hipSuccess) { fprintf(stderr, "ba_mi355x: hipStreamEndCapture: %s\n", hipGetErrorString(e)); return BA_ERR_HIP; }
        const hipError_t ei = hipGraphInstantiate(g, gr, nullptr, nullptr, 0);
        (void)hipGraphDestroy(gr);
        if (ei != hipSuccess) { *g = nullptr; return BA_ERR_HIP; }
        return BA_OK;
    }
    int launch_seg_ab() { int rc = launch_seg_a(); return rc ? rc : launch_seg_b(); }
    int launch_seg_iter() { int rc = launch_seg_ab(); return rc ? rc : launch_seg_ctl(); }

    // one segment: replay its graph, or (legacy stream, BA_NO_GRAPH, sharding through a host callback) launch it directly
    int run_seg(hipGraphExec_t *g, int (Solver::*seg)(), bool graphs)
    {
        if (!graphs) return (this->*seg)();
        int rc = capture(g, seg);
        if (rc) return rc;
        HIPCHK(hipGraphLaunch(*g, st));
        return BA_OK;
    }

    int enqueue_trial(int slot, bool graphs)
    {
        int rc;
        EvSlot &e = ring[slot];
        if (spin_next_s > 0) { // self-test (3): a launch that outlasts the watchdog (it ends by itself)
            hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, st, (long long)(spin_next_s * wall_khz * 1e3));
            spin_next_s = 0;
        }
        if (!sharded()) // ONE graph per LM iteration and no event between the launches: the times come from the device's wall clock
            return run_seg(&g_trial, &Solver::launch_seg_iter, graphs);
        HIPCHK(hipEventRecord(e.e[0], st));
        {
            if ((rc = run_seg(&g_a, &Solver::launch_seg_a, graphs))) return rc;
            HIPCHK(hipEventRecord(e.e[1], st));
            // reduced camera system (or stack of R factors) + rhs + g_c + energy tail; the distributed factor only needs each rank's own
            // block columns summed: a reduce-scatter (half the bytes) + g_c and the energy in a small all-reduce
            if (dist_on()) { if ((rc = dist_exchange())) return rc; }
            else if ((rc = allreduce(xchg_ptr(), xchg_count(), 0))) return rc;
            HIPCHK(hipEventRecord(e.e[2], st));
            if ((rc = run_seg(&g_b, &Solver::launch_seg_b, graphs && !dist_on()))) return rc; // (the distributed factor's collectives sit INSIDE segment B)
            HIPCHK(hipEventRecord(e.e[3], st));
            // test energy, rho denominator, |dx|^2 (point parts) + the guard slot and the error word
            if ((rc = allreduce(d_scal.p + SC_ETEST, N_STEP_SCALARS, 0))) return rc;
        }
        HIPCHK(hipEventRecord(e.e[4], st));
        if ((rc = run_seg(&g_ctl, &Solver::launch_seg_ctl, graphs))) return rc;
        if (more_qr() && (rc = more_outer_finish(&d_lm.p->go))) return rc; // MOREQR's outer R22: the shards' R factors meet in one more all-reduce
        HIPCHK(hipEventRecord(e.e[5], st));
        return BA_OK;
    }

    void harvest(int slot, bool accepted)
    {
        EvSlot &e = ring[slot];
        tm.trial_ms += ev_ms(e.e[0], e.e[4]);
        if (sharded()) tm.comm_ms += ev_ms(e.e[1], e.e[2]) + ev_ms(e.e[3], e.e[4]);
        tm.n_trials++;
        tm.n_graph_trials++;
        if (accepted) { tm.linearize_ms += ev_ms(e.e[4], e.e[5]); tm.n_linearize++; }
    }

    int minimize(const ba_lm_params *lmp, ba_trial_cb cb, void *user, ba_result *out) override
    {
        if (poisoned) return BA_ERR_HIP;
        ba_lm_params lm;
        if (lmp) lm = *lmp; else ba_lm_params_default(&lm);
        const auto tbeg = std::chrono::steady_clock::now();
        const ba_timing tm0 = tm;
        const bool talk = lm.verbose && rank == 0;
        if (talk) {
            // outputHeader / outputIterHeader, BacktrackLevMarqQRChol.h:65-82
            printf("############################## Backtrack LevMarq ###############################\n");
            printf("--------------------------------------------------------------------------------\n");
            printf(" Iter%15s%15s%15s%15s%15s\n", "Status", "f", "rho", "lambda", "Elapsed");
            printf("--------------------------------------------------------------------------------\n");
        }
        int rc = BA_OK;
        ba_lm_dev<T> h{};
        h.lam_min = (T)lm.lambda_min; h.lam_max = (T)lm.lambda_max; h.tol_fun = (T)lm.tol_fun; h.inc_base = (T)lm.lambda_increase_base;
        h.lambda = (T)lm.lambda_init; h.lambda_inc = h.inc_base;
        h.max_iter = lm.max_iter; h.max_fun_ev = lm.max_fun_ev; h.max_trials = lm.max_trials;
        h.status = BA_RUNNING;
        h.iter = 1;
        int launched = 0, consumed = 0, harvested = 0;
        std::vector<char> acc_of(RING_EV, 0);
        if (h.iter > lm.max_iter) { h.status = BA_MAX_ITERS; h.stop = 1; }
        else if (lm.max_trials < 0) { h.stop = 1; }
        else {
            // outer iteration 1: m_functor(x, r), df, JtRes, column norms; lambda0 = 1e-12 max diag J'J (:257-280; Cholesky.h:263-265);
            // MOREQR: 1e-6 * max column norm (BacktrackLevMarqMore.h:272-284)
            double e = 0, dmax = 0;
            if ((rc = linearize(&e, &dmax))) return rc;
            h.energy = (T)e;
            h.fun_evals = 1;
            h.lambda = kind == BA_MOREQR ? (T)(1e-6 * std::sqrt(dmax)) : (T)(1e-12 * dmax);
        }
        if (!h.stop && dist_on() && (rc = dist_setup())) return rc; // (never inside a stream capture)
        if (!h.stop && sharded() && !dense_qr() && !d_pack.p && (rc = d_pack.alloc(pack_count() + 1))) return rc;
        if (!h.stop && sharded() && dense_qr() && !d_qB.p && (rc = d_qB.alloc(qb_count()))) return rc;
        if (!h.stop) {
            h_log->done = 0; h_log->stop = 0; h_log->status = BA_RUNNING;
            if ((rc = set_lambda(h.lambda))) return rc;
            static_assert(SC_ERR == SC_GUARD + 1, "cleared together");
            HIPCHK(hipMemsetAsync(d_scal.p + SC_GUARD, 0, 2 * sizeof(T), st)); // the guard slot of trial 0 (no decision yet), no error
            HIPCHK(hipMemcpyAsync(d_lm.p, &h, sizeof h, hipMemcpyHostToDevice, st));
            HIPCHK(hipStreamSynchronize(st)); // (h is on the stack)
            // graphs unless the stream cannot be captured (legacy stream) or a host callback sits between the segments anyway
            const bool graphs = use_graph && st != nullptr; // (the distributed factor keeps segments A and C as graphs; B holds its collectives)
            auto tlast = std::chrono::steady_clock::now();
            auto drain = [&]() { // table rows that have appeared since the last look
                const int done = __atomic_load_n(&h_log->done, __ATOMIC_ACQUIRE);
                while (consumed < done) {
                    const ba_lm_row r = h_log->rows[consumed % BA_LM_RING];
                    const auto tnow = std::chrono::steady_clock::now();
                    const double el = std::chrono::duration<double>(tnow - tlast).count();
                    tlast = tnow;
                    acc_of[consumed % RING_EV] = r.accepted != 0;
                    consumed++;
                    if (r.stop == 2.0) continue; // ended by a device error: not a row of the table (the trial is repeated or the run fails)
                    if (!sharded()) { // device wall-clock stamps (k_lm_control): no events sit between the launches of an iteration
                        tm.n_graph_trials++;
                        if (r.trial_ticks >= 0) { tm.trial_ms += r.trial_ticks / wall_khz; tm.n_trials++; } // (not the first: host time sits in front of it)
                        if (r.ctl_ticks_prev >= 0 && r.prev_go != 0) { tm.linearize_ms += r.ctl_ticks_prev / wall_khz; tm.n_linearize++; }
                    }
                    if (cb) cb(user, (int)r.iter, (int)r.accepted, r.f, r.rho, r.lambda, el);
                    if (talk) // outputIter, BacktrackLevMarqQRChol.h:84-93 (f is the energy BEFORE the step)
                        printf("%5d%15s%15g%15g%15g%14gs\n", (int)r.iter, r.accepted != 0 ? "Accepted" : "Rejected", r.f, r.rho, r.lambda, el);
                }
                while (harvested + 1 < consumed) { if (sharded()) harvest(harvested % RING_EV, acc_of[harvested % RING_EV]); harvested++; } // (its events are complete)
                return done;
            };
            for (int attempt = 0;; attempt++) {
                // How many trials are enqueued depends on DEVICE DATA ONLY: trial n goes into the stream iff n < LM_DEPTH or row
                // n - LM_DEPTH has arrived with its stop mark clear (and n < max_trials).  The row that ends the run is row s on
                // every shard (same control kernel, same all-reduced scalars), so every rank enqueues exactly s + LM_DEPTH trials
                // and with them the same number of collectives -- whatever its host's timing.  (Looking at a global "stopped"
                // word instead made that number depend on WHEN a host looked: a rank one trial ahead then sat in ncclAllReduce
                // for ever.)  Trials behind row s run as no-ops (k_lm_control returns at once, nothing is accepted).
                auto tprog = std::chrono::steady_clock::now(); // last time the device made progress (a trial finished)
                int seen = consumed;
                while (true) {
                    const int done = drain();
                    if (done != seen) { seen = done; tprog = std::chrono::steady_clock::now(); }
                    bool may = !(lm.max_trials > 0 && launched >= lm.max_trials), final = !may;
                    if (may && launched >= LM_DEPTH) {
                        const int need = launched - LM_DEPTH;
                        if (done <= need) may = false;                                                       // (not there yet)
                        else if (h_log->rows[need % BA_LM_RING].stop != 0.0) { may = false; final = true; }   // the run ended at row `need`
                    }
                    if (may) {
                        if ((rc = enqueue_trial(launched % RING_EV, graphs))) break;
                        launched++;
                        continue;
                    }
                    if (final) break; // nothing more will ever be enqueued: wait for what is in the stream
                    // LM_DEPTH trials are in the stream: wait for a row (a trial is 0.1 ... 20 ms; watchdog_s without one = a hung launch)
                    if (std::chrono::duration<double>(std::chrono::steady_clock::now() - tprog).count() > watchdog_s) {
                        // Nothing can be waited for and nothing freed: the stream holds the launch that never ends (a synchronise or
                        // a hipFree would block for ever).  The handle is dead from here on; a retry belongs in a fresh process.
                        fprintf(stderr, "ba_mi355x: no LM trial completed within %g s -- giving up on this solver (nothing is freed)\n", watchdog_s);
                        poisoned = true;
                        return BA_ERR_HIP;
                    }
                    __builtin_ia32_pause();
                }
                if (!rc) { // wait for the stream with the same watchdog (a trial enqueued behind the last row may be the one that hangs)
                    const auto tw = std::chrono::steady_clock::now();
                    hipError_t eq;
                    while ((eq = hipStreamQuery(st)) == hipErrorNotReady) {
                        (void)drain(); // rows still arrive (up to LM_DEPTH of them): deliver each when it lands, with its own elapsed time
                        if (std::chrono::duration<double>(std::chrono::steady_clock::now() - tw).count() > watchdog_s) {
                            fprintf(stderr, "ba_mi355x: the stream did not drain within %g s -- giving up on this solver (nothing is freed)\n", watchdog_s);
                            poisoned = true;
                            return BA_ERR_HIP;
                        }
                        __builtin_ia32_pause();
                    }
                    if (eq != hipSuccess) { fprintf(stderr, "ba_mi355x: %s\n", hipGetErrorString(eq)); return BA_ERR_HIP; }
                } else {
                    const hipError_t es = hipStreamSynchronize(st);
                    if (es != hipSuccess) { fprintf(stderr, "ba_mi355x: %s\n", hipGetErrorString(es)); return BA_ERR_HIP; }
                }
                drain();
                while (harvested < consumed) { if (sharded()) harvest(harvested % RING_EV, acc_of[harvested % RING_EV]); harvested++; }
                if (rc) return rc;
                HIPCHK(hipMemcpy(&h, d_lm.p, sizeof h, hipMemcpyDeviceToHost));
                if (h.status != BA_DEV_FAILED) break;
                report_device_error(h.deverr);
                if (h.deverr == BA_DEVERR_DIVERGED) return BA_ERR_COMM;
                if (safe_factor || attempt > 0) return BA_ERR_HIP; // (the launch-per-step path has no hand-offs: cannot happen)
                // ---- a hand-off inside a fused launch timed out (a row workgroup not resident in time: a GPU shared with another
                // process, say).  x, lambda and the linearisation are untouched (nothing was accepted); repeat the trial -- and run
                // the rest -- with one launch per step.  Sharded: the error word rode on the scalar all-reduce, every rank is here.
                fprintf(stderr, "ba_mi355x: repeating LM trial %d with the launch-per-step factorisation\n", h.trials - 1);
                safe_factor = true;
                recoveries++;
                for (hipGraphExec_t *g : {&g_trial, &g_b})
                    if (*g) { (void)hipGraphExecDestroy(*g); *g = nullptr; }
                h.trials -= 1; h.fun_evals -= 1; // the failed trial's row does not count
                h.status = BA_RUNNING; h.stop = 0; h.deverr = 0; h.go = 0; h.timed = 0; h.prev_go = 0;
                // (h.prev_code: k_lm_control overwrote it with the FAILED trial's code -- rejected + stop = 2 -- on every shard alike; the
                // guard slot gets that same value back, so the repeated trial's check "sum == world x mine" holds on both sides)
                {
                    const T code = (T)h.prev_code;
                    *h_lam = code; // (pinned staging word; lambda itself lives on the device)
                    HIPCHK(hipMemcpyAsync(d_scal.p + SC_GUARD, h_lam, sizeof(T), hipMemcpyHostToDevice, st));
                    // (the trials that were in the stream behind the failed one ran the same fused launches and raised the word again)
                    HIPCHK(hipMemsetAsync(d_scal.p + SC_ERR, 0, sizeof(T), st));
                }
                HIPCHK(hipMemcpyAsync(d_lm.p, &h, sizeof h, hipMemcpyHostToDevice, st));
                HIPCHK(hipStreamSynchronize(st));
                launched = consumed = harvested = h.trials;
                if (lm.max_trials > 0) { /* rows already delivered count: the device compares s.trials with max_trials itself */ }
                __atomic_store_n(&h_log->done, h.trials, __ATOMIC_RELEASE);
                h_log->stop = 0; h_log->status = BA_RUNNING;
            }
            if (h.fresh) { // stopped (max_trials) right behind an accepted step: the energy of the linearisation that followed it
                if ((rc = fetch_scalars())) return rc;
                if (sharded()) { // (only this shard's part is there until a trial carries it through the all-reduce)
                    HIPCHK(hipMemcpyAsync(d_scal.p + SC_ENERGY, d_scal.p + SC_ELOC, sizeof(T), hipMemcpyDeviceToDevice, st));
                    if ((rc = allreduce(d_scal.p + SC_ENERGY, 1, 0)) || (rc = fetch_scalars())) return rc;
                }
                h.energy = h_scal[SC_ENERGY];
            }
            HIPCHK(hipGetLastError());
        }
        if (talk) printf("--------------------------------------------------------------------------------\n");
        have_step = false;
        if (out) {
            out->status = h.status; out->iterations = h.iter; out->trials = h.trials; out->fun_evals = h.fun_evals;
            out->energy = (double)h.energy; out->lambda = (double)h.lambda;
            out->seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - tbeg).count();
            const long long nt = tm.n_trials - tm0.n_trials, nl = tm.n_linearize - tm0.n_linearize; // (the timed ones)
            out->schur_ms = nt ? (tm.trial_ms - tm0.trial_ms) / nt : 0.0; // whole trial incl. the test-energy evaluation and the all-reduces
            out->linearize_ms = nl ? (tm.linearize_ms - tm0.linearize_ms) / nl : 0.0;
        }
        return rc;
    }

    // which = 1: the backward sweep with the group at the head of its chain missing (and a short spin bound): the groups behind
    // it wait for unknowns that are never published -- the production path's reaction to that must be BA_ERR_HIP.
    int selftest(int which) override
    {
        if (which == 2) { // arm: the fused factorisation's row workgroups stay silent and the panel's wait is short -- the next
                          // ba_minimize meets BA_DEVERR_ROW_FLAG on its first trial and must recover through the launch-per-step path
            if ((D + NB - 1) / NB < 2 || safe_factor) return BA_ERR_ARG; // (a single block column has no fused step)
            fault_rowflag = 1;
            for (hipGraphExec_t *g : {&g_trial, &g_b})
                if (*g) { (void)hipGraphExecDestroy(*g); *g = nullptr; }
            return BA_OK;
        }
        if (which == 3) { // arm: a 4-second kernel in front of the next trial -- with BA_WATCHDOG_S below that, ba_minimize must give up
            spin_next_s = 4.0;
            return BA_OK;
        }
        if (which != 1) return BA_ERR_ARG;
        const int nblk = (D + NB - 1) / NB, groups = (nblk + 1) / 2;
        if (groups < 2 || 2 * groups > num_cus) return BA_ERR_ARG;
        hipLaunchKernelGGL((k_fill_sentinel<T>), dim3((2 * Dp + 2 * NB + 255) / 256), dim3(256), 0, st, 2 * Dp + 2 * NB, d_dxc.p);
        hipLaunchKernelGGL((k_ldlt_backflow<T, NB>), dim3(2 * groups), dim3(256), 0, st, D, ld, D, nblk, d_S.p, d_Winv.p, d_dxc.p, d_dxc.p + Dp,
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
            case 0: launch_eval(false, 0); break;
            case 1: launch_eval(true, 0); launch_grad(); break;
            case 2: launch_eliminate(); break;
            case 3: if (dense_qr()) launch_qrkit_build(); else launch_schur(); break; // (QRKIT: J2bot instead of S)
            case 4:
                if (dense_qr()) { launch_qrkit_build(); launch_qrkit_solve(); break; }
                launch_schur(); // the factorisation is in place: rebuild S first (timed separately by phase 3)
                launch_post_reduce();
                launch_factor_solve();
                break;
            case 5: launch_backsub_retract(); break;
            case 8: // the linearisation as ba_minimize runs it behind an accepted step: fused point part (+ CHOLESKY: the next trial's records)
                launch_eval(true, 0, nullptr, false, true);
                launch_grad(nullptr, nullptr, !fuse);
                break;
            case 6: // dense factorisation only (k_ldlt_panel + k_ldlt_step / k_ldlt_update; QRKIT: the Householder QR + solve): events around it, per rep
            case 7: // backward sweep only (k_ldlt_backflow; QRKIT: nothing, the solve is part of 6)
                if (dense_qr()) {
                    launch_qrkit_build();
                    HIPCHK(hipEventRecord(ev[EV_T2], st));
                    if (phase == 6) launch_qrkit_solve();
                    HIPCHK(hipEventRecord(ev[EV_T3], st));
                    HIPCHK(hipStreamSynchronize(st));
                    acc_ms += ev_ms(EV_T2, EV_T3);
                    break;
                }
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
        *ms = ((phase == 6 || phase == 7) ? acc_ms : ev_ms(EV_T0, EV_T1)) / reps;
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
    if (kind != BA_QRKIT && kind != BA_QRCHOL && kind != BA_CHOLESKY && kind != BA_MOREQR && kind != BA_QRSPQR) return BA_ERR_ARG;
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
    // a solver the watchdog gave up on still has a launch in its stream that may never end: every HIP call that frees or
    // synchronises could block with it -- its device memory is left to the end of the process
    if (!s->impl->poisoned) delete s->impl;
    delete s;
}

int ba_solver_recoveries(const ba_solver *s) { return s ? s->impl->recoveries : -1; }

int ba_solver_set_allreduce(ba_solver *s, ba_allreduce_fn fn, void *user)
{
    if (!s) return BA_ERR_ARG;
    s->impl->ar_fn = fn; s->impl->ar_user = user;
    return BA_OK;
}

int ba_solver_comm_init(ba_solver *s, const void *id)
{
    if (!s || !id) return BA_ERR_ARG;
    if (s->impl->comm) return BA_ERR_ARG;
    return ba_rccl_init(&s->impl->comm, id, s->impl->rank, s->impl->world);
}

int ba_solver_set_stream(ba_solver *s, void *hip_stream)
{
    if (!s) return BA_ERR_ARG;
    if (s->impl->poisoned) return BA_ERR_HIP;
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
#define BA_LIVE(s) ((s) && !(s)->impl->poisoned)
int ba_solver_linearize(ba_solver *s, double *energy, double *diag_max) { return !s ? BA_ERR_ARG : !BA_LIVE(s) ? BA_ERR_HIP : s->impl->linearize(energy, diag_max); }
int ba_solver_try_step(ba_solver *s, double lambda, double *energy_test, double *rho_scale, double *dx_norm)
{
    return !s ? BA_ERR_ARG : !BA_LIVE(s) ? BA_ERR_HIP : s->impl->try_step(lambda, energy_test, rho_scale, dx_norm);
}
int ba_solver_accept(ba_solver *s) { return !s ? BA_ERR_ARG : !BA_LIVE(s) ? BA_ERR_HIP : s->impl->accept(); }
int ba_solver_stats(ba_solver *s, double *out4) { return !(s && out4) ? BA_ERR_ARG : !BA_LIVE(s) ? BA_ERR_HIP : s->impl->stats(out4); }
int ba_solver_get(ba_solver *s, int what, double *out, size_t n) { return !(s && out) ? BA_ERR_ARG : !BA_LIVE(s) ? BA_ERR_HIP : s->impl->get(what, out, n); }
int ba_solver_set_state(ba_solver *s, const double *cam15, const double *pts) { return !s ? BA_ERR_ARG : !BA_LIVE(s) ? BA_ERR_HIP : s->impl->set_state(cam15, pts); }

int ba_solver_timing(ba_solver *s, ba_timing *out, int reset)
{
    if (!s) return BA_ERR_ARG;
    if (out) *out = s->impl->tm;
    if (reset) s->impl->tm = ba_timing{};
    return BA_OK;
}

int ba_solver_time_phase(ba_solver *s, int phase, int reps, double lambda, double *ms_per_launch)
{
    return !s ? BA_ERR_ARG : !BA_LIVE(s) ? BA_ERR_HIP : s->impl->time_phase(phase, reps, lambda, ms_per_launch);
}

int ba_solver_selftest(ba_solver *s, int which) { return !s ? BA_ERR_ARG : !BA_LIVE(s) ? BA_ERR_HIP : s->impl->selftest(which); }

} // extern "C"
