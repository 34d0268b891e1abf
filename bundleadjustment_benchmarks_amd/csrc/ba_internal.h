// ba_internal.h -- host-side types shared by the translation units of libba_mi355x.so (not part of the ABI).
#ifndef BA_INTERNAL_H
#define BA_INTERNAL_H

#include "../../include/ba_mi355x.h"

#include <cstdint>
#include <vector>

// BAL problem as parsed (src/bundle_adjustment_large.cpp:59-107): observation order of the file.
struct ba_problem {
    int N = 0, M = 0, K = 0;
    std::vector<int> cam_idx, pt_idx;
    std::vector<double> meas;  // 2K interleaved (u,v)
    std::vector<double> cams9; // 9N: omega(3), T(3), f, k1, k2
    std::vector<double> pts;   // 3M
};

// Static structure of one shard, built once on the host (ba_structure.cpp) and copied to HBM.
//
// Observations are stably sorted by point (the reference's row-interleave of [J; sqrt(lambda) I] needs that order,
// src/Eigen_ext/BacktrackLevMarqQRChol.h:291-309); a shard owns a contiguous point range and its observations.
// The reduced camera matrix is assembled per camera pair (hi >= lo): every point seen by both cameras contributes
// one ENTRY (row observation, column observation); entries are grouped by pair and cut into CHUNKS of at most
// `chunk_len` entries so that the work is spread evenly over wavefronts and summed in a fixed order.
struct ba_structure {
    int N = 0;              // cameras (global)
    int M = 0, K = 0;       // global points / observations
    int p0 = 0, p1 = 0;     // owned point range
    int o0 = 0, o1 = 0;     // owned observation range (in sorted order)
    int Ml = 0, Kl = 0;     // local counts
    bool was_sorted = true; // input already sorted by point
    std::vector<int> perm;  // sorted position -> file position (size K, global)
    std::vector<int> obs_cam, obs_pt; // local (obs_pt is the LOCAL point index), size Kl
    std::vector<int> pt_ptr;          // Ml + 1
    int kmax = 0;                     // max observations per point in this shard
    // points bucketed by track length for the per-point QR (lanes per point x observations per lane):
    // <= 32 (8 x 4), <= 64 (16 x 4), <= 128 (32 x 4), <= 256 (64 x 4), <= 1024 (64 x 16)
    std::vector<int> qr_pts;          // Ml local point indices, bucket by bucket (file order inside a bucket)
    int qr_bucket_ptr[6] = {0, 0, 0, 0, 0, 0};
    // camera pairs
    int npairs = 0;                   // N (N + 1) / 2, pair id = hi (hi + 1) / 2 + lo
    std::vector<int> pair_hi, pair_lo;
    long long E = 0;                  // entries
    std::vector<int> ent_r, ent_c;    // row / column observation (local index) per entry, grouped by pair
    int chunk_len = 64;
    int nchunks = 0;
    std::vector<int> chunk_ptr;       // nchunks + 1 offsets into entries
    std::vector<int> chunk_pair;      // pair id per chunk
    std::vector<int> pair_chunk_ptr;  // npairs + 1 offsets into chunks
    // diagonal pairs only = observations of one camera (camera-sorted view), used for J_c^T J_c and g_c
    int ndchunks = 0;
    std::vector<int> dchunk_ptr;      // ndchunks + 1 offsets into cam_obs
    std::vector<int> dchunk_cam;      // camera per diagonal chunk
    std::vector<int> cam_dchunk_ptr;  // N + 1
    std::vector<int> cam_obs;         // Kl observations grouped by camera
};

// chunk_len: entries per chunk of a camera pair (one wavefront of k_schur_pairs); dchunk_len: observations per chunk of a camera (k_cam_gram)
int ba_build_structure(const ba_problem *p, int shard_rank, int shard_world, int chunk_len, int dchunk_len, ba_structure *out);

// RCCL transport (ba_comm.cpp); comm is an ncclComm_t
int ba_rccl_init(void **comm_out, const void *id128, int rank, int world);
void ba_rccl_destroy(void *comm);
int ba_rccl_allreduce(void *comm, void *buf, size_t count, int f64, int op, void *stream);
int ba_rccl_broadcast(void *comm, void *buf, size_t count, int f64, int root, void *stream);
int ba_rccl_reduce_scatter(void *comm, void *buf, size_t count_per_rank, int f64, int rank, void *stream);

#endif
