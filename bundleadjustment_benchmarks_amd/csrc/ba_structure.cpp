// ba_structure.cpp -- static (iteration-independent) structure of one shard: point-sorted observation order,
// shard boundaries, camera-pair entry lists and their chunking.  Host code, runs once per solver.
//
// What it replaces in the reference: the sparsity bookkeeping Eigen does on every call -- setFromTriplets of 24K
// triplets per outer iteration (src/Optimization/BAFunctor.cpp:95-98), the per-trial row permutation of
// [J; sqrt(lambda) I] (src/Eigen_ext/BacktrackLevMarqQRChol.h:291-315) and the symbolic analysis inside
// SimplicialLDLT::compute (src/Eigen_ext/BacktrackLevMarqCholesky.h:278).  The pattern never changes, so it is
// built once.
#include "ba_internal.h"

#include <algorithm>
#include <numeric>

int ba_build_structure(const ba_problem *p, int shard_rank, int shard_world, int chunk_len, int dchunk_len, ba_structure *s)
{
    if (!p || !s || shard_world < 1 || shard_rank < 0 || shard_rank >= shard_world || chunk_len < 1 || dchunk_len < 1) return BA_ERR_ARG;
    const int N = p->N, M = p->M, K = p->K;
    s->N = N; s->M = M; s->K = K; s->chunk_len = chunk_len;

    // 1. stable sort of the observations by point (counting sort); BAL files are already sorted.
    s->was_sorted = true;
    for (int i = 1; i < K; i++)
        if (p->pt_idx[i] < p->pt_idx[i - 1]) { s->was_sorted = false; break; }
    std::vector<int> gptr((size_t)M + 1, 0);
    for (int i = 0; i < K; i++) gptr[(size_t)p->pt_idx[i] + 1]++;
    for (int j = 0; j < M; j++) gptr[j + 1] += gptr[j];
    s->perm.resize(K);
    if (s->was_sorted) {
        std::iota(s->perm.begin(), s->perm.end(), 0);
    } else {
        std::vector<int> cur(gptr.begin(), gptr.end() - 1);
        for (int i = 0; i < K; i++) s->perm[cur[p->pt_idx[i]]++] = i;
    }

    // 2. shard boundaries: contiguous point ranges balanced by observation count.
    auto boundary = [&](int r) -> int {
        if (r <= 0) return 0;
        if (r >= shard_world) return M;
        const long long target = (long long)K * r / shard_world;
        int j = (int)(std::lower_bound(gptr.begin(), gptr.end(), (int)target) - gptr.begin());
        return std::min(j, M);
    };
    s->p0 = boundary(shard_rank);
    s->p1 = boundary(shard_rank + 1);
    s->o0 = gptr[s->p0];
    s->o1 = gptr[s->p1];
    s->Ml = s->p1 - s->p0;
    s->Kl = s->o1 - s->o0;
    const int Ml = s->Ml, Kl = s->Kl;
    s->obs_cam.resize(Kl);
    s->obs_pt.resize(Kl);
    s->pt_ptr.resize((size_t)Ml + 1);
    for (int j = 0; j <= Ml; j++) s->pt_ptr[j] = gptr[s->p0 + j] - s->o0;
    s->kmax = 0;
    for (int j = 0; j < Ml; j++) s->kmax = std::max(s->kmax, s->pt_ptr[j + 1] - s->pt_ptr[j]);
    {
        auto bucket = [](int k) { return k <= 32 ? 0 : k <= 64 ? 1 : k <= 128 ? 2 : k <= 256 ? 3 : 4; };
        int cnt[5] = {0, 0, 0, 0, 0};
        for (int j = 0; j < Ml; j++) cnt[bucket(s->pt_ptr[j + 1] - s->pt_ptr[j])]++;
        s->qr_bucket_ptr[0] = 0;
        for (int b = 0; b < 5; b++) s->qr_bucket_ptr[b + 1] = s->qr_bucket_ptr[b] + cnt[b];
        int cur[5];
        for (int b = 0; b < 5; b++) cur[b] = s->qr_bucket_ptr[b];
        s->qr_pts.resize(Ml);
        for (int j = 0; j < Ml; j++) s->qr_pts[cur[bucket(s->pt_ptr[j + 1] - s->pt_ptr[j])]++] = j;
    }
    for (int i = 0; i < Kl; i++) {
        const int src = s->perm[s->o0 + i];
        s->obs_cam[i] = p->cam_idx[src];
        s->obs_pt[i] = p->pt_idx[src] - s->p0;
    }

    // 3. camera pairs (hi >= lo), pair id = hi (hi + 1) / 2 + lo.
    const long long np = (long long)N * (N + 1) / 2;
    if (np > 0x7fffffffLL) return BA_ERR_ARG;
    s->npairs = (int)np;
    s->pair_hi.resize(np);
    s->pair_lo.resize(np);
    for (int hi = 0, q = 0; hi < N; hi++)
        for (int lo = 0; lo <= hi; lo++, q++) { s->pair_hi[q] = hi; s->pair_lo[q] = lo; }
    auto pid = [](int a, int b) -> long long {
        const int hi = std::max(a, b), lo = std::min(a, b);
        return (long long)hi * (hi + 1) / 2 + lo;
    };
    // count entries per pair.  For one point with observations i < i': entry (row = the one with the larger camera).
    // Two observations of the SAME camera in one point (not in BAL data, allowed) give both orders on the diagonal pair.
    std::vector<long long> pcount((size_t)np + 1, 0);
    for (int j = 0; j < Ml; j++) {
        const int b = s->pt_ptr[j], e = s->pt_ptr[j + 1];
        for (int i = b; i < e; i++)
            for (int i2 = b; i2 <= i; i2++) {
                const long long q = pid(s->obs_cam[i], s->obs_cam[i2]);
                pcount[q + 1] += (i != i2 && s->obs_cam[i] == s->obs_cam[i2]) ? 2 : 1;
            }
    }
    for (long long q = 0; q < np; q++) pcount[q + 1] += pcount[q];
    s->E = pcount[np];
    if (s->E > 0x7fffffffLL) return BA_ERR_NOMEM;
    s->ent_r.resize((size_t)s->E);
    s->ent_c.resize((size_t)s->E);
    {
        std::vector<long long> cur(pcount.begin(), pcount.end() - 1);
        for (int j = 0; j < Ml; j++) { // increasing point order inside each pair: fixed summation order
            const int b = s->pt_ptr[j], e = s->pt_ptr[j + 1];
            for (int i = b; i < e; i++)
                for (int i2 = b; i2 <= i; i2++) {
                    const int ca = s->obs_cam[i], cb = s->obs_cam[i2];
                    const long long q = pid(ca, cb);
                    if (ca >= cb) { s->ent_r[cur[q]] = i; s->ent_c[cur[q]] = i2; cur[q]++; }
                    else { s->ent_r[cur[q]] = i2; s->ent_c[cur[q]] = i; cur[q]++; }
                    if (i != i2 && ca == cb) { s->ent_r[cur[q]] = i2; s->ent_c[cur[q]] = i; cur[q]++; }
                }
        }
    }
    // 4. chunks
    s->pair_chunk_ptr.assign((size_t)np + 1, 0);
    s->chunk_ptr.clear();
    s->chunk_pair.clear();
    for (long long q = 0; q < np; q++) {
        for (long long b = pcount[q]; b < pcount[q + 1]; b += chunk_len) {
            s->chunk_ptr.push_back((int)b);
            s->chunk_pair.push_back((int)q);
        }
        s->pair_chunk_ptr[q + 1] = (int)s->chunk_ptr.size();
    }
    s->nchunks = (int)s->chunk_pair.size();
    s->chunk_ptr.push_back((int)s->E);

    // 5. camera-sorted view of the observations (= self entries of the diagonal pairs), in chunks of dchunk_len.
    std::vector<int> cptr((size_t)N + 1, 0);
    for (int i = 0; i < Kl; i++) cptr[(size_t)s->obs_cam[i] + 1]++;
    for (int c = 0; c < N; c++) cptr[c + 1] += cptr[c];
    s->cam_obs.resize(Kl);
    {
        std::vector<int> cur(cptr.begin(), cptr.end() - 1);
        for (int i = 0; i < Kl; i++) s->cam_obs[cur[s->obs_cam[i]]++] = i;
    }
    s->cam_dchunk_ptr.assign((size_t)N + 1, 0);
    s->dchunk_ptr.clear();
    s->dchunk_cam.clear();
    for (int c = 0; c < N; c++) {
        for (int b = cptr[c]; b < cptr[c + 1]; b += dchunk_len) {
            s->dchunk_ptr.push_back(b);
            s->dchunk_cam.push_back(c);
        }
        s->cam_dchunk_ptr[c + 1] = (int)s->dchunk_cam.size();
    }
    s->ndchunks = (int)s->dchunk_cam.size();
    s->dchunk_ptr.push_back(Kl);
    return BA_OK;
}

extern "C" int ba_shard_plan(const ba_problem *p, int shard_rank, int shard_world, long long *out8)
{
    if (!p || !out8) return BA_ERR_ARG;
    ba_structure s;
    int rc = ba_build_structure(p, shard_rank, shard_world, 64, 32, &s);
    if (rc) return rc;
    out8[0] = s.p0; out8[1] = s.p1; out8[2] = s.o0; out8[3] = s.o1; out8[4] = s.E; out8[5] = s.nchunks; out8[6] = s.npairs;
    out8[7] = s.was_sorted ? 1 : 0;
    return BA_OK;
}
