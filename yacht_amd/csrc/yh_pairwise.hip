// yh_pairwise.hip -- `yacht train`: pairwise intersection counts from the posting lists (src/cpp/main.cpp:249-308)
#include "yh_common.h"

#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <vector>

namespace {

constexpr int WAVE = 64;

// ---- pairwise -------------------------------------------------------------------------------------
// Only references that hold at least one shared hash can be in a pair, so the dense count block is
// indexed by COMPACT ids (cid[ref], ascending with the reference id; rid[] maps back): a
// dereplicated database of 85 205 genomes has ~8 000 such references (0.3 GB instead of 29 GB).
//
// One thread per posting (a = its reference): for every other reference b of the same hash,
// M[cid[a] - c0][cid[b]] += 1.  Integer atomics: the result does not depend on arrival order.
// A posting whose hash has few holders walks the list itself; a list of more than PAIR_LONG holders is
// walked by the whole wave, one holder per lane (a k-mer shared by M references is M^2 increments either
// way, but M serial steps per posting instead of M / 64 made one conserved k-mer the tail of the launch).
constexpr u32 PAIR_LONG = 32;
__global__ void __launch_bounds__(256) k_pair_accum(u64 n_post, const u32* __restrict__ pr, const u32* __restrict__ pg,
                                                    const u64* __restrict__ po, const u32* __restrict__ cid, u64 c0, u64 c1,
                                                    u64 n_c, u32* __restrict__ M) {
    const u32 lane = threadIdx.x & 63u;
    const u64 wave = (blockIdx.x * (u64)blockDim.x + threadIdx.x) >> 6;
    const u64 n_waves = ((u64)gridDim.x * blockDim.x) >> 6;
    for (u64 k0 = wave * 64; k0 < n_post; k0 += n_waves * 64) {
        const u64 k = k0 + lane;
        u32 a = 0, ca = 0xffffffffu;
        u64 b = 0, e = 0;
        if (k < n_post) {
            a = pr[k];
            ca = cid[a];
            if (ca >= c0 && ca < c1) {
                const u32 gi = pg[k];
                b = po[gi];
                e = po[gi + 1];
            }
        }
        const bool mine = e > b;
        const bool is_long = mine && (e - b) > PAIR_LONG;
        if (mine && !is_long) {
            u32* row = M + (u64)(ca - c0) * n_c;
            for (u64 q = b; q < e; ++q) {
                const u32 o = pr[q];
                if (o != a) atomicAdd(&row[cid[o]], 1u);
            }
        }
        u64 todo = __ballot(is_long);
        while (todo) {
            const int src = __ffsll((long long)todo) - 1;
            todo &= todo - 1;
            const u32 a_s = (u32)__shfl((int)a, src), ca_s = (u32)__shfl((int)ca, src);
            const u64 b_s = ((u64)(u32)__shfl((int)(u32)(b >> 32), src) << 32) | (u32)__shfl((int)(u32)b, src);
            const u64 e_s = ((u64)(u32)__shfl((int)(u32)(e >> 32), src) << 32) | (u32)__shfl((int)(u32)e, src);
            u32* row = M + (u64)(ca_s - c0) * n_c;
            for (u64 q = b_s + lane; q < e_s; q += 64) {
                const u32 o = pr[q];
                if (o != a_s) atomicAdd(&row[cid[o]], 1u);
            }
        }
    }
}

__device__ __forceinline__ bool pair_keep(u32 cnt, u32 i, u32 j, const u32* __restrict__ sizes, double c_relaxed) {
    if (cnt == 0 || i == j) return false;
    const u32 si = sizes[i], sj = sizes[j];
    if (si == 0 || sj == 0) return false;
    // relaxed device-side filter; the exact `!(1.0*cnt/|R_i| < C)` of main.cpp:297-303 is applied
    // on the host to the survivors, so no decision depends on device floating point
    return !((double)cnt / (double)si < c_relaxed);
}

// one wave per row: count survivors
__global__ void __launch_bounds__(256) k_pair_count(const u32* __restrict__ M, u64 r0, u64 r1, u64 n_refs,
                                                    const u32* __restrict__ rid, const u32* __restrict__ sizes,
                                                    double c_relaxed, u32* __restrict__ rowcnt) {
    const u64 wave = (blockIdx.x * (u64)blockDim.x + threadIdx.x) / WAVE;
    const u64 n_waves = ((u64)gridDim.x * blockDim.x) / WAVE;
    const int lane = threadIdx.x & (WAVE - 1);
    for (u64 i = r0 + wave; i < r1; i += n_waves) {
        const u32* row = M + (i - r0) * n_refs;
        u32 c = 0;
        for (u64 j0 = 0; j0 < n_refs; j0 += WAVE) {
            const u64 j = j0 + lane;
            const bool keep = (j < n_refs) && pair_keep(row[j], rid[i], rid[j], sizes, c_relaxed);
            c += (u32)__popcll(__ballot(keep));
        }
        if (lane == 0) rowcnt[i - r0] = c;
    }
}

__global__ void __launch_bounds__(1024) k_scan_u32_to_u64(const u32* __restrict__ in, u64 n, u64* __restrict__ out) {
    // single workgroup; out[n] = total
    __shared__ u64 wsum[17];
    const int lane = threadIdx.x & (WAVE - 1);
    const int wid = threadIdx.x / WAVE;
    const int nw = blockDim.x / WAVE;
    u64 carry = 0;
    for (u64 base = 0; base < n; base += blockDim.x) {
        const u64 i = base + threadIdx.x;
        const u64 v = (i < n) ? in[i] : 0;
        u64 inc = v;
#pragma unroll
        for (int d = 1; d < WAVE; d <<= 1) {
            const u64 t = __shfl_up(inc, d, WAVE);
            if (lane >= d) inc += t;
        }
        if (lane == WAVE - 1) wsum[wid] = inc;
        __syncthreads();
        if (threadIdx.x == 0) {
            u64 acc = 0;
            for (int w = 0; w < nw; ++w) { const u64 t = wsum[w]; wsum[w] = acc; acc += t; }
            wsum[16] = acc;
        }
        __syncthreads();
        if (i < n) out[i] = carry + wsum[wid] + inc - v;
        carry += wsum[16];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[n] = carry;
}

// one wave per row: ordered compaction (j ascending inside a row, rows ascending)
__global__ void __launch_bounds__(256) k_pair_emit(const u32* __restrict__ M, u64 r0, u64 r1, u64 n_refs,
                                                   const u32* __restrict__ rid, const u32* __restrict__ sizes,
                                                   double c_relaxed,
                                                   const u64* __restrict__ rowoff, u32* __restrict__ out_i,
                                                   u32* __restrict__ out_j, u32* __restrict__ out_c) {
    const u64 wave = (blockIdx.x * (u64)blockDim.x + threadIdx.x) / WAVE;
    const u64 n_waves = ((u64)gridDim.x * blockDim.x) / WAVE;
    const int lane = threadIdx.x & (WAVE - 1);
    for (u64 i = r0 + wave; i < r1; i += n_waves) {
        const u32* row = M + (i - r0) * n_refs;
        u64 w = rowoff[i - r0];
        for (u64 j0 = 0; j0 < n_refs; j0 += WAVE) {
            const u64 j = j0 + lane;
            u32 cnt = 0;
            bool keep = false;
            if (j < n_refs) {
                cnt = row[j];
                keep = pair_keep(cnt, rid[i], rid[j], sizes, c_relaxed);
            }
            const u64 bal = __ballot(keep);
            if (keep) {
                const u64 dst = w + __popcll(bal & ((1ull << lane) - 1ull));
                out_i[dst] = rid[i];
                out_j[dst] = rid[j];
                out_c[dst] = cnt;
            }
            w += __popcll(bal);
        }
    }
}
inline u32 grid_for(u64 work_items, u32 block, u32 max_blocks = 16384) {
    u64 g = (work_items + block - 1) / block;
    if (g < 1) g = 1;
    if (g > max_blocks) g = max_blocks;
    return (u32)g;
}

}  // namespace

// Fills the handle's host-side pair cache (h_pw_*) for rows [r0, r1).
int yh_q_pairwise(yh_db* db, double c_thresh, u64 r0, u64 r1) {
    if (!db->has_index) { yh_set_error("this handle was created with YH_DB_NO_INDEX"); return YH_ERR_UNSUPPORTED; }
    hipStream_t st = db->stream;
    const u64 N = db->n_refs;
    free(db->h_pw_i); free(db->h_pw_j); free(db->h_pw_c);
    db->h_pw_i = db->h_pw_j = db->h_pw_c = nullptr;
    db->pw_n = 0;
    db->pw_valid = false;
    if (r1 > N) r1 = N;
    if (r0 >= r1) { db->pw_valid = true; db->pw_c = c_thresh; db->pw_r0 = r0; db->pw_r1 = r1; return YH_OK; }

    // compact ids of the references that hold a shared hash (ascending with the reference id)
    std::vector<u32> h_nsh(N), h_cid(N), h_rid;
    YH_HIP(hipMemcpyAsync(h_nsh.data(), db->d_nshared, N * sizeof(u32), hipMemcpyDeviceToHost, st));
    YH_HIP(hipStreamSynchronize(st));
    for (u64 j = 0; j < N; ++j) {
        if (h_nsh[j]) { h_cid[j] = (u32)h_rid.size(); h_rid.push_back((u32)j); }
        else h_cid[j] = 0xffffffffu;
    }
    const u64 NC = h_rid.size();
    const u64 c_begin = std::lower_bound(h_rid.begin(), h_rid.end(), (u32)r0) - h_rid.begin();
    const u64 c_end = std::lower_bound(h_rid.begin(), h_rid.end(), (u32)std::min<u64>(r1, 0xffffffffull)) - h_rid.begin();
    if (NC == 0 || c_begin >= c_end) { db->pw_valid = true; db->pw_c = c_thresh; db->pw_r0 = r0; db->pw_r1 = r1; return YH_OK; }

    // dense row blocks (compact rows x compact columns) of int32 counts: at most ~32 GiB, and at most 60 % of what the
    // device has free now (other handles, other ranks sharing the GPU); halved again when the allocation still fails
    u64 budget = 32ull << 30;
    {
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && free_b > 0) budget = std::min<u64>(budget, (u64)((double)free_b * 0.6));
        else (void)hipGetLastError();
    }
    u64 rows_per_block = std::max<u64>(1, budget / (NC * sizeof(u32)));
    if (rows_per_block > c_end - c_begin) rows_per_block = c_end - c_begin;
    const double c_relaxed = c_thresh * (1.0 - 1e-9) - 1e-300;

    u32 *d_M = nullptr, *d_rowcnt = nullptr, *d_oi = nullptr, *d_oj = nullptr, *d_oc = nullptr;
    u32 *d_cid = nullptr, *d_rid = nullptr;
    u64* d_rowoff = nullptr;
    std::vector<u32> hi, hj, hc;
    int rc = YH_OK;
#define PW_HIP(call)                                                                          \
    if (rc == YH_OK) {                                                                        \
        hipError_t e__ = (call);                                                              \
        if (e__ != hipSuccess) {                                                              \
            yh_set_error("%s failed: %s", #call, hipGetErrorString(e__));                     \
            rc = (e__ == hipErrorOutOfMemory) ? YH_ERR_OOM : YH_ERR_HIP;                      \
        }                                                                                     \
    }
    for (;;) {
        const hipError_t em = hipMalloc((void**)&d_M, rows_per_block * NC * sizeof(u32));
        if (em == hipSuccess) break;
        (void)hipGetLastError();
        d_M = nullptr;
        if (em != hipErrorOutOfMemory || rows_per_block == 1) {
            yh_set_error("hipMalloc of the %llu-row count block failed: %s", (u64)rows_per_block, hipGetErrorString(em));
            rc = (em == hipErrorOutOfMemory) ? YH_ERR_OOM : YH_ERR_HIP;
            break;
        }
        rows_per_block = (rows_per_block + 1) / 2;
    }
    PW_HIP(hipMalloc((void**)&d_rowcnt, rows_per_block * sizeof(u32)));
    PW_HIP(hipMalloc((void**)&d_rowoff, (rows_per_block + 1) * sizeof(u64)));
    PW_HIP(hipMalloc((void**)&d_cid, N * sizeof(u32)));
    PW_HIP(hipMalloc((void**)&d_rid, NC * sizeof(u32)));
    PW_HIP(hipMemcpyAsync(d_cid, h_cid.data(), N * sizeof(u32), hipMemcpyHostToDevice, st));
    PW_HIP(hipMemcpyAsync(d_rid, h_rid.data(), NC * sizeof(u32), hipMemcpyHostToDevice, st));
    yh_ring_record_begin(db, db->ev_pair);
    for (u64 b0 = c_begin; b0 < c_end && rc == YH_OK; b0 += rows_per_block) {
        const u64 b1 = std::min(c_end, b0 + rows_per_block);
        const u64 rows = b1 - b0;
        PW_HIP(hipMemsetAsync(d_M, 0, rows * NC * sizeof(u32), st));
        if (rc == YH_OK && db->n_postings) {
            k_pair_accum<<<grid_for(db->n_postings, 256, 1u << 20), 256, 0, st>>>(db->n_postings, db->d_pr, db->d_pg,
                                                                                  db->d_po, d_cid, b0, b1, NC, d_M);
        }
        if (rc == YH_OK) {
            k_pair_count<<<grid_for(rows * WAVE, 256, 8192), 256, 0, st>>>(d_M, b0, b1, NC, d_rid, db->d_sizes, c_relaxed,
                                                                           d_rowcnt);
            k_scan_u32_to_u64<<<1, 1024, 0, st>>>(d_rowcnt, rows, d_rowoff);
        }
        PW_HIP(hipGetLastError());
        u64 n_out = 0;
        PW_HIP(hipMemcpyAsync(&n_out, d_rowoff + rows, sizeof(u64), hipMemcpyDeviceToHost, st));
        PW_HIP(hipStreamSynchronize(st));
        if (rc == YH_OK && n_out) {
            PW_HIP(hipMalloc((void**)&d_oi, n_out * sizeof(u32)));
            PW_HIP(hipMalloc((void**)&d_oj, n_out * sizeof(u32)));
            PW_HIP(hipMalloc((void**)&d_oc, n_out * sizeof(u32)));
            if (rc == YH_OK) {
                k_pair_emit<<<grid_for(rows * WAVE, 256, 8192), 256, 0, st>>>(d_M, b0, b1, NC, d_rid, db->d_sizes,
                                                                              c_relaxed, d_rowoff, d_oi, d_oj, d_oc);
            }
            PW_HIP(hipGetLastError());
            const size_t base = hi.size();
            hi.resize(base + n_out); hj.resize(base + n_out); hc.resize(base + n_out);
            PW_HIP(hipMemcpyAsync(hi.data() + base, d_oi, n_out * sizeof(u32), hipMemcpyDeviceToHost, st));
            PW_HIP(hipMemcpyAsync(hj.data() + base, d_oj, n_out * sizeof(u32), hipMemcpyDeviceToHost, st));
            PW_HIP(hipMemcpyAsync(hc.data() + base, d_oc, n_out * sizeof(u32), hipMemcpyDeviceToHost, st));
            PW_HIP(hipStreamSynchronize(st));
            (void)hipFree(d_oi); (void)hipFree(d_oj); (void)hipFree(d_oc);
            d_oi = d_oj = d_oc = nullptr;
        }
    }
    yh_ring_record_end(db, db->ev_pair);
#undef PW_HIP
    (void)hipFree(d_M); (void)hipFree(d_rowcnt); (void)hipFree(d_rowoff);
    (void)hipFree(d_oi); (void)hipFree(d_oj); (void)hipFree(d_oc);
    (void)hipFree(d_cid); (void)hipFree(d_rid);
    if (rc != YH_OK) return rc;

    // exact host-side filter (main.cpp:297-303): keep iff !(1.0*count/|R_i| < C)
    std::vector<u32> hsizes(N);
    YH_HIP(hipMemcpy(hsizes.data(), db->d_sizes, N * sizeof(u32), hipMemcpyDeviceToHost));
    size_t w = 0;
    for (size_t k = 0; k < hi.size(); ++k) {
        const double cij = 1.0 * hc[k] / hsizes[hi[k]];
        if (cij < c_thresh) continue;
        hi[w] = hi[k]; hj[w] = hj[k]; hc[w] = hc[k];
        ++w;
    }
    db->pw_n = w;
    db->h_pw_i = (u32*)malloc(std::max<size_t>(w, 1) * sizeof(u32));
    db->h_pw_j = (u32*)malloc(std::max<size_t>(w, 1) * sizeof(u32));
    db->h_pw_c = (u32*)malloc(std::max<size_t>(w, 1) * sizeof(u32));
    if (!db->h_pw_i || !db->h_pw_j || !db->h_pw_c) { yh_set_error("host allocation failed"); return YH_ERR_OOM; }
    memcpy(db->h_pw_i, hi.data(), w * sizeof(u32));
    memcpy(db->h_pw_j, hj.data(), w * sizeof(u32));
    memcpy(db->h_pw_c, hc.data(), w * sizeof(u32));
    db->pw_valid = true;
    db->pw_c = c_thresh;
    db->pw_r0 = r0;
    db->pw_r1 = r1;
    return YH_OK;
}
