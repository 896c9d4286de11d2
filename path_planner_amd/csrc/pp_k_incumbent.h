// pp_k_incumbent.h — incumbent selection (AStarPlanner.cpp:109-117 in batch form): lexicographic min of (f bits, edge index).  Included by pp_kernels.h.
#pragma once
// ------------------------------------------------------------------------------------------
// Incumbent selection: lexicographic min of (bits of f, edge index) over feasible edges — the
// batch form of `if (!best || v->f() < best->f()) best = v` (AStarPlanner.cpp:109-117).
// Stage 1: wave shuffle-reduce + LDS across the 4 waves -> one partial per workgroup;
// stage 2: one workgroup over the partials.  Deterministic (no atomics).
__device__ __forceinline__ void pp_key_min(unsigned long long& f, unsigned long long& i, unsigned long long f2, unsigned long long i2) {
    if (f2 < f || (f2 == f && i2 < i)) { f = f2; i = i2; }
}
__device__ __forceinline__ void pp_key_wave_min(unsigned long long& f, unsigned long long& i) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        unsigned long long f2 = __shfl_xor(f, o, PP_WAVE), i2 = __shfl_xor(i, o, PP_WAVE);
        pp_key_min(f, i, f2, i2);
    }
}
__global__ __launch_bounds__(256) void pp_k_best_stage1(const ppgpu_edge_result* res, long long n, int goal_only,
                                                        unsigned long long base, unsigned long long* partial) {
    __shared__ unsigned long long sf[4], si[4];
    unsigned long long f = ~0ull, idx = ~0ull;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long long)gridDim.x * 256) {
        unsigned fl = res[e].flags;
        bool ok = !(fl & PPGPU_F_INFEASIBLE) && (!goal_only || (fl & PPGPU_F_GOAL));
        if (ok) {
            unsigned long long fb = (unsigned long long)__double_as_longlong(res[e].f);
            pp_key_min(f, idx, fb, base + (unsigned long long)e);
        }
    }
    pp_key_wave_min(f, idx);
    if (pp_lane() == 0) { sf[threadIdx.x >> 6] = f; si[threadIdx.x >> 6] = idx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; w++) pp_key_min(f, idx, sf[w], si[w]);
        partial[2 * blockIdx.x] = f; partial[2 * blockIdx.x + 1] = idx;
    }
}
__global__ __launch_bounds__(256) void pp_k_best_stage2(const unsigned long long* partial, int nparts, unsigned long long* key2) {
    __shared__ unsigned long long sf[4], si[4];
    unsigned long long f = ~0ull, idx = ~0ull;
    for (int i = threadIdx.x; i < nparts; i += 256) pp_key_min(f, idx, partial[2 * i], partial[2 * i + 1]);
    pp_key_wave_min(f, idx);
    if (pp_lane() == 0) { sf[threadIdx.x >> 6] = f; si[threadIdx.x >> 6] = idx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; w++) pp_key_min(f, idx, sf[w], si[w]);
        key2[0] = f; key2[1] = idx;
    }
}
// after an all-gather of per-rank keys: lexicographic min of `n` (f, idx) pairs
__global__ void pp_k_key_min_n(const unsigned long long* keys, int n, unsigned long long* key2) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        unsigned long long f = ~0ull, idx = ~0ull;
        for (int i = 0; i < n; i++) pp_key_min(f, idx, keys[2 * i], keys[2 * i + 1]);
        key2[0] = f; key2[1] = idx;
    }
}
