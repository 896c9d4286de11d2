// ppgpu.hip — C ABI (include/ppgpu.h) over the gfx950 kernels in pp_kernels.h / pp_sampler.h.
//
// Build (see __graft_entry__.build()):
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -shared ppgpu.hip -o libppgpu.so
// -ffp-contract=off is part of the contract: the reference rounds every product and sum
// separately (baseline x86-64), and so must the device code.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/ppgpu.h"
#include "pp_kernels.h"
#include "pp_sampler.h"

static thread_local std::string g_err = "";
static int fail(int code, const std::string& msg) { g_err = msg; return code; }

#define HIP_TRY(call)                                                                              \
    do {                                                                                           \
        hipError_t _e = (call);                                                                    \
        if (_e != hipSuccess)                                                                      \
            return fail(PPGPU_EHIP, std::string(#call) + ": " + hipGetErrorString(_e));            \
    } while (0)

// Workspace of one costing slice: PPEdgeSetup + what the pose sweep leaves for the cover sweep, (256 + 2 * ngp + 12 * nch + 16)
// bytes per edge (~3.6 KB per edge for a 1 500-step horizon: 0.9 GB for the 236 140 edges of the bench step; the Gaussian
// obstacle model adds 8 bytes per step).  A launch whose workspace
// would exceed PP_SLICE_BYTES runs as consecutive slices.  (Running slice i's cover sweep next to slice i+1's pose sweep
// on a second stream was measured and gains nothing: both sweeps are bound by fp64 VALU issue.)
#ifndef PP_PREPASS_MIN_EDGES
#define PP_PREPASS_MIN_EDGES 8192   // launches below this skip pp_k_plan_skips / pp_k_approach_events (their results are optional)
#endif
#ifndef PP_SLICE_BYTES
#define PP_SLICE_BYTES (32ull << 30)
#endif
// Device / pinned allocations made after start-up are what a real-time caller must know about (a hipMalloc or hipHostMalloc inside a
// 100 ms planning cycle costs milliseconds): every growth of a library buffer is counted, with the wall time it took
// (ppgpu_growth_stats; process-wide, the host planner reports the difference over a plan() call).
#include <atomic>
#include <chrono>
static std::atomic<unsigned long long> g_growth_count{0}, g_growth_ns{0};
struct GrowthTimer {
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    ~GrowthTimer() {
        g_growth_count.fetch_add(1, std::memory_order_relaxed);
        g_growth_ns.fetch_add((unsigned long long)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count(), std::memory_order_relaxed);
    }
};
template <typename T>
struct DevBuf {
    T* p = nullptr;
    size_t cap = 0;  // elements
    int reserve(size_t n, bool keep, hipStream_t s) {
        if (n <= cap) return PPGPU_OK;
        GrowthTimer growth;
        size_t ncap = cap ? cap : 64;
        while (ncap < n) ncap *= 2;
        T* np = nullptr;
        HIP_TRY(hipMalloc((void**)&np, ncap * sizeof(T)));
        if (keep && p && cap) {
            HIP_TRY(hipMemcpyAsync(np, p, cap * sizeof(T), hipMemcpyDeviceToDevice, s));
            HIP_TRY(hipStreamSynchronize(s));
        }
        if (p) HIP_TRY(hipFree(p));
        p = np; cap = ncap;
        return PPGPU_OK;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
};

#define PP_TIMING_RING 8
struct ppgpu_ctx {
    int device = 0;
    hipStream_t own_stream = nullptr, stream = nullptr;
    hipStream_t side_stream = nullptr;                 // pp_k_heuristic_listed runs beside the lane heuristic (launch_cost)
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    bool have_cfg = false;
    ppgpu_config cfg{};
    // Map
    DevBuf<uint32_t> grid;
    DevBuf<unsigned char> grid_clear, grid_rowclear;   // clearance map of the grid (PPGrid::clearance) and its row pass
    int rows = 0, cols = 0, wpr = 0;
    double res = 0;
    // dynamic obstacles
    DevBuf<PPObst> obst;
    int n_obst = 0;
    int obst_model = PPGPU_OBST_NONE;
    // open vertices
    DevBuf<ppgpu_vertex> verts;
    DevBuf<double> ribbons, tgrid;
    int nverts = 0, nribbons = 0, ng = 0;
    // sample (target state) store, SoA
    DevBuf<double> sx, sy, sh;
    long long n_samples = 0;
    long long n_extra = 0;              // explicit target states appended behind the samples (ppgpu_set_extra_targets)
    // sampler (StateGenerator) state
    PPSamplerState sampler{};
    DevBuf<double> samp_ribbons;
    DevBuf<unsigned long long> samp_pos;           // groups of 8 words: [0] stream position (pair slots consumed), [1] sticky chain error, [2..3] report of the last
                                                   // add (pp_k_sampler_advance).  ppgpu_sampler_init moves on to the next group — all zero: "position 0, no error" —
                                                   // and zeroes the whole array again when it wraps: a fresh generator costs no launch of its own
    int samp_group = -1;
    unsigned long long* pinned_counts = nullptr;   // 64 bytes of pinned host memory for the sampler's count read-backs
    std::vector<double> samp_ribbons_host;   // what samp_ribbons holds (ppgpu_sampler_init skips the upload of an unchanged table)
    DevBuf<unsigned char> s_bytes;      // scan / compaction scratch
    DevBuf<unsigned long long> s_u64;
    DevBuf<unsigned> s_u32a, s_u32b;
    DevBuf<double> s_cand;              // candidate states before the map filter
    // scratch for host-convenience entry points and reductions
    DevBuf<unsigned long long> tmp_edges, partial;
    DevBuf<ppgpu_wrapper_edge> tmp_wedges;
    DevBuf<ppgpu_edge_result> tmp_results;
    DevBuf<double> tmp_child, tmp_lengths, tmp_len_out, int_child;
    bool quiet_finish = true;           // env PPGPU_QUIET_FINISH=0: every edge's phase C stays with its wave
    bool lane_split = true;             // env PPGPU_LANE_SPLIT=0: the wave makes every split itself (tests compare the two)
    bool lane_finish = true;            // env PPGPU_LANE_FINISH=0: every wave of the cover sweep finishes its own edges (tests compare the two)
    bool lane_heuristic = true;         // env PPGPU_LANE_HEURISTIC=0: large launches keep the wave-per-edge enumeration too (tests compare the two)
    long long prepass_min_edges = PP_PREPASS_MIN_EDGES;   // env PPGPU_PREPASS_MIN_EDGES overrides (tests run the prepasses on small launches too)
    size_t slice_bytes = PP_SLICE_BYTES; // workspace budget of one costing slice (env PPGPU_SLICE_BYTES overrides: tests)
    DevBuf<PPEdgeSetup> setup;          // workspace of the current costing slice: phase-0 records ...
    DevBuf<unsigned short> track_hits;  // ... and the pose sweep's track (see PPParams)
    DevBuf<unsigned long long> track_eq;
    DevBuf<unsigned> track_chunk_hits;
    DevBuf<PPTrackSummary> track_summary;
    DevBuf<int2> track_far;
    DevBuf<unsigned char> track_skip;
    DevBuf<unsigned> need_big;          // [0] need_big, [1 + n] number of deferred edges with n ribbons (pp_k_deferred_list)
    DevBuf<unsigned> defer_list, live_list, hw_list;
    DevBuf<PPCoverState> cover_state;
    DevBuf<unsigned long long> work;    // queue heads of the resident per-edge grids (PP_Q_*)
    int n_cu = 0;
    int resident[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // workgroups of each per-edge kernel the device holds at once (0 = not asked yet)
    DevBuf<double> track_pen, track_chunk_pen;   // Gaussian obstacle model only
    // optional per-kernel timing of costing launches (ppgpu_enable_timing)
    bool timing = false;
    // a ring of event sets: the last PP_TIMING_RING launches can be read back without a host wait between them
    hipEvent_t ev_ring[PP_TIMING_RING][6] = {};   // [0..4] as the launch goes; [5] between the approach prepass and the cover sweep
    double ms_ring[PP_TIMING_RING][4] = {};    // solve / pose / cover / approach time of the slices before the last one of a sliced launch
    hipEvent_t* ev = ev_ring[0];               // the set of the launch being recorded / recorded last
    double* ms_earlier_slices = ms_ring[0];
    int ev_slot = 0;
    long long last_launch_edges = 0;           // edges of the last costing launch, and whether its cover sweep took a packed list
    bool last_launch_packed = false;           // (then the list's length is at need_big[12]: last slice)
    long long live_earlier_slices = 0;
    long long ev_launches = 0;                 // timed launches so far
    int max_vertex_ribbons = 0;
    DevBuf<int> tmp_idx;
    DevBuf<double> ord_key;             // pp_k_expand_order: candidate scratch beyond what LDS holds, push-order output, fallback counter
    DevBuf<int> ord_val, ord_idx, ord_blockcnt, ord_count, near_idx, near_count;   // (near_*: the samples within a vertex's probe bound, pp_k_expand_near)
    DevBuf<double> ord_len, ord_blockmin, ord_bound, probe_bound;
    DevBuf<unsigned> ord_fallbacks;
    unsigned long long order_fallbacks = 0;   // (vertex, radius) lists of ppgpu_expand_host / ppgpu_expand_order that fell back to ascending length
    DevBuf<unsigned char> dstage_in, dstage_out;            // device ends of ppgpu_expand_host's single upload / download
    void* stage_in = nullptr; size_t stage_in_cap = 0;      // pinned host staging of ppgpu_expand_host
    void* stage_out = nullptr; size_t stage_out_cap = 0;
    DevBuf<unsigned long long> gather;
    void* comm = nullptr;               // ncclComm_t owned by the handle (ppgpu_comm_init_rank / ppgpu_comm_init_all)
    // ppgpu_copy_engine_read: the HSA agents of this device and of the host, the completion signal of the copy in flight
    unsigned long long hsa_gpu = 0, hsa_cpu = 0, hsa_signal = 0;
    bool copy_pending = false;
};

static int require_cfg(ppgpu_ctx* c) {
    if (!c) return fail(PPGPU_EINVAL, "null context");
    if (!c->have_cfg) return fail(PPGPU_ESTATE, "ppgpu_set_config must be called first");
    return PPGPU_OK;
}

static void ppgpu_copy_engine_release(ppgpu_ctx* c);      // (further down, with the HSA entry points)
extern "C" int ppgpu_copy_engine_wait(ppgpu_ctx* c);

extern "C" {

const char* ppgpu_last_error(void) { return g_err.c_str(); }

int ppgpu_create(int device, ppgpu_ctx** out) {
    if (!out) return fail(PPGPU_EINVAL, "out is null");
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return fail(PPGPU_ENODEV, "no HIP device visible");
    if (device < 0 || device >= n) return fail(PPGPU_ENODEV, "device index out of range");
    HIP_TRY(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(PPGPU_ENODEV, std::string("device is ") + prop.gcnArchName + ", this library carries gfx950 code only");
    ppgpu_ctx* c = new ppgpu_ctx();
    c->device = device;
    c->n_cu = prop.multiProcessorCount;
    HIP_TRY(hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking));
    c->stream = c->own_stream;
    HIP_TRY(hipStreamCreateWithFlags(&c->side_stream, hipStreamNonBlocking));
    HIP_TRY(hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming));
    HIP_TRY(hipHostMalloc((void**)&c->pinned_counts, 64, hipHostMallocDefault));
    if (const char* pm = std::getenv("PPGPU_PREPASS_MIN_EDGES")) c->prepass_min_edges = std::atoll(pm);
    if (const char* lh = std::getenv("PPGPU_LANE_HEURISTIC")) c->lane_heuristic = std::atoi(lh) != 0;
    if (const char* qf = std::getenv("PPGPU_QUIET_FINISH")) c->quiet_finish = std::atoi(qf) != 0;
    if (const char* lf = std::getenv("PPGPU_LANE_FINISH")) c->lane_finish = std::atoi(lf) != 0;
    if (const char* ls = std::getenv("PPGPU_LANE_SPLIT")) c->lane_split = std::atoi(ls) != 0;
    if (const char* sb = std::getenv("PPGPU_SLICE_BYTES")) {
        const long long v = std::atoll(sb);
        if (v > 0) c->slice_bytes = (size_t)v;
    }
    *out = c;
    return PPGPU_OK;
}

int ppgpu_destroy(ppgpu_ctx* c) {
    if (!c) return PPGPU_OK;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    if (c->comm) (void)ppgpu_comm_destroy(c);
    if (c->copy_pending) (void)ppgpu_copy_engine_wait(c);
    ppgpu_copy_engine_release(c);
    c->grid.release(); c->grid_clear.release(); c->grid_rowclear.release(); c->obst.release(); c->verts.release(); c->ribbons.release(); c->tgrid.release();
    c->sx.release(); c->sy.release(); c->sh.release(); c->samp_ribbons.release(); c->samp_pos.release();
    c->s_bytes.release(); c->s_u64.release(); c->s_u32a.release(); c->s_u32b.release(); c->s_cand.release();
    c->tmp_edges.release(); c->tmp_wedges.release(); c->partial.release(); c->tmp_results.release(); c->tmp_child.release();
    c->ord_key.release(); c->ord_val.release(); c->ord_idx.release(); c->ord_fallbacks.release(); c->ord_len.release();
    c->ord_blockmin.release(); c->ord_blockcnt.release(); c->ord_bound.release(); c->ord_count.release(); c->near_idx.release(); c->near_count.release(); c->probe_bound.release();
    c->tmp_lengths.release(); c->tmp_len_out.release(); c->tmp_idx.release(); c->gather.release(); c->int_child.release();
    c->setup.release(); c->track_hits.release(); c->track_eq.release(); c->track_chunk_hits.release();
    c->track_summary.release(); c->track_far.release(); c->track_skip.release(); c->track_pen.release(); c->track_chunk_pen.release(); c->need_big.release(); c->defer_list.release(); c->live_list.release(); c->hw_list.release(); c->cover_state.release(); c->work.release(); c->dstage_in.release(); c->dstage_out.release();
    if (c->pinned_counts) (void)hipHostFree(c->pinned_counts);
    if (c->stage_in) (void)hipHostFree(c->stage_in);
    if (c->stage_out) (void)hipHostFree(c->stage_out);
    for (int r = 0; r < PP_TIMING_RING; r++) for (int i = 0; i < 6; i++) if (c->ev_ring[r][i]) (void)hipEventDestroy(c->ev_ring[r][i]);
    if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
    if (c->ev_join) (void)hipEventDestroy(c->ev_join);
    if (c->side_stream) (void)hipStreamDestroy(c->side_stream);
    if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
    delete c;
    return PPGPU_OK;
}

int ppgpu_set_stream(ppgpu_ctx* c, void* s) {
    if (!c) return fail(PPGPU_EINVAL, "null context");
    c->stream = s ? (hipStream_t)s : c->own_stream;
    return PPGPU_OK;
}

int ppgpu_enable_timing(ppgpu_ctx* c, int32_t on) {
    if (!c) return fail(PPGPU_EINVAL, "null context");
    HIP_TRY(hipSetDevice(c->device));
    if (on && !c->ev_ring[0][0])
        for (int r = 0; r < PP_TIMING_RING; r++) for (int i = 0; i < 6; i++) HIP_TRY(hipEventCreate(&c->ev_ring[r][i]));
    c->timing = on != 0;
    c->ev_launches = 0;
    return PPGPU_OK;
}

int ppgpu_past_timing(ppgpu_ctx* c, int32_t back, double* ms_solve, double* ms_pose, double* ms_cover, double* ms_heuristic) {
    if (!c || !ms_solve || !ms_pose || !ms_cover || !ms_heuristic) return fail(PPGPU_EINVAL, "null argument");
    if (!c->timing || c->ev_launches == 0) return fail(PPGPU_ESTATE, "no timed costing launch (ppgpu_enable_timing, then cost edges)");
    if (back < 0 || back >= PP_TIMING_RING || back >= c->ev_launches) return fail(PPGPU_EINVAL, "past_timing: that launch is no longer (or not yet) in the ring");
    HIP_TRY(hipSetDevice(c->device));
    const int slot = (c->ev_slot - back + PP_TIMING_RING) % PP_TIMING_RING;
    hipEvent_t* ev = c->ev_ring[slot];
    HIP_TRY(hipEventSynchronize(ev[4]));
    float t[4] = {0, 0, 0, 0}, ta = 0;
    for (int i = 0; i < 4; i++) HIP_TRY(hipEventElapsedTime(&t[i], ev[i == 2 ? 5 : i], ev[i + 1]));   // the cover sweep alone: from event 5
    HIP_TRY(hipEventElapsedTime(&ta, ev[2], ev[5]));                                                   // the approach prepass
    *ms_solve = t[0] + c->ms_ring[slot][0]; *ms_pose = t[1] + c->ms_ring[slot][1]; *ms_cover = t[2] + c->ms_ring[slot][2];
    *ms_heuristic = t[3] + ta + c->ms_ring[slot][3];
    return PPGPU_OK;
}
int ppgpu_last_timing(ppgpu_ctx* c, double* ms_solve, double* ms_pose, double* ms_cover, double* ms_heuristic) {
    return ppgpu_past_timing(c, 0, ms_solve, ms_pose, ms_cover, ms_heuristic);
}

int ppgpu_last_cover_edges(ppgpu_ctx* c, int64_t* n_edges) {
    if (!c || !n_edges) return fail(PPGPU_EINVAL, "null argument");
    HIP_TRY(hipSetDevice(c->device));
    *n_edges = c->last_launch_edges;
    if (c->last_launch_packed) {
        unsigned live = 0;
        HIP_TRY(hipStreamSynchronize(c->stream));
        HIP_TRY(hipMemcpy(&live, c->need_big.p + 12, sizeof(unsigned), hipMemcpyDeviceToHost));
        *n_edges = (int64_t)live + c->live_earlier_slices;
    }
    return PPGPU_OK;
}

int ppgpu_reserve_samples(ppgpu_ctx* c, int64_t max_samples, int32_t max_vertices) {
    if (!c) return fail(PPGPU_EINVAL, "null context");
    if (max_samples <= 0 || max_vertices <= 0) return fail(PPGPU_EINVAL, "reserve_samples: sizes must be positive");
    HIP_TRY(hipSetDevice(c->device));
    hipStream_t st = c->stream;
    const size_t ns = (size_t)max_samples, nv = (size_t)max_vertices;
    const size_t nblk = (ns + 255) / 256;
    size_t cap = 64;
    while (cap < ns && cap < 65536) cap <<= 1;
    const size_t nq = 6 * (size_t)524288 + 8;                   // the sampler's largest batch (ppgpu_sampler_add)
    int rc;
    if ((rc = c->sx.reserve(ns + nv, true, st)) || (rc = c->sy.reserve(ns + nv, true, st)) || (rc = c->sh.reserve(ns + nv, true, st)) ||
        (rc = c->s_cand.reserve((size_t)524288 * 3, false, st)) || (rc = c->s_bytes.reserve(nq + 64, false, st)) || (rc = c->s_u32a.reserve(nq + 64, false, st)) ||
        (rc = c->tmp_lengths.reserve(nv * ns * 2, false, st)) ||
        (rc = c->ord_key.reserve(nv * 2 * cap, false, st)) || (rc = c->ord_val.reserve(nv * 2 * cap, false, st)) || (rc = c->ord_len.reserve(nv * 2 * cap, false, st)) ||
        (rc = c->ord_blockmin.reserve(nv * nblk * 2 * (256 / PP_NEAR_GROUP), false, st)) || (rc = c->ord_blockcnt.reserve(nv * nblk * (256 / PP_NEAR_GROUP), false, st)) ||
        (rc = c->near_idx.reserve(nv * ns, false, st)) || (rc = c->near_count.reserve(nv, false, st)) || (rc = c->probe_bound.reserve(nv * 2 * PP_PROBE_GROUPS, false, st)))
        return rc;
    return PPGPU_OK;
}

int ppgpu_growth_stats(ppgpu_ctx* c, uint64_t* count, double* seconds) {
    if (!c) return fail(PPGPU_EINVAL, "null context");
    if (count) *count = g_growth_count.load(std::memory_order_relaxed);
    if (seconds) *seconds = 1e-9 * (double)g_growth_ns.load(std::memory_order_relaxed);
    return PPGPU_OK;
}

int ppgpu_synchronize(ppgpu_ctx* c) {
    if (!c) return fail(PPGPU_EINVAL, "null context");
    HIP_TRY(hipStreamSynchronize(c->stream));
    return PPGPU_OK;
}

int ppgpu_device_alloc(ppgpu_ctx* c, uint64_t bytes, void** d_out) {
    if (!c || !d_out || bytes == 0) return fail(PPGPU_EINVAL, "device_alloc: bad arguments");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipMalloc(d_out, (size_t)bytes));
    return PPGPU_OK;
}
int ppgpu_device_free(ppgpu_ctx* c, void* d_ptr) {
    if (!c) return fail(PPGPU_EINVAL, "null context");
    if (!d_ptr) return PPGPU_OK;
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(hipFree(d_ptr));
    return PPGPU_OK;
}
int ppgpu_device_read(ppgpu_ctx* c, void* h_dst, const void* d_src, uint64_t bytes) {
    if (!c || !h_dst || !d_src) return fail(PPGPU_EINVAL, "device_read: null argument");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipMemcpyAsync(h_dst, d_src, (size_t)bytes, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return PPGPU_OK;
}

int ppgpu_set_config(ppgpu_ctx* c, const ppgpu_config* cfg) {
    if (!c || !cfg) return fail(PPGPU_EINVAL, "null argument");
    if (!(cfg->max_speed > 0) || !(cfg->collision_checking_increment > 0) || !(cfg->turning_radius > 0) ||
        !(cfg->coverage_turning_radius > 0) || !(cfg->time_horizon > 0))
        return fail(PPGPU_EINVAL, "config: speeds, radii, increment and horizon must be positive");
    if (cfg->heuristic < PPGPU_H_MAX_DISTANCE || cfg->heuristic > PPGPU_H_TSP_DUBINS_K)
        return fail(PPGPU_EINVAL, "config: unknown heuristic");
    if ((cfg->heuristic == PPGPU_H_TSP_DUBINS_ALL || cfg->heuristic == PPGPU_H_TSP_DUBINS_K) && !(cfg->heuristic_turning_radius > 0))
        return fail(PPGPU_EINVAL, "config: the Dubins-TSP heuristics need heuristic_turning_radius > 0 "
                                  "(RibbonManager throws \"Cannot compute ribbon dubins distance with unset turning radius\")");
    double steps = cfg->time_horizon / (cfg->collision_checking_increment / cfg->max_speed);
    if (!(steps < 60000.0)) return fail(PPGPU_ECAPACITY, "config: more than 60000 collision-check steps per edge");
    c->cfg = *cfg;
    c->have_cfg = true;
    c->ng = (int)steps + 8;
    c->nverts = 0;  // time grids depend on the config: vertices must be set again
    return PPGPU_OK;
}

int ppgpu_set_grid(ppgpu_ctx* c, const uint8_t* cells, int32_t rows, int32_t cols, double res) {
    if (!c) return fail(PPGPU_EINVAL, "null context");
    HIP_TRY(hipSetDevice(c->device));
    if (rows == 0) { c->rows = c->cols = c->wpr = 0; c->res = 0; return PPGPU_OK; }
    if (!cells || rows < 0 || cols <= 0 || !(res > 0)) return fail(PPGPU_EINVAL, "grid: bad shape or resolution");
    int wpr = (cols + 31) / 32;
    std::vector<uint32_t> bits((size_t)rows * wpr, 0u);
    for (int r = 0; r < rows; r++) {
        const uint8_t* row = cells + (size_t)r * cols;
        uint32_t* brow = bits.data() + (size_t)r * wpr;
        for (int x = 0; x < cols; x++)
            if (row[x]) brow[x >> 5] |= 1u << (x & 31);
    }
    int rc = c->grid.reserve(bits.size(), false, c->stream);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(c->grid.p, bits.data(), bits.size() * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
    // clearance map: chessboard distance to the nearest blocked-or-outside cell, capped (two separable passes on the device)
    if ((rc = c->grid_clear.reserve((size_t)rows * cols, false, c->stream)) || (rc = c->grid_rowclear.reserve((size_t)rows * cols, false, c->stream)))
        return rc;
    {
        const long long ncell = (long long)rows * cols;
        hipLaunchKernelGGL(pp_k_grid_row_clear, dim3((unsigned)((ncell + 255) / 256)), dim3(256), 0, c->stream, c->grid.p, rows, cols, wpr, c->grid_rowclear.p);
        hipLaunchKernelGGL(pp_k_grid_clear, dim3((unsigned)((ncell + 255) / 256)), dim3(256), 0, c->stream, c->grid_rowclear.p, rows, cols, c->grid_clear.p);
        HIP_TRY(hipGetLastError());
    }
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->rows = rows; c->cols = cols; c->wpr = wpr; c->res = res;
    return PPGPU_OK;
}

int ppgpu_set_obstacles(ppgpu_ctx* c, int32_t model, int32_t n, const double* o7) {
    if (!c) return fail(PPGPU_EINVAL, "null context");
    HIP_TRY(hipSetDevice(c->device));
    if (model == PPGPU_OBST_NONE || n == 0) { c->n_obst = 0; c->obst_model = PPGPU_OBST_NONE; return PPGPU_OK; }
    if (model == PPGPU_OBST_GAUSSIAN) return fail(PPGPU_EINVAL, "obstacles: the Gaussian model takes its rows through ppgpu_set_gaussian_obstacles");
    if (model != PPGPU_OBST_BINARY) return fail(PPGPU_EINVAL, "obstacles: unknown model");
    if (n < 0 || !o7) return fail(PPGPU_EINVAL, "obstacles: bad arguments");
    std::vector<PPObst> h((size_t)n);
    for (int i = 0; i < n; i++) {
        const double* o = o7 + 7 * (size_t)i;
        // Obstacle(x, y, heading, speed, time, width, length): Yaw = M_PI_2 - heading
        // (BinaryDynamicObstaclesManager.h:17-19); strict: Width += 2, Length += 2 (.cpp:8-11)
        double yaw = M_PI_2 - o[2];
        h[i].X = o[0]; h[i].Y = o[1];
        h[i].cosYaw = std::cos(yaw); h[i].sinYaw = std::sin(yaw);
        h[i].Speed = o[3]; h[i].Time = o[4];
        h[i].halfW = (o[5] + 2) / 2;
        h[i].halfL = (o[6] + 2) / 2;
        h[i].reach = std::sqrt(h[i].halfW * h[i].halfW + h[i].halfL * h[i].halfL) * (1.0 + 1e-12);
        h[i].pad[0] = h[i].pad[1] = h[i].pad[2] = 0;
    }
    int rc = c->obst.reserve((size_t)n, false, c->stream);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(c->obst.p, h.data(), h.size() * sizeof(PPObst), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->n_obst = n;
    c->obst_model = PPGPU_OBST_BINARY;
    return PPGPU_OK;
}

int ppgpu_set_gaussian_obstacles(ppgpu_ctx* c, int32_t n, const double* rows, int32_t covariance_given) {
    if (!c) return fail(PPGPU_EINVAL, "null context");
    HIP_TRY(hipSetDevice(c->device));
    if (n == 0) { c->n_obst = 0; c->obst_model = PPGPU_OBST_NONE; return PPGPU_OK; }
    if (n < 0 || !rows) return fail(PPGPU_EINVAL, "gaussian obstacles: bad arguments");
    static_assert(sizeof(PPGauss) == sizeof(PPObst), "both obstacle records share one buffer and the culling fields");
    std::vector<PPGauss> h((size_t)n);
    const int stride = covariance_given ? 9 : 5;
    for (int i = 0; i < n; i++) {
        const double* o = rows + (size_t)stride * i;
        // Obstacle(x, y, heading, speed, time[, covariance]): Yaw = M_PI_2 - heading, default covariance
        // [[30,10],[10,30]] (GaussianDynamicObstaclesManager.h:23-29)
        const double yaw = M_PI_2 - o[2];
        double c00 = 30, c01 = 10, c10 = 10, c11 = 30;
        if (covariance_given) { c00 = o[5]; c01 = o[6]; c10 = o[7]; c11 = o[8]; }
        // pdf(): covariance.inverse() and covariance.determinant() of a fixed 2x2 (Eigen's closed forms), norm = 1/2pi/sqrt(det)
        const double det = c00 * c11 - c10 * c01;
        const double invdet = 1.0 / det;
        PPGauss& g = h[i];
        g.X = o[0]; g.Y = o[1]; g.cosYaw = std::cos(yaw); g.sinYaw = std::sin(yaw); g.Speed = o[3]; g.Time = o[4];
        g.i00 = c11 * invdet; g.i10 = -c10 * invdet; g.i01 = -c01 * invdet; g.i11 = c00 * invdet;
        const double twoPi = 2 * M_PI;
        g.norm = 1.0 / twoPi / std::sqrt(det);
        // culling radius: pdf <= |norm| exp(-lmin d^2 / 2), lmin = smallest eigenvalue of the symmetric part of the inverse;
        // beyond reach it is below 1e-13 (collisionExists floors its SUM at 1e-5).  Not positive definite: never culled.
        const double sa = g.i00, sc = g.i11, sb = 0.5 * (g.i01 + g.i10);
        const double lmin = 0.5 * (sa + sc) - std::sqrt(0.25 * (sa - sc) * (sa - sc) + sb * sb);
        if (!(det > 0) || !(lmin > 0) || !std::isfinite(g.norm)) g.reach = 1e300;
        else if (!(std::fabs(g.norm) > 1e-13)) g.reach = 0;
        else g.reach = std::sqrt(2.0 * std::log(std::fabs(g.norm) / 1e-13) / lmin) * (1.0 + 1e-9);
    }
    int rc = c->obst.reserve((size_t)n, false, c->stream);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(c->obst.p, h.data(), h.size() * sizeof(PPGauss), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->n_obst = n;
    c->obst_model = PPGPU_OBST_GAUSSIAN;
    return PPGPU_OK;
}

int ppgpu_set_vertices(ppgpu_ctx* c, int32_t n, const ppgpu_vertex* hv, int32_t n_ribbons, const double* hr) {
    int rc = require_cfg(c);
    if (rc) return rc;
    HIP_TRY(hipSetDevice(c->device));
    if (n <= 0 || !hv || n_ribbons < 0 || (n_ribbons > 0 && !hr)) return fail(PPGPU_EINVAL, "vertices: bad arguments");
    if (n >= (1 << 24)) return fail(PPGPU_ECAPACITY, "vertices: at most 2^24-1 open vertices");
    for (int i = 0; i < n; i++) {
        if (hv[i].ribbon_count < 0 || hv[i].ribbon_offset < 0 || hv[i].ribbon_offset + hv[i].ribbon_count > n_ribbons)
            return fail(PPGPU_EINVAL, "vertices: ribbon range outside the pool");
        if (hv[i].ribbon_count > PP_WAVE) return fail(PPGPU_ECAPACITY, "vertices: more than 64 ribbons on one vertex");
        if (hv[i].time < c->cfg.start_state_time) return fail(PPGPU_EINVAL, "vertices: vertex time before start_state_time");
    }
    if ((rc = c->verts.reserve((size_t)n, false, c->stream))) return rc;
    if ((rc = c->ribbons.reserve((size_t)(n_ribbons > 0 ? n_ribbons : 1) * 4, false, c->stream))) return rc;
    if ((rc = c->tgrid.reserve((size_t)n * c->ng, false, c->stream))) return rc;
    HIP_TRY(hipMemcpyAsync(c->verts.p, hv, (size_t)n * sizeof(ppgpu_vertex), hipMemcpyHostToDevice, c->stream));
    if (n_ribbons > 0)
        HIP_TRY(hipMemcpyAsync(c->ribbons.p, hr, (size_t)n_ribbons * 4 * sizeof(double), hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(pp_k_time_grid, dim3((unsigned)n), dim3(64), 0, c->stream, c->verts.p, n, c->cfg.start_state_time,
                       c->cfg.collision_checking_increment, c->cfg.max_speed, c->ng, c->tgrid.p);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(c->stream));  // the host arrays may go away after return
    c->nverts = n; c->nribbons = n_ribbons;
    c->max_vertex_ribbons = 0;
    for (int i = 0; i < n; i++) if (hv[i].ribbon_count > c->max_vertex_ribbons) c->max_vertex_ribbons = hv[i].ribbon_count;
    return PPGPU_OK;
}

// ------------------------------------------------------------------------------ samples
int ppgpu_set_samples(ppgpu_ctx* c, int64_t n, const double* hx, const double* hy, const double* hh) {
    if (!c) return fail(PPGPU_EINVAL, "null context");
    HIP_TRY(hipSetDevice(c->device));
    if (n < 0 || (n > 0 && (!hx || !hy || !hh))) return fail(PPGPU_EINVAL, "samples: bad arguments");
    if (n >= (1ll << 32)) return fail(PPGPU_ECAPACITY, "samples: at most 2^32-1");
    int rc;
    size_t need = (size_t)(n > 0 ? n : 1);
    if ((rc = c->sx.reserve(need, false, c->stream)) || (rc = c->sy.reserve(need, false, c->stream)) ||
        (rc = c->sh.reserve(need, false, c->stream)))
        return rc;
    if (n > 0) {
        HIP_TRY(hipMemcpyAsync(c->sx.p, hx, (size_t)n * sizeof(double), hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipMemcpyAsync(c->sy.p, hy, (size_t)n * sizeof(double), hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipMemcpyAsync(c->sh.p, hh, (size_t)n * sizeof(double), hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
    }
    c->n_samples = n;
    c->n_extra = 0;
    return PPGPU_OK;
}

int ppgpu_set_extra_targets(ppgpu_ctx* c, int32_t n, const double* hx, const double* hy, const double* hh, int64_t* first_index) {
    if (!c) return fail(PPGPU_EINVAL, "null context");
    HIP_TRY(hipSetDevice(c->device));
    if (n < 0 || (n > 0 && (!hx || !hy || !hh))) return fail(PPGPU_EINVAL, "extra_targets: bad arguments");
    int rc;
    size_t need = (size_t)(c->n_samples + n > 0 ? c->n_samples + n : 1);
    if ((rc = c->sx.reserve(need, true, c->stream)) || (rc = c->sy.reserve(need, true, c->stream)) ||
        (rc = c->sh.reserve(need, true, c->stream)))
        return rc;
    if (n > 0) {
        HIP_TRY(hipMemcpyAsync(c->sx.p + c->n_samples, hx, (size_t)n * sizeof(double), hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipMemcpyAsync(c->sy.p + c->n_samples, hy, (size_t)n * sizeof(double), hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipMemcpyAsync(c->sh.p + c->n_samples, hh, (size_t)n * sizeof(double), hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
    }
    c->n_extra = n;
    if (first_index) *first_index = c->n_samples;
    return PPGPU_OK;
}

int ppgpu_get_samples(ppgpu_ctx* c, int64_t first, int64_t n, double* out5) {
    int rc = require_cfg(c);
    if (rc) return rc;
    HIP_TRY(hipSetDevice(c->device));
    if (first < 0 || n < 0 || first + n > c->n_samples || (n > 0 && !out5)) return fail(PPGPU_EINVAL, "get_samples: range");
    if (n == 0) return PPGPU_OK;
    std::vector<double> x((size_t)n), y((size_t)n), h((size_t)n);
    HIP_TRY(hipMemcpyAsync(x.data(), c->sx.p + first, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipMemcpyAsync(y.data(), c->sy.p + first, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipMemcpyAsync(h.data(), c->sh.p + first, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    for (int64_t i = 0; i < n; i++) {
        out5[5 * i] = x[(size_t)i]; out5[5 * i + 1] = y[(size_t)i]; out5[5 * i + 2] = h[(size_t)i];
        // speed is not stored per sample: expand() overwrites it with maxSpeed before any use
        // (SamplingBasedPlanner.cpp:113), so the store keeps x, y, heading only.
        out5[5 * i + 3] = c->cfg.max_speed;
        out5[5 * i + 4] = 0;   // State(..., 0) (StateGenerator.cpp:16-20)
    }
    return PPGPU_OK;
}

int64_t ppgpu_num_samples(ppgpu_ctx* c) { return c ? c->n_samples : 0; }

int ppgpu_sampler_init(ppgpu_ctx* c, const double* b6, uint64_t seed, int32_t n_ribbons, const double* hr) {
    int rc = require_cfg(c);
    if (rc) return rc;
    HIP_TRY(hipSetDevice(c->device));
    if (!b6) return fail(PPGPU_EINVAL, "sampler: bounds are null");
    if (n_ribbons > 0 && !hr) return fail(PPGPU_EINVAL, "sampler: ribbons are null");
    if (n_ribbons > 1024) return fail(PPGPU_ECAPACITY, "sampler: more than 1024 ribbons");
    PPSamplerState& s = c->sampler;
    for (int i = 0; i < 6; i++) s.b[i] = b6[i];
    // std::linear_congruential_engine::seed: c == 0 and s mod m == 0 -> 1
    unsigned long long x = seed % 2147483647ull;
    s.seed = (unsigned)(x == 0 ? 1 : x);
    s.on_ribbons = n_ribbons >= 0 ? 1 : 0;   // the ribbon constructor sets m_SampleOnRibbons even for an empty manager
    s.n_ribbons = n_ribbons > 0 ? n_ribbons : 0;
    const int kPosGroups = 1024;
    if (!c->samp_pos.p) c->samp_group = -1;
    if ((rc = c->samp_pos.reserve((size_t)8 * kPosGroups, false, c->stream))) return rc;
    c->samp_group = (c->samp_group + 1) % kPosGroups;
    if (c->samp_group == 0)      // (in stream order after everything that used the old groups: no host wait)
        HIP_TRY(hipMemsetAsync(c->samp_pos.p, 0, (size_t)8 * kPosGroups * sizeof(unsigned long long), c->stream));
    s.d_pos = c->samp_pos.p + (size_t)8 * c->samp_group;
    s.initialised = 1;
    if (n_ribbons > 0) {
        // the generator of every iteration of a plan() call is built from the same ribbon manager (AStarPlanner.cpp:34): the table is
        // uploaded (one host round trip) only when it differs from the one already on the device
        const size_t nd = (size_t)n_ribbons * 4;
        if (c->samp_ribbons_host.size() != nd || std::memcmp(c->samp_ribbons_host.data(), hr, nd * sizeof(double)) != 0) {
            if ((rc = c->samp_ribbons.reserve(nd, false, c->stream))) return rc;
            c->samp_ribbons_host.clear();             // not valid until the copy below is known to have happened
            HIP_TRY(hipMemcpyAsync(c->samp_ribbons.p, hr, nd * sizeof(double), hipMemcpyHostToDevice, c->stream));
            HIP_TRY(hipStreamSynchronize(c->stream));
            c->samp_ribbons_host.assign(hr, hr + nd);
        }
    }
    c->n_samples = 0;
    c->n_extra = 0;
    return PPGPU_OK;
}

// Steps 1-3 of the sampler: which stream slots do the next n samples start at?  Leaves qpos[0..n)
// (relative slots) in c->s_u32a, the projection bits in c->s_bytes and the relative slot of sample n
// (= where the stream resumes) in *d_end (device).  Without ribbons the layout is fixed (4 slots per sample).
static int sampler_chain(ppgpu_ctx* c, long long n, long long& nq, int& nblk_q, unsigned long long** d_end, bool skip = false) {
    PPSamplerState& s = c->sampler;
    hipStream_t st = c->stream;
    int rc;
    nq = s.on_ribbons ? (6 * n + 8) : n;   // worst case every sample is projected: 6 slots per sample
    nblk_q = (int)((nq + PP_SCAN_TILE - 1) / PP_SCAN_TILE);
    if (nblk_q > PP_SCAN_TILE) return fail(PPGPU_ECAPACITY, "sampler: batch too large for the two-level scan (max ~690k samples per call)");
    if ((rc = c->s_bytes.reserve((size_t)nq + 64 + (size_t)n + 64, false, st))) return rc;   // projection / visited bits, then the keep flags
    if ((rc = c->s_u64.reserve((size_t)nblk_q + 64, false, st))) return rc;
    if ((rc = c->s_u32a.reserve((size_t)nq + 64, false, st))) return rc;
    {   // per-tile counts of the chain scan, later (ppgpu_sampler_add) per-workgroup counts of the compaction: sized for both now,
        // so that nothing is re-allocated while launches that use it are in flight
        const size_t nblk_n = (size_t)((n + 255) / 256);
        if ((rc = c->s_u32b.reserve(((size_t)nblk_q > nblk_n ? (size_t)nblk_q : nblk_n) + 64, false, st))) return rc;
    }
    *d_end = c->s_u64.p + nblk_q + 8;
    if (!s.on_ribbons) { HIP_TRY(hipMemsetAsync(c->s_u64.p + nblk_q + 8, 0, 16 * sizeof(unsigned long long), st)); return PPGPU_OK; }
    unsigned char* proj = c->s_bytes.p;
    unsigned* tilefn = reinterpret_cast<unsigned*>(c->s_u64.p);      // one 18-bit function per tile (the first nblk_q words' worth of s_u64)
    // 1. proj[q] for every slot in range (LCG jump-ahead) + 2a. the function of every tile of PP_SCAN_TILE slots: the chain
    //    q -> q + 5 + proj[q] as a scan over compositions of functions on its six states (also zeroes the end slot / total of this call)
    hipLaunchKernelGGL(pp_k_proj_reduce, dim3(nblk_q), dim3(256), 0, st, s.seed, s.d_pos, nq, proj, tilefn, c->s_u64.p + nblk_q + 8);
    // 2b. visited[q] + 3a. visited slots per tile
    hipLaunchKernelGGL(pp_k_chain_apply_count, dim3(nblk_q), dim3(256), 0, st, proj, nq, tilefn, c->s_u32b.p);
    // 3b. rank the visited slots: slot of each sample, and of sample n
    hipLaunchKernelGGL(pp_k_chain_positions_scan, dim3(nblk_q), dim3(256), 0, st, proj, nq, c->s_u32b.p, n, c->s_u32a.p, *d_end,
                       skip ? const_cast<unsigned long long*>(s.d_pos) : (unsigned long long*)nullptr);
    HIP_TRY(hipGetLastError());
    return PPGPU_OK;
}

int ppgpu_sampler_skip(ppgpu_ctx* c, int64_t n_attempts) {
    int rc = require_cfg(c);
    if (rc) return rc;
    HIP_TRY(hipSetDevice(c->device));
    if (!c->sampler.initialised) return fail(PPGPU_ESTATE, "ppgpu_sampler_init must be called first");
    if (n_attempts < 0) return fail(PPGPU_EINVAL, "sampler: negative count");
    PPSamplerState& s = c->sampler;
    // Launches only: the position after the skip is data-dependent (which samples were projected onto a ribbon), but it stays on the
    // device (samp_pos); a chain that does not reach the end of its batch raises samp_pos[1], which the next ppgpu_sampler_add reports.
    long long left = n_attempts;
    while (left > 0) {
        long long n = left < 524288 ? left : 524288;
        long long nq = 0; int nblk_q = 0; unsigned long long* d_end = nullptr;
        if (s.on_ribbons) {
            // three launches; the last of them (pp_k_chain_positions_scan) advances the position itself
            if ((rc = sampler_chain(c, n, nq, nblk_q, &d_end, true))) return rc;
        } else {
            hipLaunchKernelGGL(pp_k_sampler_advance, dim3(1), dim3(1), 0, c->stream, const_cast<unsigned long long*>(s.d_pos), d_end, (const unsigned long long*)nullptr,
                               4ull * (unsigned long long)n, 0);
            HIP_TRY(hipGetLastError());
        }
        left -= n;
    }
    return PPGPU_OK;
}

int ppgpu_sampler_add(ppgpu_ctx* c, int64_t n_attempts, int64_t* n_total_out) {
    int rc = require_cfg(c);
    if (rc) return rc;
    HIP_TRY(hipSetDevice(c->device));
    if (!c->sampler.initialised) return fail(PPGPU_ESTATE, "ppgpu_sampler_init must be called first");
    if (n_attempts < 0) return fail(PPGPU_EINVAL, "sampler: negative count");
    if (n_attempts == 0) { if (n_total_out) *n_total_out = c->n_samples; return PPGPU_OK; }
    if (n_attempts > 524288) return fail(PPGPU_ECAPACITY, "sampler: at most 524288 attempts per call");
    const long long n = n_attempts;
    PPSamplerState& s = c->sampler;
    hipStream_t st = c->stream;
    size_t need_samples = (size_t)(c->n_samples + n);
    if ((rc = c->sx.reserve(need_samples, true, st)) || (rc = c->sy.reserve(need_samples, true, st)) ||
        (rc = c->sh.reserve(need_samples, true, st)))
        return rc;
    if ((rc = c->s_cand.reserve((size_t)n * 3, false, st))) return rc;
    long long nq; int nblk_q; unsigned long long* d_end;
    if ((rc = sampler_chain(c, n, nq, nblk_q, &d_end))) return rc;
    const int nblk_n = (int)((n + 255) / 256);
    unsigned char* proj = c->s_bytes.p;
    unsigned* qpos = c->s_u32a.p;
    unsigned* blk32 = c->s_u32b.p;
    // 4. generate the n candidate states (thread per sample) + 5a. SamplingBasedPlanner::addSamples' map filter
    PPGrid g{c->grid.p, c->rows, c->cols, c->wpr, c->res, c->res > 0 ? 1.0 / c->res : 0.0, nullptr};
    unsigned char* keep = c->s_bytes.p + (size_t)nq + 64;          // (the projection bits are still being read by this launch)
    hipLaunchKernelGGL(pp_k_generate_keep, dim3((unsigned)nblk_n), dim3(256), 0, st, s, s.on_ribbons ? qpos : nullptr, s.on_ribbons ? proj : nullptr,
                       c->samp_ribbons.p, n, c->s_cand.p, g, keep, blk32);
    // 5b. order-preserving compaction into the store
    unsigned long long* d_total = d_end + 8;
    hipLaunchKernelGGL(pp_k_compact_scan, dim3((unsigned)nblk_n), dim3(256), 0, st, keep, n, blk32, c->s_cand.p, c->sx.p, c->sy.p,
                       c->sh.p, c->n_samples, d_total);
    HIP_TRY(hipGetLastError());
    hipLaunchKernelGGL(pp_k_sampler_advance, dim3(1), dim3(1), 0, st, const_cast<unsigned long long*>(s.d_pos), d_end, d_total, 4ull * (unsigned long long)n, s.on_ribbons);
    HIP_TRY(hipGetLastError());
    // {chain error, kept, slots consumed} come back in one copy through pinned memory (a copy to pageable memory waits for the
    // stream by itself); the position itself stays on the device
    volatile unsigned long long* h2 = c->pinned_counts;
    h2[0] = 0; h2[1] = 0; h2[2] = 0;
    HIP_TRY(hipMemcpyAsync((void*)&h2[0], s.d_pos + 1, 3 * sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    if (h2[0] != 0) return fail(PPGPU_EHIP, "sampler: chain scan did not reach the end of the batch (this call or a ppgpu_sampler_skip before it)");
    c->n_samples += (long long)h2[1];
    c->n_extra = 0;                        // the appended samples overwrote any explicit targets
    if (n_total_out) *n_total_out = c->n_samples;
    return PPGPU_OK;
}

// ------------------------------------------------------------------------------ edge generation
static void fill_params(ppgpu_ctx* c, PPParams& p) {
    const ppgpu_config& g = c->cfg;
    p.max_speed = g.max_speed;
    p.slow_speed = g.slow_speed <= 0 ? g.max_speed : g.slow_speed;   // PlannerConfig::slowSpeed()
    p.rho = g.turning_radius; p.rho_cov = g.coverage_turning_radius;
    p.horizon = g.time_horizon; p.tmin = g.time_minimum; p.inc_d = g.collision_checking_increment;
    p.sst = g.start_state_time; p.ribw = g.ribbon_width;
    p.inv_inc_d = 1.0 / p.inc_d;
    p.cpf = g.collision_penalty_factor; p.tpf = g.time_penalty_factor;
    p.heuristic = g.heuristic; p.tsp_k = g.tsp_k; p.h_rho = g.heuristic_turning_radius;
    p.fuse_h = 0;
    p.lane_split = 0; p.defer_h = 0; p.defer_list = nullptr; p.defer_count = nullptr; p.quiet_finish = 0; p.live_list = nullptr; p.live_count = nullptr; p.cover_state = nullptr; p.hw_list = nullptr; p.hw_count = nullptr;
    p.grid = PPGrid{c->grid.p, c->rows, c->cols, c->wpr, c->res, c->res > 0 ? 1.0 / c->res : 0.0, c->rows > 0 ? c->grid_clear.p : nullptr};
    p.obst = c->obst.p; p.n_obst = c->n_obst; p.obst_model = c->obst_model;
    p.verts = c->verts.p; p.ribbons = c->ribbons.p; p.tgrid = c->tgrid.p; p.ng = c->ng; p.nverts = c->nverts;
    p.sx = c->sx.p; p.sy = c->sy.p; p.sh = c->sh.p; p.n_samples = c->n_samples + c->n_extra;
}

static int require_world(ppgpu_ctx* c) {
    int rc = require_cfg(c);
    if (rc) return rc;
    if (c->nverts <= 0) return fail(PPGPU_ESTATE, "ppgpu_set_vertices must be called (after ppgpu_set_config)");
    if (c->n_samples + c->n_extra <= 0) return fail(PPGPU_ESTATE, "no targets: call ppgpu_sampler_add, ppgpu_set_samples or ppgpu_set_extra_targets");
    return PPGPU_OK;
}

int ppgpu_dubins_lengths(ppgpu_ctx* c, int32_t v0, int32_t nv, double* d_lengths) {
    int rc = require_world(c);
    if (rc) return rc;
    HIP_TRY(hipSetDevice(c->device));
    if (v0 < 0 || nv <= 0 || v0 + nv > c->nverts || !d_lengths) return fail(PPGPU_EINVAL, "dubins_lengths: vertex range");
    if (nv > 65535) return fail(PPGPU_ECAPACITY, "dubins_lengths: at most 65535 vertices per call");
    const long long ns = c->n_samples;
    if (ns <= 0) return fail(PPGPU_ESTATE, "dubins_lengths: the sample store is empty");
    hipLaunchKernelGGL(pp_k_dubins_lengths, dim3((unsigned)((ns + 255) / 256), (unsigned)nv), dim3(256), 0, c->stream, c->verts.p,
                       v0, c->sx.p, c->sy.p, c->sh.p, ns, c->cfg.turning_radius, c->cfg.coverage_turning_radius,
                       c->cfg.collision_checking_increment, d_lengths);
    HIP_TRY(hipGetLastError());
    return PPGPU_OK;
}

int ppgpu_select_nearest(ppgpu_ctx* c, int32_t v0, int32_t nv, int32_t k, int32_t* h_idx, double* h_len) {
    int rc = require_world(c);
    if (rc) return rc;
    HIP_TRY(hipSetDevice(c->device));
    if (k <= 0 || !h_idx || !h_len) return fail(PPGPU_EINVAL, "select_nearest: bad arguments");
    const long long ns = c->n_samples;
    if ((rc = c->tmp_lengths.reserve((size_t)nv * ns * 2, false, c->stream))) return rc;
    if ((rc = ppgpu_dubins_lengths(c, v0, nv, c->tmp_lengths.p))) return rc;
    size_t nout = (size_t)nv * 2 * k;
    if ((rc = c->tmp_idx.reserve(nout, false, c->stream)) || (rc = c->tmp_len_out.reserve(nout, false, c->stream))) return rc;
    hipLaunchKernelGGL(pp_k_select_nearest, dim3((unsigned)(nv * 2)), dim3(256), 0, c->stream, c->tmp_lengths.p, ns, k,
                       c->tmp_idx.p, c->tmp_len_out.p);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(h_idx, c->tmp_idx.p, nout * sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipMemcpyAsync(h_len, c->tmp_len_out.p, nout * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return PPGPU_OK;
}

// The k winners of every (vertex, radius) of vertices [0, nv) in the reference's push order -> c->ord_idx: probe bound, near list,
// lengths of the near samples + block minima (-> c->tmp_lengths, one double2 per near slot), bound, candidate lists, sort + replay
// (pp_k_expand.h).
// Asynchronous; *c->ord_fallbacks.p counts lists that kept ascending length.
static int launch_expand_order(ppgpu_ctx* c, int nv, int k, bool zero_fallbacks = true) {
    const long long ns = c->n_samples;
    long long cap = 64;
    while (cap < ns && cap < 65536) cap <<= 1;         // candidates per list; pp_k_expand_order filters them down to at most PP_ORD_CAP
    const int nblk = (int)((ns + 255) / 256);
    int rc;
    if ((rc = c->tmp_lengths.reserve((size_t)nv * ns * 2, false, c->stream)) ||
        (rc = c->ord_key.reserve((size_t)nv * 2 * cap, false, c->stream)) || (rc = c->ord_val.reserve((size_t)nv * 2 * cap, false, c->stream)) ||
        (rc = c->ord_len.reserve((size_t)nv * 2 * cap, false, c->stream)) || (rc = c->ord_idx.reserve((size_t)nv * 2 * k, false, c->stream)) ||
        (rc = c->ord_fallbacks.reserve(1, false, c->stream)) || (rc = c->ord_blockmin.reserve((size_t)nv * nblk * 2 * (256 / PP_NEAR_GROUP), false, c->stream)) ||
        (rc = c->ord_blockcnt.reserve((size_t)nv * nblk * (256 / PP_NEAR_GROUP), false, c->stream)) || (rc = c->ord_bound.reserve((size_t)nv * 2, false, c->stream)) ||
        (rc = c->ord_count.reserve((size_t)nv * 2, false, c->stream)) || (rc = c->near_idx.reserve((size_t)nv * ns, false, c->stream)) ||
        (rc = c->near_count.reserve((size_t)nv, false, c->stream)) || (rc = c->probe_bound.reserve((size_t)nv * 2 * PP_PROBE_GROUPS, false, c->stream)))
        return rc;
    if (zero_fallbacks) HIP_TRY(hipMemsetAsync(c->ord_fallbacks.p, 0, sizeof(unsigned), c->stream));      // (ppgpu_expand_host: its unpack kernel does it)
    const int two_radii = (c->cfg.coverage_turning_radius != c->cfg.turning_radius) ? 1 : 0;
    hipLaunchKernelGGL(pp_k_expand_probe, dim3((unsigned)(PP_PROBE / 256), (unsigned)nv), dim3(256), 0, c->stream, c->verts.p, c->sx.p, c->sy.p, c->sh.p, ns,
                       c->cfg.turning_radius, c->cfg.coverage_turning_radius, c->cfg.collision_checking_increment, two_radii, c->probe_bound.p, c->near_count.p);
    hipLaunchKernelGGL(pp_k_expand_near, dim3((unsigned)((ns + 256 * PP_NEAR_PER - 1) / (256 * PP_NEAR_PER)), (unsigned)nv), dim3(256), 0, c->stream, c->verts.p, c->sx.p, c->sy.p, ns,
                       c->cfg.collision_checking_increment, c->probe_bound.p, k, c->near_idx.p, c->near_count.p);
    hipLaunchKernelGGL(pp_k_near_lengths, dim3((unsigned)nblk, (unsigned)nv), dim3(256), 0, c->stream, c->verts.p, c->sx.p, c->sy.p, c->sh.p, ns,
                       c->near_idx.p, c->near_count.p, c->cfg.turning_radius, c->cfg.coverage_turning_radius, two_radii, c->tmp_lengths.p,
                       c->ord_blockmin.p, c->ord_blockcnt.p);
    hipLaunchKernelGGL(pp_k_expand_bound, dim3((unsigned)(nv * 2)), dim3(256), 0, c->stream, c->ord_blockmin.p, c->ord_blockcnt.p, nblk * (256 / PP_NEAR_GROUP), c->near_count.p, k,
                       c->ord_bound.p, c->ord_count.p);
    hipLaunchKernelGGL(pp_k_expand_candidates, dim3((unsigned)nblk, (unsigned)nv), dim3(256), 0, c->stream, c->tmp_lengths.p, c->verts.p, c->sx.p,
                       c->sy.p, ns, c->near_idx.p, c->near_count.p, two_radii, c->ord_bound.p, c->ord_key.p, c->ord_val.p, c->ord_len.p, cap, c->ord_count.p);
    hipLaunchKernelGGL(pp_k_expand_order, dim3((unsigned)(nv * 2)), dim3(256), 0, c->stream, ns, k, c->cfg.max_speed, c->cfg.time_penalty_factor,
                       two_radii, c->ord_bound.p, c->ord_key.p, c->ord_val.p, c->ord_len.p, cap, c->ord_count.p, c->tmp_lengths.p, c->near_idx.p,
                       c->near_count.p, c->ord_idx.p, c->ord_fallbacks.p);
    HIP_TRY(hipGetLastError());
    return PPGPU_OK;
}

int ppgpu_expand_order(ppgpu_ctx* c, int32_t v0, int32_t nv, int32_t k, int32_t* h_idx, uint32_t* h_fallbacks) {
    int rc = require_world(c);
    if (rc) return rc;
    HIP_TRY(hipSetDevice(c->device));
    if (k <= 0 || !h_idx || v0 != 0 || nv <= 0 || nv > c->nverts) return fail(PPGPU_EINVAL, "expand_order: bad arguments (v0 must be 0)");
    if (nv > 65535) return fail(PPGPU_ECAPACITY, "expand_order: at most 65535 vertices per call");
    if (c->n_samples <= 0) return fail(PPGPU_ESTATE, "expand_order: the sample store is empty");
    const size_t nout = (size_t)nv * 2 * k;
    if ((rc = launch_expand_order(c, nv, k))) return rc;
    unsigned fb = 0;
    HIP_TRY(hipMemcpyAsync(h_idx, c->ord_idx.p, nout * sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipMemcpyAsync(&fb, c->ord_fallbacks.p, sizeof(unsigned), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->order_fallbacks += fb;
    if (h_fallbacks) *h_fallbacks = fb;
    return PPGPU_OK;
}

uint64_t ppgpu_order_fallbacks(ppgpu_ctx* c) { return c ? c->order_fallbacks : 0; }

// ------------------------------------------------------------------------------ edge costing
int64_t ppgpu_dense_edge_count(int32_t nv, int64_t ns, uint32_t cfg_mask) {
    return (int64_t)nv * ns * (int64_t)__builtin_popcount(cfg_mask & 0xFu);
}

// Grid of a per-edge kernel: as many workgroups as the device keeps resident (its waves pull edges from the queue), or fewer
// when the launch has fewer edges than that.
static unsigned resident_grid(ppgpu_ctx* c, int slot, void (*kernel)(PPParams), long long n_edges) {
    const int bit = slot < 2 ? 1 : (slot < 4 ? 2 : 4);
    const int wpb = slot < 4 ? PP_WPB : PP_H_WPB;
    if (!((PP_QUEUE_MASK) & bit)) return (unsigned)((n_edges + wpb - 1) / wpb);   // this kernel takes one edge per wave
    if (c->resident[slot] == 0) {
        int per_cu = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)kernel, wpb * 64, 0) != hipSuccess || per_cu <= 0) per_cu = 2;
        c->resident[slot] = per_cu * (c->n_cu > 0 ? c->n_cu : 256);
    }
    const long long need = (n_edges + wpb - 1) / wpb;
    return (unsigned)(need < c->resident[slot] ? need : c->resident[slot]);
}

static int launch_cost(ppgpu_ctx* c, PPParams& p) {
    if (p.n_edges <= 0) return PPGPU_OK;
    const long long total = p.n_edges;
    p.total_edges = total;
    if ((total + PP_WPB - 1) / PP_WPB > 0x3fffffffll) return fail(PPGPU_ECAPACITY, "cost_edges: too many edges for one launch");
    if (!p.child) {
        // the heuristic kernel reads the child ribbon lists: keep them in a scratch the caller never sees
        int stride = c->max_vertex_ribbons + 8;
        if (stride > PP_WAVE) stride = PP_WAVE;
        int rc = c->int_child.reserve((size_t)total * stride * 4, false, c->stream);
        if (rc) return rc;
        p.child = c->int_child.p;
        p.stride = stride;
    }
    p.nch = (p.ng + PP_WAVE - 1) / PP_WAVE;
    if (p.nch < 1) p.nch = 1;
    p.ngp = p.nch * PP_WAVE;
    const size_t per_edge = sizeof(PPEdgeSetup) + (size_t)p.ngp * sizeof(unsigned short) +
                            (size_t)p.nch * (sizeof(unsigned long long) + sizeof(unsigned)) + sizeof(PPTrackSummary) +
                            ((p.n_obst > 0 && p.obst_model == PPGPU_OBST_GAUSSIAN) ? (size_t)(p.ngp + p.nch) * sizeof(double) : 0);
    long long slice = (long long)(c->slice_bytes / per_edge);
    if (slice < PP_WPB) slice = PP_WPB;
    if (slice > total) slice = total;
    const size_t ws = (size_t)slice;
    {
        int rc;
        if ((rc = c->setup.reserve(ws, false, c->stream)) ||
            (rc = c->track_hits.reserve(ws * p.ngp, false, c->stream)) ||
            (rc = c->track_eq.reserve(ws * p.nch, false, c->stream)) ||
            (rc = c->track_chunk_hits.reserve(ws * p.nch, false, c->stream)) ||
            (rc = c->track_summary.reserve(ws, false, c->stream)) || (rc = c->track_far.reserve(ws, false, c->stream)) ||
            (rc = c->track_skip.reserve(ws * p.nch, false, c->stream)))
            return rc;
        if (p.n_obst > 0 && p.obst_model == PPGPU_OBST_GAUSSIAN &&
            ((rc = c->track_pen.reserve(ws * p.ngp, false, c->stream)) || (rc = c->track_chunk_pen.reserve(ws * p.nch, false, c->stream))))
            return rc;
    }
    p.setup = c->setup.p; p.track_hits = c->track_hits.p; p.track_eq = c->track_eq.p;
    p.track_chunk_hits = c->track_chunk_hits.p; p.track_summary = c->track_summary.p; p.track_far = c->track_far.p;   // (cleared below for small launches)
    // chunk skipping needs the clearance map (or no grid at all) and solved curves (a given curve may start late or end early)
    // both prepasses pay for themselves on large launches only: a planner round trip of a few hundred edges is latency-bound
    const bool big = total >= c->prepass_min_edges;
    p.track_skip = (big && !p.wedges && (c->rows == 0 || c->grid_clear.p) && p.ng >= PP_WAVE) ? c->track_skip.p : nullptr;
    if (!big) p.track_far = nullptr;
    p.track_pen = c->track_pen.p; p.track_chunk_pen = c->track_chunk_pen.p;
    {
        int rc = c->need_big.reserve(32, false, c->stream);
        if (rc) return rc;
        p.need_big = c->need_big.p;   // cleared by the first slice's pp_k_solve_edges
        if ((rc = c->work.reserve(PP_WORK_WORDS, false, c->stream))) return rc;
        p.work = c->work.p;
    }
    const bool dubinsH = p.heuristic == PPGPU_H_TSP_DUBINS_ALL || p.heuristic == PPGPU_H_TSP_DUBINS_K;
    const bool gaussianSweep = p.n_obst > 0 && p.obst_model == PPGPU_OBST_GAUSSIAN;
    p.fuse_h = (PP_FUSE_HEUR && !dubinsH && !gaussianSweep) ? 1 : 0;
    p.defer_h = (p.fuse_h && big && PP_LANE_HEUR && c->lane_heuristic && total < (1ll << 32) &&
                 (p.heuristic == PPGPU_H_TSP_POINT_ALL || p.heuristic == PPGPU_H_TSP_POINT_K)) ? 1 : 0;
    p.quiet_finish = (p.track_far && c->quiet_finish) ? 1 : 0;
    p.lane_split = (p.track_far && c->lane_split && !gaussianSweep) ? 1 : 0;
    if (p.track_far && total < (1ll << 32)) {
        int rc = c->live_list.reserve((size_t)slice * 2, false, c->stream);
        if (rc) return rc;
        p.live_list = c->live_list.p; p.live_count = c->need_big.p + 12;
    }
#if defined(PP_DBG_COUNTS)
    const bool laneFinishBuilt = false;     // (those builds keep every edge with its wave: pp_cover_sweep_edge)
#else
    const bool laneFinishBuilt = true;
#endif
    if (laneFinishBuilt && p.live_list && c->lane_finish && !gaussianSweep) {
        // phase C of the edges the cover sweep's waves visit: one lane per edge (pp_k_cover_finish)
        int rc;
        if ((rc = c->cover_state.reserve((size_t)slice, false, c->stream)) || (rc = c->hw_list.reserve((size_t)total, false, c->stream))) return rc;
        p.cover_state = c->cover_state.p; p.hw_list = c->hw_list.p; p.hw_count = c->need_big.p + 13;
    }
    if (p.defer_h) {
        int rc = c->defer_list.reserve((size_t)total * PP_HL_MAX_N, false, c->stream);
        if (rc) return rc;
        p.defer_list = c->defer_list.p; p.defer_count = c->need_big.p + 16;     // [n] = deferred edges with n ribbons, n = 1 .. PP_HL_MAX_N
    }
    if (c->timing) {                                  // the next set of the ring
        c->ev_slot = (int)(c->ev_launches % PP_TIMING_RING);
        c->ev = c->ev_ring[c->ev_slot];
        c->ms_earlier_slices = c->ms_ring[c->ev_slot];
    }
    c->ms_earlier_slices[0] = c->ms_earlier_slices[1] = c->ms_earlier_slices[2] = c->ms_earlier_slices[3] = 0;
    for (long long e0 = 0; e0 < total; e0 += slice) {
        p.e_base = e0; p.ws_base = 0; p.n_edges = (total - e0 < slice) ? (total - e0) : slice;
        if (e0 == 0) c->live_earlier_slices = 0;
        if (c->timing && e0 > 0) {
            // a sliced launch re-uses the events: bank the previous slice's three durations first (timing is a measurement aid: the
            // wait costs the overlap between slices, nothing else)
            HIP_TRY(hipEventSynchronize(c->ev[3]));
            for (int i = 0; i < 3; i++) {
                float ms = 0;
                HIP_TRY(hipEventElapsedTime(&ms, c->ev[i == 2 ? 5 : i], c->ev[i + 1]));
                c->ms_earlier_slices[i] += ms;
            }
            float msa = 0;
            HIP_TRY(hipEventElapsedTime(&msa, c->ev[2], c->ev[5]));
            c->ms_earlier_slices[3] += msa;
            if (p.live_list) { unsigned live = 0; HIP_TRY(hipMemcpy(&live, c->need_big.p + 12, sizeof(unsigned), hipMemcpyDeviceToHost)); c->live_earlier_slices += live; }
        }
        if (c->timing) HIP_TRY(hipEventRecord(c->ev[0], c->stream));
        hipLaunchKernelGGL(pp_k_solve_edges, dim3((unsigned)((p.n_edges + 255) / 256)), dim3(256), 0, c->stream, p);
        if (c->timing) HIP_TRY(hipEventRecord(c->ev[1], c->stream));
        if (p.track_skip) {                                          // timed with the pose sweep
            // one workgroup per epw consecutive edges: as many as give it 256 (edge, chunk) threads
            int epw = 256 / p.nch;
            if (epw < 1) epw = 1;
            if (epw > PP_PLAN_EDGES_MAX) epw = PP_PLAN_EDGES_MAX;
            const unsigned blocks = (unsigned)((p.n_edges + epw - 1) / epw);
            const bool gauss = p.n_obst > 0 && p.obst_model == PPGPU_OBST_GAUSSIAN, many = p.n_obst > PP_WAVE;
            void (*planner)(PPParams, int) = gauss ? (many ? pp_k_plan_skips_gaussian_many : pp_k_plan_skips_gaussian) : (many ? pp_k_plan_skips_many : pp_k_plan_skips);
            hipLaunchKernelGGL(planner, dim3(blocks, (unsigned)((epw * p.nch + 255) / 256)), dim3(256), 0, c->stream, p, epw);
        }
        if (p.n_obst > 0 && p.obst_model == PPGPU_OBST_GAUSSIAN)
            hipLaunchKernelGGL(pp_k_pose_sweep_gaussian, dim3(resident_grid(c, 0, pp_k_pose_sweep_gaussian, p.n_edges)), dim3(PP_WPB * 64), 0, c->stream, p);
        else
            hipLaunchKernelGGL(pp_k_pose_sweep, dim3(resident_grid(c, 1, pp_k_pose_sweep, p.n_edges)), dim3(PP_WPB * 64), 0, c->stream, p);
        if (c->timing) HIP_TRY(hipEventRecord(c->ev[2], c->stream));
        if (p.track_far) hipLaunchKernelGGL(pp_k_approach_events, dim3((unsigned)((p.n_edges + PP_APPROACH_THREADS - 1) / PP_APPROACH_THREADS)), dim3(PP_APPROACH_THREADS), 0, c->stream, p);
        if (c->timing) HIP_TRY(hipEventRecord(c->ev[5], c->stream));
        if (p.n_obst > 0 && p.obst_model == PPGPU_OBST_GAUSSIAN)
            hipLaunchKernelGGL(pp_k_cover_sweep_gaussian, dim3(resident_grid(c, 2, pp_k_cover_sweep_gaussian, p.n_edges)), dim3(PP_WPB * 64), 0, c->stream, p);
        else
            hipLaunchKernelGGL(pp_k_cover_sweep, dim3(resident_grid(c, 3, pp_k_cover_sweep, p.n_edges)), dim3(PP_WPB * 64), 0, c->stream, p);
        if (p.cover_state)       // (the list's length is on the device: a grid for "every edge of the slice is on it", whose spare workgroups leave at once)
            hipLaunchKernelGGL(pp_k_cover_finish, dim3((unsigned)((p.n_edges + PP_FINISH_THREADS - 1) / PP_FINISH_THREADS)), dim3(PP_FINISH_THREADS), 0, c->stream, p);
        if (c->timing) HIP_TRY(hipEventRecord(c->ev[3], c->stream));
    }
    p.e_base = 0;
    p.n_edges = total;
    if (p.heuristic == PPGPU_H_TSP_DUBINS_ALL || p.heuristic == PPGPU_H_TSP_DUBINS_K)
        hipLaunchKernelGGL(pp_k_heuristic_dubins, dim3(resident_grid(c, 4, pp_k_heuristic_dubins, total)), dim3(PP_H_WPB * 64), 0, c->stream, p);
    else if (!p.fuse_h)
        hipLaunchKernelGGL(pp_k_heuristic, dim3(resident_grid(c, 5, pp_k_heuristic, total)), dim3(PP_H_WPB * 64), 0, c->stream, p);
    // The edges whose TSP enumeration the sweeps deferred: packed into one list per ribbon count, then a few lanes each
    // (pp_k_heuristic_lanes).  The rare edge pp_k_cover_finish left to a whole wave (pp_k_heuristic_listed: 7 or 8 child ribbons,
    // ~1 100 of config 3's 236 140 edges, each a single wave's work for ~170 us) runs on a second stream BESIDE it: the lane kernel
    // fills the rest of the machine meanwhile, and the two touch different records.  The main stream waits for the side stream
    // before the launch is over.
    const bool forked = p.cover_state && p.fuse_h && p.heuristic != PPGPU_H_MAX_DISTANCE;
    if (forked) {
        HIP_TRY(hipEventRecord(c->ev_fork, c->stream));
        HIP_TRY(hipStreamWaitEvent(c->side_stream, c->ev_fork, 0));
        // (as many workgroups as pp_k_heuristic keeps resident: same footprint; all but a few leave at once)
        hipLaunchKernelGGL(pp_k_heuristic_listed, dim3(resident_grid(c, 5, pp_k_heuristic, total)), dim3(PP_H_WPB * 64), 0, c->side_stream, p);
        HIP_TRY(hipEventRecord(c->ev_join, c->side_stream));
    }
    if (p.defer_h)
        hipLaunchKernelGGL(pp_k_deferred_list, dim3((unsigned)((total + 256 * PP_DL_PER - 1) / (256 * PP_DL_PER))), dim3(256), 0, c->stream, p);
    if (p.defer_h)
        hipLaunchKernelGGL(pp_k_heuristic_lanes, dim3((unsigned)((total * PP_HL_SPLIT + PP_HL_THREADS - 1) / PP_HL_THREADS) + PP_HL_MAX_N), dim3(PP_HL_THREADS), 0, c->stream, p);
    // child lists of 9..12 ribbons under the K variant: a second pass that touches only those edges (the others cost it one
    // 8-byte read each)
    if (p.heuristic == PPGPU_H_TSP_POINT_K) {
        const long long need = total;                 // (a workgroup per edge, striding)
        hipLaunchKernelGGL(pp_k_heuristic_big, dim3((unsigned)(need < PP_BIG_GRID ? need : PP_BIG_GRID)), dim3(PP_BIG_WPB * 64), 0, c->stream, p);
    }
    if (forked) HIP_TRY(hipStreamWaitEvent(c->stream, c->ev_join, 0));
    c->last_launch_edges = total; c->last_launch_packed = p.live_list != nullptr;
    if (c->timing) { HIP_TRY(hipEventRecord(c->ev[4], c->stream)); c->ev_launches++; }
    HIP_TRY(hipGetLastError());
    return PPGPU_OK;
}

int ppgpu_cost_edges_dense(ppgpu_ctx* c, int32_t v0, int32_t nv, int64_t s0, int64_t ns, uint32_t cfg_mask,
                           ppgpu_edge_result* d_results, double* d_child, int32_t stride) {
    int rc = require_world(c);
    if (rc) return rc;
    HIP_TRY(hipSetDevice(c->device));
    cfg_mask &= 0xFu;
    if (v0 < 0 || nv <= 0 || v0 + nv > c->nverts) return fail(PPGPU_EINVAL, "cost_edges_dense: vertex range");
    if (s0 < 0 || ns <= 0 || s0 + ns > c->n_samples) return fail(PPGPU_EINVAL, "cost_edges_dense: sample range");
    if (!cfg_mask || !d_results) return fail(PPGPU_EINVAL, "cost_edges_dense: empty configuration mask or null results");
    if (d_child && stride <= 0) return fail(PPGPU_EINVAL, "cost_edges_dense: ribbon_stride must be positive");
    PPParams p;
    fill_params(c, p);
    p.edges = nullptr; p.wedges = nullptr;
    p.v0 = v0; p.nv = nv; p.s0 = s0; p.ns = ns; p.cfg_mask = cfg_mask; p.per = __builtin_popcount(cfg_mask);
    p.n_edges = (long long)nv * ns * p.per;
    p.out = d_results; p.child = d_child; p.stride = stride;
    return launch_cost(c, p);
}

int ppgpu_cost_edges_list(ppgpu_ctx* c, int64_t n, const uint64_t* d_edges, ppgpu_edge_result* d_results, double* d_child,
                          int32_t stride) {
    int rc = require_world(c);
    if (rc) return rc;
    HIP_TRY(hipSetDevice(c->device));
    if (n < 0 || (n > 0 && (!d_edges || !d_results))) return fail(PPGPU_EINVAL, "cost_edges_list: bad arguments");
    if (d_child && stride <= 0) return fail(PPGPU_EINVAL, "cost_edges_list: ribbon_stride must be positive");
    PPParams p;
    fill_params(c, p);
    p.edges = (const unsigned long long*)d_edges; p.wedges = nullptr;
    p.v0 = 0; p.nv = 0; p.s0 = 0; p.ns = 1; p.cfg_mask = 0; p.per = 1;
    p.n_edges = n;
    p.out = d_results; p.child = d_child; p.stride = stride;
    return launch_cost(c, p);
}

int ppgpu_cost_edges_host(ppgpu_ctx* c, int64_t n, const uint64_t* h_edges, ppgpu_edge_result* h_results, double* h_child,
                          int32_t stride) {
    int rc = require_world(c);
    if (rc) return rc;
    HIP_TRY(hipSetDevice(c->device));
    if (n < 0 || (n > 0 && (!h_edges || !h_results))) return fail(PPGPU_EINVAL, "cost_edges_host: bad arguments");
    if (n == 0) return PPGPU_OK;
    if (h_child && stride <= 0) return fail(PPGPU_EINVAL, "cost_edges_host: ribbon_stride must be positive");
    if ((rc = c->tmp_edges.reserve((size_t)n, false, c->stream))) return rc;
    if ((rc = c->tmp_results.reserve((size_t)n, false, c->stream))) return rc;
    if (h_child && (rc = c->tmp_child.reserve((size_t)n * stride * 4, false, c->stream))) return rc;
    HIP_TRY(hipMemcpyAsync(c->tmp_edges.p, h_edges, (size_t)n * sizeof(uint64_t), hipMemcpyHostToDevice, c->stream));
    if (h_child) HIP_TRY(hipMemsetAsync(c->tmp_child.p, 0, (size_t)n * stride * 4 * sizeof(double), c->stream));
    if ((rc = ppgpu_cost_edges_list(c, n, (const uint64_t*)c->tmp_edges.p, c->tmp_results.p, h_child ? c->tmp_child.p : nullptr, stride)))
        return rc;
    HIP_TRY(hipMemcpyAsync(h_results, c->tmp_results.p, (size_t)n * sizeof(ppgpu_edge_result), hipMemcpyDeviceToHost, c->stream));
    if (h_child)
        HIP_TRY(hipMemcpyAsync(h_child, c->tmp_child.p, (size_t)n * stride * 4 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return PPGPU_OK;
}

int ppgpu_cost_wrapper_edges_host(ppgpu_ctx* c, int64_t n, const ppgpu_wrapper_edge* h_edges, ppgpu_edge_result* h_results,
                                  double* h_child, int32_t stride) {
    int rc = require_cfg(c);
    if (rc) return rc;
    if (c->nverts <= 0) return fail(PPGPU_ESTATE, "ppgpu_set_vertices must be called (after ppgpu_set_config)");
    HIP_TRY(hipSetDevice(c->device));
    if (n < 0 || (n > 0 && (!h_edges || !h_results))) return fail(PPGPU_EINVAL, "cost_wrapper_edges_host: bad arguments");
    if (n == 0) return PPGPU_OK;
    if (h_child && stride <= 0) return fail(PPGPU_EINVAL, "cost_wrapper_edges_host: ribbon_stride must be positive");
    for (int64_t i = 0; i < n; i++) {
        if (!(h_edges[i].rho > 0) || !(h_edges[i].speed > 0)) return fail(PPGPU_EINVAL, "cost_wrapper_edges_host: rho and speed must be positive");
        if (h_edges[i].vertex < 0 || h_edges[i].vertex >= c->nverts) return fail(PPGPU_EINVAL, "cost_wrapper_edges_host: vertex out of range");
    }
    if ((rc = c->tmp_wedges.reserve((size_t)n, false, c->stream))) return rc;
    if ((rc = c->tmp_results.reserve((size_t)n, false, c->stream))) return rc;
    if (h_child && (rc = c->tmp_child.reserve((size_t)n * stride * 4, false, c->stream))) return rc;
    HIP_TRY(hipMemcpyAsync(c->tmp_wedges.p, h_edges, (size_t)n * sizeof(ppgpu_wrapper_edge), hipMemcpyHostToDevice, c->stream));
    if (h_child) HIP_TRY(hipMemsetAsync(c->tmp_child.p, 0, (size_t)n * stride * 4 * sizeof(double), c->stream));
    PPParams p;
    fill_params(c, p);
    p.edges = nullptr; p.wedges = c->tmp_wedges.p;
    p.v0 = 0; p.nv = 0; p.s0 = 0; p.ns = 1; p.cfg_mask = 0; p.per = 1;
    p.n_edges = n;
    p.out = c->tmp_results.p; p.child = h_child ? c->tmp_child.p : nullptr; p.stride = stride;
    if ((rc = launch_cost(c, p))) return rc;
    HIP_TRY(hipMemcpyAsync(h_results, c->tmp_results.p, (size_t)n * sizeof(ppgpu_edge_result), hipMemcpyDeviceToHost, c->stream));
    if (h_child)
        HIP_TRY(hipMemcpyAsync(h_child, c->tmp_child.p, (size_t)n * stride * 4 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return PPGPU_OK;
}

// ------------------------------------------------------------------------------ heuristic on its own
int ppgpu_heuristic_host(ppgpu_ctx* c, int32_t n, const double* poses3, const int32_t* counts, const double* ribbons, double* h_out, uint32_t* h_flags) {
    int rc = require_cfg(c);
    if (rc) return rc;
    HIP_TRY(hipSetDevice(c->device));
    if (n <= 0 || !poses3 || !counts || !h_out) return fail(PPGPU_EINVAL, "heuristic_host: bad arguments");
    int stride = 1;
    size_t total = 0;
    for (int i = 0; i < n; i++) {
        if (counts[i] < 0 || counts[i] > PP_WAVE) return fail(PPGPU_ECAPACITY, "heuristic_host: between 0 and 64 ribbons per pose");
        if (counts[i] > stride) stride = counts[i];
        total += (size_t)counts[i];
    }
    if (total > 0 && !ribbons) return fail(PPGPU_EINVAL, "heuristic_host: ribbons are null");
    // records as the cover sweep would leave them: end pose, g = 0, ribbon count; child ribbons at `stride` per record
    std::vector<ppgpu_edge_result> rec((size_t)n);
    std::vector<double> child((size_t)n * stride * 4, 0.0);
    size_t off = 0;
    for (int i = 0; i < n; i++) {
        std::memset(&rec[i], 0, sizeof(ppgpu_edge_result));
        rec[i].info = (uint32_t)counts[i] << 8;
        rec[i].end_x = poses3[3 * i]; rec[i].end_y = poses3[3 * i + 1]; rec[i].end_heading = poses3[3 * i + 2];
        if (counts[i] > 0) std::memcpy(child.data() + (size_t)i * stride * 4, ribbons + off * 4, (size_t)counts[i] * 4 * sizeof(double));
        off += (size_t)counts[i];
    }
    if ((rc = c->tmp_results.reserve((size_t)n, false, c->stream)) || (rc = c->tmp_child.reserve(child.size(), false, c->stream)) ||
        (rc = c->need_big.reserve(32, false, c->stream)) || (rc = c->work.reserve(PP_WORK_WORDS, false, c->stream)))
        return rc;
    HIP_TRY(hipMemsetAsync(c->work.p, 0, (size_t)PP_WORK_WORDS * sizeof(unsigned long long), c->stream));   // queue heads (no solve kernel runs here)
    HIP_TRY(hipMemcpyAsync(c->tmp_results.p, rec.data(), rec.size() * sizeof(ppgpu_edge_result), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(c->tmp_child.p, child.data(), child.size() * sizeof(double), hipMemcpyHostToDevice, c->stream));
    const unsigned one = 1u;       // let the 12-ribbon pass look at every record
    HIP_TRY(hipMemcpyAsync(c->need_big.p, &one, sizeof(unsigned), hipMemcpyHostToDevice, c->stream));
    PPParams p;
    fill_params(c, p);
    p.edges = nullptr; p.wedges = nullptr; p.n_edges = n; p.total_edges = n; p.e_base = 0; p.ws_base = 0;
    p.out = c->tmp_results.p; p.child = c->tmp_child.p; p.stride = stride; p.need_big = c->need_big.p; p.work = c->work.p;
    const dim3 grid((unsigned)((n + PP_H_WPB - 1) / PP_H_WPB)), block(PP_H_WPB * 64);   // n is small: never more than fits
    if (p.heuristic == PPGPU_H_TSP_DUBINS_ALL || p.heuristic == PPGPU_H_TSP_DUBINS_K) hipLaunchKernelGGL(pp_k_heuristic_dubins, grid, block, 0, c->stream, p);
    else hipLaunchKernelGGL(pp_k_heuristic, grid, block, 0, c->stream, p);
    if (p.heuristic == PPGPU_H_TSP_POINT_K)
        hipLaunchKernelGGL(pp_k_heuristic_big, dim3((unsigned)(n < PP_BIG_GRID ? n : PP_BIG_GRID)), dim3(PP_BIG_WPB * 64), 0, c->stream, p);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(rec.data(), c->tmp_results.p, rec.size() * sizeof(ppgpu_edge_result), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    for (int i = 0; i < n; i++) {
        h_out[i] = rec[i].h;
        if (h_flags) h_flags[i] = rec[i].flags;
    }
    return PPGPU_OK;
}

// ------------------------------------------------------------------------------ expand
int64_t ppgpu_expand_capacity(int32_t nv, int32_t k) {
    if (nv <= 0) return 0;
    if (k < 0) k = 0;
    return (int64_t)nv * (4 + 4 * (int64_t)k);
}

// grow-only pinned staging: one H2D in, one D2H out per call instead of a dozen pageable copies
static int stage_reserve(void** p, size_t* cap, size_t n) {
    if (n <= *cap) return PPGPU_OK;
    GrowthTimer growth;
    size_t ncap = *cap ? *cap : 4096;
    while (ncap < n) ncap *= 2;
    if (*p) HIP_TRY(hipHostFree(*p));
    *p = nullptr; *cap = 0;
    HIP_TRY(hipHostMalloc(p, ncap, hipHostMallocDefault));
    *cap = ncap;
    return PPGPU_OK;
}

int ppgpu_expand_host(ppgpu_ctx* c, int32_t nv, const ppgpu_vertex* hv, int32_t n_ribbons, const double* hr, const double* h_nearest,
                      int32_t k, int64_t* n_edges, uint64_t* h_edges, ppgpu_edge_result* h_results, double* h_child, int32_t stride) {
    int rc = require_cfg(c);
    if (rc) return rc;
    HIP_TRY(hipSetDevice(c->device));
    if (nv <= 0 || !hv || n_ribbons < 0 || (n_ribbons > 0 && !hr) || k < 0 || !n_edges || !h_edges || !h_results)
        return fail(PPGPU_EINVAL, "expand_host: bad arguments");
    if (h_child && stride <= 0) return fail(PPGPU_EINVAL, "expand_host: ribbon_stride must be positive");
    if (nv > 65535) return fail(PPGPU_ECAPACITY, "expand_host: at most 65535 vertices per call");
    int maxr = 0;
    for (int i = 0; i < nv; i++) {
        if (hv[i].ribbon_count < 0 || hv[i].ribbon_offset < 0 || hv[i].ribbon_offset + hv[i].ribbon_count > n_ribbons)
            return fail(PPGPU_EINVAL, "expand_host: ribbon range outside the pool");
        if (hv[i].ribbon_count > PP_WAVE) return fail(PPGPU_ECAPACITY, "expand_host: more than 64 ribbons on one vertex");
        if (hv[i].time < c->cfg.start_state_time) return fail(PPGPU_EINVAL, "expand_host: vertex time before start_state_time");
        if (hv[i].ribbon_count > maxr) maxr = hv[i].ribbon_count;
    }
    const long long ns = c->n_samples;
    const bool select = ns > 0 && k > 0;
    const int E = 4 + 4 * k;
    const long long cap = (long long)nv * E;
    hipStream_t st = c->stream;
    // ---- stage in: vertices | ribbons | extra x | extra y | extra heading | has_extra
    const size_t o_v = 0, o_r = o_v + (size_t)nv * sizeof(ppgpu_vertex), o_x = o_r + (size_t)n_ribbons * 4 * sizeof(double),
                 o_y = o_x + (size_t)nv * sizeof(double), o_h = o_y + (size_t)nv * sizeof(double), o_f = o_h + (size_t)nv * sizeof(double),
                 in_bytes = o_f + (size_t)nv;
    if ((rc = stage_reserve(&c->stage_in, &c->stage_in_cap, in_bytes))) return rc;
    char* sin = (char*)c->stage_in;
    std::memcpy(sin + o_v, hv, (size_t)nv * sizeof(ppgpu_vertex));
    if (n_ribbons > 0) std::memcpy(sin + o_r, hr, (size_t)n_ribbons * 4 * sizeof(double));
    for (int i = 0; i < nv; i++) {
        const bool has = h_nearest && !std::isnan(h_nearest[3 * i]);
        ((double*)(sin + o_x))[i] = has ? h_nearest[3 * i] : 0.0;
        ((double*)(sin + o_y))[i] = has ? h_nearest[3 * i + 1] : 0.0;
        ((double*)(sin + o_h))[i] = has ? h_nearest[3 * i + 2] : 0.0;
        ((unsigned char*)(sin + o_f))[i] = has ? 1 : 0;
    }
    const size_t need = (size_t)(ns + nv);
    if ((rc = c->verts.reserve((size_t)nv, false, st)) || (rc = c->ribbons.reserve((size_t)(n_ribbons > 0 ? n_ribbons : 1) * 4, false, st)) ||
        (rc = c->tgrid.reserve((size_t)nv * c->ng, false, st)) || (rc = c->sx.reserve(need, true, st)) || (rc = c->sy.reserve(need, true, st)) ||
        (rc = c->sh.reserve(need, true, st)) || (rc = c->s_bytes.reserve((size_t)nv, false, st)))
        return rc;
    // ---- stage out (device end): header {push-order fallbacks} | descriptors | records | child ribbons, one block, one download
    const size_t q_e = 128, q_r = q_e + (size_t)cap * sizeof(uint64_t), q_c = q_r + (size_t)cap * sizeof(ppgpu_edge_result),
                 out_bytes = q_c + (h_child ? (size_t)cap * stride * 4 * sizeof(double) : 0);
    if ((rc = c->dstage_out.reserve(out_bytes, false, st))) return rc;
    unsigned long long* d_header = (unsigned long long*)c->dstage_out.p;
    unsigned long long* d_edges = (unsigned long long*)(c->dstage_out.p + q_e);
    ppgpu_edge_result* d_results = (ppgpu_edge_result*)(c->dstage_out.p + q_r);
    double* d_child = h_child ? (double*)(c->dstage_out.p + q_c) : nullptr;
    if ((rc = c->dstage_in.reserve(in_bytes, false, st)) || (rc = c->ord_fallbacks.reserve(1, false, st))) return rc;
    HIP_TRY(hipMemcpyAsync(c->dstage_in.p, sin, in_bytes, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(pp_k_expand_unpack, dim3((unsigned)((in_bytes / 8 + nv + 255) / 256)), dim3(256), 0, st, c->dstage_in.p, nv, n_ribbons,
                       c->verts.p, c->ribbons.p, c->sx.p + ns, c->sy.p + ns, c->sh.p + ns, c->s_bytes.p, c->ord_fallbacks.p);
    c->nverts = nv; c->nribbons = n_ribbons; c->max_vertex_ribbons = maxr; c->n_extra = nv;
    hipLaunchKernelGGL(pp_k_time_grid, dim3((unsigned)nv), dim3(64), 0, st, c->verts.p, nv, c->cfg.start_state_time,
                       c->cfg.collision_checking_increment, c->cfg.max_speed, c->ng, c->tgrid.p);
    // ---- k nearest per (vertex, radius), on the device
    if (select && (rc = launch_expand_order(c, nv, k, false))) return rc;   // the k winners per (vertex, radius) in the order expand() pushes them
    const double slow = c->cfg.slow_speed <= 0 ? c->cfg.max_speed : c->cfg.slow_speed;     // PlannerConfig::slowSpeed()
    const int two_speeds = (slow != c->cfg.max_speed) ? 1 : 0;                              // SamplingBasedPlanner.cpp:57-59
    const int two_radii = (c->cfg.coverage_turning_radius != c->cfg.turning_radius) ? 1 : 0;  // :60-63
    hipLaunchKernelGGL(pp_k_build_expand_edges, dim3((unsigned)((nv + 63) / 64)), dim3(64), 0, st, nv, k, select ? c->ord_idx.p : nullptr,
                       c->s_bytes.p, ns, two_speeds, two_radii, E, d_edges, select ? c->ord_fallbacks.p : nullptr, d_header);
    HIP_TRY(hipGetLastError());
    // ---- cost the whole list.  (The child block is NOT cleared on the device: an edge's slots beyond its own ribbon count are
    // written by nobody and read by nobody there; the copy-out below hands the caller zeros for them.)
    PPParams p;
    fill_params(c, p);
    p.edges = d_edges; p.wedges = nullptr;
    p.v0 = 0; p.nv = 0; p.s0 = 0; p.ns = 1; p.cfg_mask = 0; p.per = 1;
    p.n_edges = cap;
    p.out = d_results; p.child = d_child; p.stride = stride;
    if ((rc = launch_cost(c, p))) return rc;
    if ((rc = stage_reserve(&c->stage_out, &c->stage_out_cap, out_bytes))) return rc;
    char* sout = (char*)c->stage_out;
    HIP_TRY(hipMemcpyAsync(sout, c->dstage_out.p, out_bytes, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    c->order_fallbacks += *(const unsigned long long*)sout;
    const uint64_t* se = (const uint64_t*)(sout + q_e);
    const ppgpu_edge_result* sr = (const ppgpu_edge_result*)(sout + q_r);
    const double* sc = (const double*)(sout + q_c);
    int64_t n = 0;
    for (long long i = 0; i < cap; i++) {
        if (se[i] == ~0ull) continue;
        h_edges[n] = se[i];
        h_results[n] = sr[i];
        if (h_child) {
            // the edge's own ribbons, zeros behind them (a record that reports more ribbons than the stride holds: the stride's worth)
            int have = (int)((sr[i].info >> 8) & 0xffu);
            if (have > stride || (sr[i].flags & PPGPU_F_THROWS)) have = (sr[i].flags & PPGPU_F_THROWS) ? 0 : stride;
            double* dst = h_child + (size_t)n * stride * 4;
            std::memcpy(dst, sc + (size_t)i * stride * 4, (size_t)have * 4 * sizeof(double));
            std::memset(dst + (size_t)have * 4, 0, (size_t)(stride - have) * 4 * sizeof(double));
        }
        n++;
    }
    *n_edges = n;
    return PPGPU_OK;
}

// ------------------------------------------------------------------------------ incumbent
int ppgpu_best_edge(ppgpu_ctx* c, int64_t n, const ppgpu_edge_result* d_results, int32_t goal_only, uint64_t base,
                    uint64_t* d_key2) {
    if (!c) return fail(PPGPU_EINVAL, "null context");
    HIP_TRY(hipSetDevice(c->device));
    if (n < 0 || !d_key2 || (n > 0 && !d_results)) return fail(PPGPU_EINVAL, "best_edge: bad arguments");
    int nparts = (int)((n + 255) / 256);
    if (nparts > 1024) nparts = 1024;
    if (nparts < 1) nparts = 1;
    int rc = c->partial.reserve((size_t)nparts * 2, false, c->stream);
    if (rc) return rc;
    hipLaunchKernelGGL(pp_k_best_stage1, dim3(nparts), dim3(256), 0, c->stream, d_results, (long long)n, goal_only,
                       (unsigned long long)base, c->partial.p);
    hipLaunchKernelGGL(pp_k_best_stage2, dim3(1), dim3(256), 0, c->stream, c->partial.p, nparts, (unsigned long long*)d_key2);
    HIP_TRY(hipGetLastError());
    return PPGPU_OK;
}

// Lexicographic min of n (f bits, index) pairs already resident on the device (e.g. the output of an
// all-gather issued by the caller's own communicator).
int ppgpu_key_min(ppgpu_ctx* c, int32_t n, const uint64_t* d_keys, uint64_t* d_key2) {
    if (!c || !d_keys || !d_key2 || n <= 0) return fail(PPGPU_EINVAL, "key_min: bad arguments");
    HIP_TRY(hipSetDevice(c->device));
    hipLaunchKernelGGL(pp_k_key_min_n, dim3(1), dim3(64), 0, c->stream, (const unsigned long long*)d_keys, n, (unsigned long long*)d_key2);
    HIP_TRY(hipGetLastError());
    return PPGPU_OK;
}

}  // extern "C"

// ------------------------------------------------------------------------------ RCCL (incumbent exchange between ranks)
// RCCL is loaded lazily so that single-GPU users (and the CPU-only build check) do not need it.  The NCCL ABI is used through
// dlsym: ncclUniqueId = 128 opaque bytes passed by value, ncclUint64 == 5, ncclSuccess == 0.
#include <dlfcn.h>
namespace {
struct PPNcclId { char internal[PPGPU_COMM_ID_BYTES]; };
struct Rccl {
    int (*get_unique_id)(PPNcclId*) = nullptr;
    int (*comm_init_rank)(void**, int, PPNcclId, int) = nullptr;
    int (*comm_init_all)(void**, int, const int*) = nullptr;
    int (*comm_count)(void*, int*) = nullptr;
    int (*comm_user_rank)(void*, int*) = nullptr;
    int (*comm_destroy)(void*) = nullptr;
    int (*comm_abort)(void*) = nullptr;
    int (*all_gather)(const void*, void*, size_t, int, void*, hipStream_t) = nullptr;
    const char* (*error_string)(int) = nullptr;
    std::string error;
    bool ok = false;
};
// resolved once, published only when complete (contexts may live on several threads)
const Rccl& rccl_table() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        void* lib = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
        if (!lib) lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!lib) lib = dlopen("/opt/rocm/lib/librccl.so", RTLD_NOW | RTLD_GLOBAL);
        if (!lib) { r.error = std::string("cannot load librccl.so: ") + dlerror(); return; }
        r.get_unique_id = (decltype(r.get_unique_id))dlsym(lib, "ncclGetUniqueId");
        r.comm_init_rank = (decltype(r.comm_init_rank))dlsym(lib, "ncclCommInitRank");
        r.comm_init_all = (decltype(r.comm_init_all))dlsym(lib, "ncclCommInitAll");
        r.comm_count = (decltype(r.comm_count))dlsym(lib, "ncclCommCount");
        r.comm_user_rank = (decltype(r.comm_user_rank))dlsym(lib, "ncclCommUserRank");
        r.comm_destroy = (decltype(r.comm_destroy))dlsym(lib, "ncclCommDestroy");
        r.comm_abort = (decltype(r.comm_abort))dlsym(lib, "ncclCommAbort");
        r.all_gather = (decltype(r.all_gather))dlsym(lib, "ncclAllGather");
        r.error_string = (decltype(r.error_string))dlsym(lib, "ncclGetErrorString");
        if (!r.get_unique_id || !r.comm_init_rank || !r.comm_init_all || !r.comm_count || !r.comm_user_rank || !r.comm_destroy || !r.all_gather) {
            r.error = "librccl.so lacks one of ncclGetUniqueId/CommInitRank/CommInitAll/CommCount/CommUserRank/CommDestroy/AllGather";
            return;
        }
        r.ok = true;
    });
    return r;
}
int rccl_fail(const Rccl& r, const char* what, int code) {
    return fail(PPGPU_ERCCL, std::string(what) + " failed: " + (r.error_string ? r.error_string(code) : "") + " (ncclResult " + std::to_string(code) + ")");
}
}  // namespace

extern "C" int ppgpu_comm_unique_id(uint8_t* id128) {
    if (!id128) return fail(PPGPU_EINVAL, "comm_unique_id: null argument");
    const Rccl& r = rccl_table();
    if (!r.ok) return fail(PPGPU_ERCCL, r.error);
    PPNcclId id;
    std::memset(&id, 0, sizeof(id));
    const int rc = r.get_unique_id(&id);
    if (rc != 0) return rccl_fail(r, "ncclGetUniqueId", rc);
    std::memcpy(id128, id.internal, PPGPU_COMM_ID_BYTES);
    return PPGPU_OK;
}

extern "C" int ppgpu_comm_init_rank(ppgpu_ctx* c, int32_t world, int32_t rank, const uint8_t* id128) {
    if (!c || !id128) return fail(PPGPU_EINVAL, "comm_init_rank: null argument");
    if (world <= 0 || rank < 0 || rank >= world) return fail(PPGPU_EINVAL, "comm_init_rank: rank must lie in [0, world)");
    if (c->comm) return fail(PPGPU_ESTATE, "comm_init_rank: the handle already has a communicator (ppgpu_comm_destroy first)");
    const Rccl& r = rccl_table();
    if (!r.ok) return fail(PPGPU_ERCCL, r.error);
    HIP_TRY(hipSetDevice(c->device));
    PPNcclId id;
    std::memcpy(id.internal, id128, PPGPU_COMM_ID_BYTES);
    // the gather buffer first: nothing can fail once the communicator exists
    int rc2 = c->gather.reserve((size_t)world * 2, false, c->stream);
    if (rc2) return rc2;
    void* comm = nullptr;
    const int rc = r.comm_init_rank(&comm, world, id, rank);
    if (rc != 0 || !comm) return rccl_fail(r, "ncclCommInitRank", rc);
    c->comm = comm;
    return PPGPU_OK;
}

extern "C" int ppgpu_comm_init_all(ppgpu_ctx** ctxs, int32_t n) {
    if (!ctxs || n <= 0) return fail(PPGPU_EINVAL, "comm_init_all: bad arguments");
    std::vector<int> devs((size_t)n);
    for (int i = 0; i < n; i++) {
        if (!ctxs[i]) return fail(PPGPU_EINVAL, "comm_init_all: null context");
        if (ctxs[i]->comm) return fail(PPGPU_ESTATE, "comm_init_all: a handle already has a communicator");
        devs[(size_t)i] = ctxs[i]->device;
        for (int j = 0; j < i; j++)
            if (devs[(size_t)j] == devs[(size_t)i]) return fail(PPGPU_EINVAL, "comm_init_all: RCCL takes one rank per device; two handles share device " + std::to_string(devs[(size_t)i]));
    }
    const Rccl& r = rccl_table();
    if (!r.ok) return fail(PPGPU_ERCCL, r.error);
    // every handle's gather buffer first, so that the handles end up with a communicator each or none at all
    for (int i = 0; i < n; i++) {
        HIP_TRY(hipSetDevice(ctxs[i]->device));
        int rc2 = ctxs[i]->gather.reserve((size_t)n * 2, false, ctxs[i]->stream);
        if (rc2) return rc2;
    }
    std::vector<void*> comms((size_t)n, nullptr);
    const int rc = r.comm_init_all(comms.data(), n, devs.data());
    if (rc != 0) {
        for (void* cm : comms) if (cm) (void)r.comm_destroy(cm);
        return rccl_fail(r, "ncclCommInitAll", rc);
    }
    for (int i = 0; i < n; i++) ctxs[i]->comm = comms[(size_t)i];
    return PPGPU_OK;
}

extern "C" int ppgpu_comm_info(ppgpu_ctx* c, int32_t* world, int32_t* rank) {
    if (!c) return fail(PPGPU_EINVAL, "comm_info: null context");
    if (!c->comm) return fail(PPGPU_ESTATE, "comm_info: the handle has no communicator");
    const Rccl& r = rccl_table();
    if (!r.ok) return fail(PPGPU_ERCCL, r.error);
    int w = 0, k = 0, rc;
    if ((rc = r.comm_count(c->comm, &w)) != 0) return rccl_fail(r, "ncclCommCount", rc);
    if ((rc = r.comm_user_rank(c->comm, &k)) != 0) return rccl_fail(r, "ncclCommUserRank", rc);
    if (world) *world = w;
    if (rank) *rank = k;
    return PPGPU_OK;
}

extern "C" int ppgpu_comm_destroy(ppgpu_ctx* c) {
    if (!c) return fail(PPGPU_EINVAL, "comm_destroy: null context");
    if (!c->comm) return PPGPU_OK;
    const Rccl& r = rccl_table();
    if (!r.ok) return fail(PPGPU_ERCCL, r.error);
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    const int rc = r.comm_destroy(c->comm);
    c->comm = nullptr;
    if (rc != 0) return rccl_fail(r, "ncclCommDestroy", rc);
    return PPGPU_OK;
}

// ncclCommAbort: frees the communicator WITHOUT waiting for its outstanding collectives — the way out for the ranks that entered a
// collective a failed rank never joined.  May be called from another host thread than the one stuck in the handle's stream.
extern "C" int ppgpu_comm_abort(ppgpu_ctx* c) {
    if (!c) return fail(PPGPU_EINVAL, "comm_abort: null context");
    void* comm = c->comm;
    if (!comm) return PPGPU_OK;
    const Rccl& r = rccl_table();
    if (!r.ok) return fail(PPGPU_ERCCL, r.error);
    c->comm = nullptr;
    const int rc = r.comm_abort ? r.comm_abort(comm) : r.comm_destroy(comm);
    if (rc != 0) return rccl_fail(r, "ncclCommAbort", rc);
    return PPGPU_OK;
}

// ------------------------------------------------------------------------------ records home on a copy engine
// hipMemcpyAsync from device memory into pinned host memory runs as a blit KERNEL on this stack (__amd_rocclr_copyBuffer in every
// rocprofv3 kernel trace of bench.py's end-to-end loops, with HSA_ENABLE_SDMA=1 and with GPU_BLIT_ENGINE_TYPE=2 alike): 30 MB of
// records take 0.62 ms of wave slots beside fp64-bound kernels.  hsa_amd_memory_async_copy puts the same copy on an SDMA engine.  The
// HSA runtime is the one HIP itself runs on (same soname: the loader hands back the instance already in the process).
#include <hsa/hsa.h>
#include <hsa/hsa_ext_amd.h>
namespace {
struct HsaApi {
    decltype(&hsa_iterate_agents) iterate_agents = nullptr;
    decltype(&hsa_agent_get_info) agent_get_info = nullptr;
    decltype(&hsa_signal_create) signal_create = nullptr;
    decltype(&hsa_signal_destroy) signal_destroy = nullptr;
    decltype(&hsa_signal_store_relaxed) signal_store_relaxed = nullptr;
    decltype(&hsa_signal_wait_scacquire) signal_wait_scacquire = nullptr;
    decltype(&hsa_amd_memory_async_copy) memory_async_copy = nullptr;
    decltype(&hsa_amd_pointer_info) pointer_info = nullptr;
    std::string error;
    bool ok = false;
};
const HsaApi& hsa_api() {
    static HsaApi a;
    static std::once_flag once;
    std::call_once(once, [] {
        void* lib = dlopen("libhsa-runtime64.so.1", RTLD_NOW);
        if (!lib) lib = dlopen("libhsa-runtime64.so", RTLD_NOW);
        if (!lib) { a.error = std::string("cannot load libhsa-runtime64.so.1: ") + dlerror(); return; }
        a.iterate_agents = (decltype(a.iterate_agents))dlsym(lib, "hsa_iterate_agents");
        a.agent_get_info = (decltype(a.agent_get_info))dlsym(lib, "hsa_agent_get_info");
        a.signal_create = (decltype(a.signal_create))dlsym(lib, "hsa_signal_create");
        a.signal_destroy = (decltype(a.signal_destroy))dlsym(lib, "hsa_signal_destroy");
        a.signal_store_relaxed = (decltype(a.signal_store_relaxed))dlsym(lib, "hsa_signal_store_relaxed");
        a.signal_wait_scacquire = (decltype(a.signal_wait_scacquire))dlsym(lib, "hsa_signal_wait_scacquire");
        a.memory_async_copy = (decltype(a.memory_async_copy))dlsym(lib, "hsa_amd_memory_async_copy");
        a.pointer_info = (decltype(a.pointer_info))dlsym(lib, "hsa_amd_pointer_info");
        if (!a.iterate_agents || !a.agent_get_info || !a.signal_create || !a.signal_destroy || !a.signal_store_relaxed || !a.signal_wait_scacquire || !a.memory_async_copy || !a.pointer_info) {
            a.error = "libhsa-runtime64 lacks one of the entry points ppgpu_copy_engine_read needs";
            return;
        }
        a.ok = true;
    });
    return a;
}
struct AgentScan { const HsaApi* api; int want_gpu; int seen_gpu; hsa_agent_t gpu, cpu; bool have_gpu, have_cpu; };
hsa_status_t scan_agent(hsa_agent_t agent, void* data) {
    AgentScan* s = (AgentScan*)data;
    hsa_device_type_t type;
    if (s->api->agent_get_info(agent, HSA_AGENT_INFO_DEVICE, &type) != HSA_STATUS_SUCCESS) return HSA_STATUS_SUCCESS;
    if (type == HSA_DEVICE_TYPE_CPU && !s->have_cpu) { s->cpu = agent; s->have_cpu = true; }
    if (type == HSA_DEVICE_TYPE_GPU) {
        if (s->seen_gpu == s->want_gpu) { s->gpu = agent; s->have_gpu = true; }
        s->seen_gpu++;
    }
    return HSA_STATUS_SUCCESS;
}
}  // namespace

static void ppgpu_copy_engine_release(ppgpu_ctx* c) {
    if (!c->hsa_signal) return;
    hsa_signal_t sig; sig.handle = c->hsa_signal;
    (void)hsa_api().signal_destroy(sig);
    c->hsa_signal = 0;
}

extern "C" int ppgpu_copy_engine_wait(ppgpu_ctx* c) {
    if (!c) return fail(PPGPU_EINVAL, "copy_engine_wait: null context");
    if (!c->copy_pending) return PPGPU_OK;
    const HsaApi& a = hsa_api();
    hsa_signal_t sig; sig.handle = c->hsa_signal;
    // (the signal starts at 1; the copy engine takes it to 0, or below on an error)
    const hsa_signal_value_t v = a.signal_wait_scacquire(sig, HSA_SIGNAL_CONDITION_LT, 1, UINT64_MAX, HSA_WAIT_STATE_BLOCKED);
    c->copy_pending = false;
    if (v < 0) return fail(PPGPU_EHIP, "copy_engine_wait: the copy engine reported an error");
    return PPGPU_OK;
}

extern "C" int ppgpu_copy_engine_read(ppgpu_ctx* c, void* h_pinned_dst, const void* d_src, uint64_t bytes) {
    if (!c || !h_pinned_dst || !d_src || bytes == 0) return fail(PPGPU_EINVAL, "copy_engine_read: bad arguments");
    const HsaApi& a = hsa_api();
    if (!a.ok) return fail(PPGPU_EHIP, a.error);
    int rc;
    if (c->copy_pending && (rc = ppgpu_copy_engine_wait(c))) return rc;          // one copy in flight per handle
    if (!c->hsa_signal) {
        AgentScan s{&a, c->device, 0, {}, {}, false, false};
        if (a.iterate_agents(scan_agent, &s) != HSA_STATUS_SUCCESS || !s.have_cpu)
            return fail(PPGPU_EHIP, "copy_engine_read: cannot find the HSA agent of the host");
        // the GPU agent is the one that OWNS the source allocation (a HIP device ordinal is not an index into HSA's agent list once
        // HIP_VISIBLE_DEVICES reorders or masks devices)
        hsa_amd_pointer_info_t info;
        std::memset(&info, 0, sizeof(info));
        info.size = sizeof(info);
        if (a.pointer_info(d_src, &info, nullptr, nullptr, nullptr) != HSA_STATUS_SUCCESS || info.type == HSA_EXT_POINTER_TYPE_UNKNOWN)
            return fail(PPGPU_EHIP, "copy_engine_read: the source is not a device allocation HSA knows");
        hsa_device_type_t owner_type;
        if (a.agent_get_info(info.agentOwner, HSA_AGENT_INFO_DEVICE, &owner_type) != HSA_STATUS_SUCCESS || owner_type != HSA_DEVICE_TYPE_GPU)
            return fail(PPGPU_EHIP, "copy_engine_read: the source allocation is not owned by a GPU agent");
        s.gpu = info.agentOwner; s.have_gpu = true;
        hsa_signal_t sig;
        if (a.signal_create(1, 0, nullptr, &sig) != HSA_STATUS_SUCCESS) return fail(PPGPU_EHIP, "copy_engine_read: hsa_signal_create failed");
        c->hsa_gpu = s.gpu.handle; c->hsa_cpu = s.cpu.handle; c->hsa_signal = sig.handle;
    }
    hsa_signal_t sig; sig.handle = c->hsa_signal;
    hsa_agent_t gpu, cpu; gpu.handle = c->hsa_gpu; cpu.handle = c->hsa_cpu;
    a.signal_store_relaxed(sig, 1);
    const hsa_status_t st = a.memory_async_copy(h_pinned_dst, cpu, d_src, gpu, (size_t)bytes, 0, nullptr, sig);
    if (st != HSA_STATUS_SUCCESS) return fail(PPGPU_EHIP, "copy_engine_read: hsa_amd_memory_async_copy failed (status " + std::to_string((int)st) + "): is the destination pinned host memory?");
    c->copy_pending = true;
    return PPGPU_OK;
}

extern "C" int ppgpu_allreduce_best(ppgpu_ctx* c, void* comm, uint64_t* d_key2) {
    if (!c || !d_key2) return fail(PPGPU_EINVAL, "allreduce_best: null argument");
    if (!comm) comm = c->comm;
    if (!comm) return fail(PPGPU_ESTATE, "allreduce_best: no communicator (pass one, or ppgpu_comm_init_rank / ppgpu_comm_init_all first)");
    HIP_TRY(hipSetDevice(c->device));
    const Rccl& r = rccl_table();
    if (!r.ok) return fail(PPGPU_ERCCL, r.error);
    int world = 0, rc;
    if ((rc = r.comm_count(comm, &world)) != 0 || world <= 0) return rccl_fail(r, "ncclCommCount", rc);
    int rc2 = c->gather.reserve((size_t)world * 2, false, c->stream);
    if (rc2) return rc2;
    // one collective: 16 bytes per rank over xGMI; ncclUint64 == 5 in the NCCL ABI
    if ((rc = r.all_gather(d_key2, c->gather.p, 2, 5, comm, c->stream)) != 0) return rccl_fail(r, "ncclAllGather", rc);
    hipLaunchKernelGGL(pp_k_key_min_n, dim3(1), dim3(64), 0, c->stream, c->gather.p, world, (unsigned long long*)d_key2);
    HIP_TRY(hipGetLastError());
    return PPGPU_OK;
}
