// planner.cpp — Planner / GpuContext / GpuAStarPlanner (see include/path_planner_amd/Planner.h).
//
// Control flow follows /root/reference/path_planner/src/planner/AStarPlanner.cpp:12-148 and
// SamplingBasedPlanner.cpp:7-27,42-151 line by line, including where the clock is polled; the data-parallel work
// (sampling, Dubins lengths to every sample, k-nearest selection, edge costing with coverage/heuristic) is done by the
// device library behind include/ppgpu.h.  Any device error becomes std::runtime_error, which Executive::planLoop
// catches like any other planner exception (executive.cpp:191-195).
#include "path_planner_amd/Planner.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <map>
#include <mutex>
#include <stdexcept>
#include <string>
#include <utility>

#include "../../../include/ppgpu.h"

namespace ppamd {

// PPAMD_PROFILE=1: where a plan() call's wall time goes on the host side, to stderr (developer aid)
namespace {
struct HostProfile {
    bool on = std::getenv("PPAMD_PROFILE") != nullptr;
    double t[6] = {0, 0, 0, 0, 0, 0};
    unsigned long trips = 0;
    static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
    void report(const char* what) {
        if (!on) return;
        std::fprintf(stderr, "[profile] %s: round trips %lu | pick %.1f ms | pack %.1f ms | device %.1f ms | children %.1f ms | push %.1f ms | samples %.1f ms\n",
                     what, trips, t[0] * 1e3, t[1] * 1e3, t[2] * 1e3, t[3] * 1e3, t[4] * 1e3, t[5] * 1e3);
        for (double& x : t) x = 0;
        trips = 0;
    }
};
HostProfile g_prof;
// PPAMD_DUMP_EDGES=<file>: every costed edge the search consumes, in consumption order, 16 doubles per line in the layout of the
// oracle's edge dump (source state, Dubins parameters and type, radius, coverage flag, infeasible, true cost, g, h, end time)
struct EdgeDump {
    FILE* f = nullptr;
    EdgeDump() { if (const char* p = std::getenv("PPAMD_DUMP_EDGES")) f = std::fopen(p, "w"); }
    ~EdgeDump() { if (f) std::fclose(f); }
    void write(const State& src, const ppgpu_edge_result& r, double rho, bool cov) {
        if (!f) return;
        std::fprintf(f, "%.17g %.17g %.17g %.17g %.17g %.17g %.17g %.17g %d %.17g %d %d %.17g %.17g %.17g %.17g\n", src.x(), src.y(), src.heading(),
                     src.speed(), src.time(), r.param[0], r.param[1], r.param[2], (int)(r.info & 0xff), rho, cov ? 1 : 0,
                     (r.flags & PPGPU_F_INFEASIBLE) ? 1 : 0, r.true_cost, r.g, r.h, r.end_time);
    }
};
EdgeDump g_dump;
struct Lap {
    int slot; double t0;
    explicit Lap(int s) : slot(s), t0(g_prof.on ? HostProfile::now() : 0) {}
    ~Lap() { if (g_prof.on) g_prof.t[slot] += HostProfile::now() - t0; }
};
}

static const double kTimePenaltyFactor = 1;       // Edge::timePenaltyFactor() (Edge.h:152)
static const double kCollisionPenaltyFactor = 600;  // Edge::collisionPenaltyFactor() (Edge.h:151)
static const double kHostMargin = 1e-4;           // seconds the guarded search loop stops short of the deadline (tracing the plan, statistics, the return)
static const int kRibbonStride = 64;              // child ribbon capacity per edge = the device's per-vertex limit
// What the deadline guard allows a round trip on top of its prediction when some open vertex of the batch carries a long ribbon list: its
// children may carry one ribbon more, and the K-ribbon heuristic of a 9-12-ribbon list is enumerated by pp_k_heuristic_big — 0.08, 0.4 and
// 2 ms per list of 10, 11 and 12 ribbons (tools/big_heuristic_time.py), in a trip that otherwise takes 0.6 ms.  Such trips were the guard's
// largest under-predictions (round 4: two cycles of a hundred at 101.4 ms late in the mission, when the lists are long).
static double heavyListAllowance(int maxParentRibbons) {
    return maxParentRibbons >= 11 ? 3.0e-3 : maxParentRibbons == 10 ? 1.0e-3 : maxParentRibbons == 9 ? 0.5e-3 : maxParentRibbons == 8 ? 0.25e-3 : 0.0;
}
static const size_t kNodeArenaMin = 1u << 20;     // nodes the search tree has room for before its first node arrives (250 MB of address space, touched as used;
                                                  // a 100 ms cycle of config 5 makes 250 000 - 300 000 of them)

Planner::Planner() : m_Config(PlannerConfig(&std::cerr)) {}

Planner::Stats Planner::plan(const RibbonManager&, const State&, PlannerConfig config, const DubinsPlan&, double) {
    m_Config = std::move(config);
    throw std::runtime_error("Ribbon point-to-point planner is not yet implemented");   // Planner.cpp:41-45
}

// ------------------------------------------------------------------------------------------------ GpuContext
GpuContext::GpuContext(int device) : m_Device(device) {
    if (ppgpu_create(device, &m_Handle) != PPGPU_OK) throw std::runtime_error(std::string("ppgpu_create: ") + ppgpu_last_error());
    // The anytime search doubles its sample set every iteration; a 10 Hz cycle with a 100 ms budget reaches one to four million.
    // Sizing the sample-dependent buffers once (8 M samples, batches of up to 64 vertices: 9 GB of 288) keeps device allocations out
    // of every later cycle's budget.  PPAMD_RESERVE_SAMPLES=0 leaves them to grow on demand.
    long long reserve = 8ll << 20;
    if (const char* e = std::getenv("PPAMD_RESERVE_SAMPLES")) reserve = std::atoll(e);
    if (reserve > 0 && ppgpu_reserve_samples(m_Handle, reserve, 64) != PPGPU_OK) {
        const std::string why = ppgpu_last_error();
        ppgpu_destroy(m_Handle);      // the destructor does not run for a constructor that throws
        m_Handle = nullptr;
        throw std::runtime_error("ppgpu_reserve_samples: " + why);
    }
    m_Thread = std::thread(&GpuContext::serve, this);
}

GpuContext::~GpuContext() {
    {
        std::lock_guard<std::mutex> lock(m_Mutex);
        m_Quit = true;
    }
    m_Wake.notify_all();
    if (m_Thread.joinable()) m_Thread.join();
    ppgpu_destroy(m_Handle);
}

// After a job the thread keeps polling for the next one for a millisecond before it sleeps on the condition variable: inside a
// planning cycle the next round trip comes within that time, and a thread that never slept needs no wake-up (measured in round 4:
// the wake-up is 40 us at the 90th percentile but 2.4 ms once in a few thousand, enough to carry a cycle's last round trip past
// the deadline).  Between cycles it sleeps.
void GpuContext::serve() {
    std::unique_lock<std::mutex> lock(m_Mutex);
    bool justWorked = false;
    for (;;) {
        if (justWorked) {
            lock.unlock();
            const auto t0 = std::chrono::steady_clock::now();
            while (!m_Posted.load(std::memory_order_acquire) &&
                   std::chrono::steady_clock::now() - t0 < std::chrono::microseconds(1000)) {
#if defined(__x86_64__)
                __builtin_ia32_pause();
#endif
            }
            lock.lock();
        }
        m_Wake.wait(lock, [this] { return m_Quit || (m_Busy && m_Job); });
        if (m_Quit) return;
        std::function<void()> job = std::move(m_Job);
        m_Job = nullptr;
        m_Posted.store(false, std::memory_order_release);
        justWorked = true;
        lock.unlock();
        std::exception_ptr err;
        try { job(); } catch (...) { err = std::current_exception(); }
        lock.lock();
        m_Error = err;
        m_Busy = false;
        m_Running.store(false, std::memory_order_release);
        m_Wake.notify_all();
    }
}

void GpuContext::run(std::function<void()> job) {
    std::unique_lock<std::mutex> lock(m_Mutex);
    m_Wake.wait(lock, [this] { return !m_Busy; });
    m_Job = std::move(job);
    m_Busy = true;
    m_Error = nullptr;
    m_Running.store(true, std::memory_order_release);
    m_Posted.store(true, std::memory_order_release);
    m_Wake.notify_all();
}

void GpuContext::wait() {
    {
        // (the planner's thread has nothing else to do while it waits for a round trip: polling for up to 3 ms spares it the wake-up)
        const auto t0 = std::chrono::steady_clock::now();
        while (m_Running.load(std::memory_order_acquire) && std::chrono::steady_clock::now() - t0 < std::chrono::microseconds(3000)) {
#if defined(__x86_64__)
            __builtin_ia32_pause();
#endif
        }
    }
    std::unique_lock<std::mutex> lock(m_Mutex);
    m_Wake.wait(lock, [this] { return !m_Busy; });
    if (m_Error) {
        std::exception_ptr e = m_Error;
        m_Error = nullptr;
        std::rethrow_exception(e);
    }
}

bool GpuContext::idle() {
    std::lock_guard<std::mutex> lock(m_Mutex);
    return !m_Busy;
}

void TripBlock::reserve(size_t nEdges, size_t childDoubles) {
    if (nEdges > edgeCap) {
        edges.reset(new uint64_t[nEdges]);
        records.reset(new unsigned char[nEdges * sizeof(ppgpu_edge_result)]);
        edgeCap = nEdges;
    }
    if (childDoubles > childCap) { child.reset(new double[childDoubles]); childCap = childDoubles; }
}

std::shared_ptr<TripBlock> GpuContext::takeTripBlock(size_t nEdges, size_t childDoubles) {
    for (auto& b : tripPool)
        if (b.use_count() == 1) { b->reserve(nEdges, childDoubles); return b; }       // nobody but the pool holds it
    tripPool.push_back(std::make_shared<TripBlock>());
    tripPool.back()->reserve(nEdges, childDoubles);
    return tripPool.back();
}

namespace {
std::mutex g_ctxMutex;
// deliberately never destroyed: contexts must not outlive the HIP runtime's own static teardown
std::map<std::pair<int, int>, std::shared_ptr<GpuContext>>& contextCache() {
    static auto* cache = new std::map<std::pair<int, int>, std::shared_ptr<GpuContext>>();
    return *cache;
}
}  // namespace

std::shared_ptr<GpuContext> GpuContext::shared(int device, int lane) {
    std::lock_guard<std::mutex> lock(g_ctxMutex);
    std::shared_ptr<GpuContext>& sp = contextCache()[{device, lane}];
    if (!sp) sp = std::make_shared<GpuContext>(device);
    return sp;
}

// a device id that repeats names a further context (its own stream and buffers) on that device: {0, 0} = two streams on device 0
std::vector<std::shared_ptr<GpuContext>> GpuContext::shared(const std::vector<int>& devices) {
    std::vector<std::shared_ptr<GpuContext>> out;
    std::map<int, int> seen;
    for (int d : devices) out.push_back(shared(d, seen[d]++));
    if (out.empty()) out.push_back(shared(0));
    return out;
}

void GpuContext::releaseShared() {
    std::lock_guard<std::mutex> lock(g_ctxMutex);
    contextCache().clear();
}

// ------------------------------------------------------------------------------------------------ helpers
GpuAStarPlanner::~GpuAStarPlanner() {
    try { drainInFlight(); } catch (...) {}     // (after an exception left plan(): nothing of this planner may still run on a context)
    // the tree goes (every node's ribbon list is its own allocation), its array goes back to the context for the next cycle's planner
    m_Nodes.clear();
    if (m_Ctx && m_Nodes.capacity() > m_Ctx->nodeArena.capacity()) m_Nodes.swap(m_Ctx->nodeArena);
}

void GpuAStarPlanner::check(int rc, const char* what) const {
    if (rc != PPGPU_OK) throw std::runtime_error(std::string(what) + ": " + ppgpu_last_error());
}

static void ribbonsToArray(const RibbonManager& rm, std::vector<double>& out) {
    out.assign(rm.rows(), rm.rows() + 4 * (size_t)rm.count());   // the manager keeps its list in the device's row layout
}

static ppgpu_vertex makeVertex(const GpuAStarPlanner::Node& n) {
    ppgpu_vertex v;
    v.x = n.state.x(); v.y = n.state.y(); v.heading = n.state.heading(); v.speed = n.state.speed(); v.time = n.state.time();
    v.g = n.g;
    v.coverage_completed_time = n.ribbons.coverageCompletedTime();
    v.ribbon_offset = 0;
    v.ribbon_count = (int32_t)n.ribbons.count();
    return v;
}

int GpuAStarPlanner::depth(int v) const {
    int d = 0;
    for (int p = m_Nodes[v].parent; p >= 0; p = m_Nodes[p].parent) d++;
    return d;
}

// ------------------------------------------------------------------------------------------------ world upload
// The snapshot every device of a planner works from: configuration, occupancy grid, obstacle table (SURVEY 8 e: replicated on
// every device once per replan).
// -> true when the occupancy grid was rasterised and uploaded (false: every device already held this very map, Map::version)
static bool uploadSnapshot(const std::vector<std::shared_ptr<GpuContext>>& ctxs, const PlannerConfig& config, const RibbonManager& ribbonManager,
                           double startStateTime) {
    ppgpu_config c{};
    c.max_speed = config.maxSpeed();
    c.slow_speed = config.slowSpeed();
    c.turning_radius = config.turningRadius();
    c.coverage_turning_radius = config.coverageTurningRadius();
    c.time_horizon = config.timeHorizon();
    c.time_minimum = config.timeMinimum();
    c.collision_checking_increment = config.collisionCheckingIncrement();
    c.start_state_time = startStateTime;
    c.ribbon_width = Ribbon::RibbonWidth;
    c.collision_penalty_factor = kCollisionPenaltyFactor;
    c.time_penalty_factor = kTimePenaltyFactor;
    c.heuristic_turning_radius = ribbonManager.turningRadius();
    c.heuristic = (int32_t)ribbonManager.heuristic();
    c.tsp_k = ribbonManager.k();
    c.branching_factor = config.branchingFactor();
    // The grid of a 2048 x 2048 map is 4 million cells to rasterise, pack and upload, and a clearance map to rebuild: ~10 ms of a
    // 100 ms cycle, for a map the Executive replaces once in minutes.  A map that vouches for its cells (Map::version != 0) is
    // uploaded once per device and recognised afterwards.
    const Map* map = config.map().get();
    const unsigned long mapVersion = map ? map->version() : 0;
    bool needGrid = false;
    for (const auto& ctx : ctxs) needGrid = needGrid || mapVersion == 0 || ctx->gridOf != (const void*)map || ctx->gridVersion != mapVersion;
    std::vector<uint8_t> cells;
    int rows = 0, cols = 0;
    double res = 0;
    if (map && needGrid) map->rasterize(cells, rows, cols, res);
    std::vector<double> orows;
    const DynamicObstaclesManager& om = config.obstaclesManager();
    om.deviceRows(orows);
    auto check = [](int rc, const char* what) { if (rc != PPGPU_OK) throw std::runtime_error(std::string(what) + ": " + ppgpu_last_error()); };
    for (const auto& ctx : ctxs) {
        ppgpu_ctx* h = ctx->handle();
        check(ppgpu_set_config(h, &c), "ppgpu_set_config");
        if (needGrid) {
            ctx->gridOf = nullptr; ctx->gridVersion = 0;       // until the upload below has succeeded
            check(ppgpu_set_grid(h, rows ? cells.data() : nullptr, rows, cols, res), "ppgpu_set_grid");
            if (mapVersion != 0) { ctx->gridOf = (const void*)map; ctx->gridVersion = mapVersion; }
        }
        if (om.deviceModel() == PPGPU_OBST_GAUSSIAN)
            check(ppgpu_set_gaussian_obstacles(h, (int32_t)(orows.size() / 9), orows.empty() ? nullptr : orows.data(), 1), "ppgpu_set_gaussian_obstacles");
        else
            check(ppgpu_set_obstacles(h, om.deviceModel(), (int32_t)(orows.size() / 7), orows.empty() ? nullptr : orows.data()), "ppgpu_set_obstacles");
    }
    return needGrid;
}

void GpuAStarPlanner::uploadWorld(const State& start) { m_Stats.Budget.GridUploaded = uploadSnapshot(m_Ctxs, m_Config, m_RibbonManager, start.time()); }

// ------------------------------------------------------------------------------------------------ budget bookkeeping
void GpuAStarPlanner::addNode(Node&& n) {
    if (m_Nodes.size() == m_Nodes.capacity()) {        // the vector is about to move every node: counted and timed (Stats::Budget)
        const double t0 = HostProfile::now();
        m_Nodes.push_back(std::move(n));
        m_Stats.Budget.NodeRegrowths++;
        m_Stats.Budget.NodeRegrowthMs += 1e3 * (HostProfile::now() - t0);
        return;
    }
    m_Nodes.push_back(std::move(n));
}

void GpuAStarPlanner::noteOperation(int kind, double startedAt, double predicted, double actual) {
    Stats::BudgetTrace& b = m_Stats.Budget;
    b.LastOpKind = kind;
    b.LastOpStartMs = 1e3 * (startedAt - m_PlanEntry);
    b.LastOpPredictedMs = 1e3 * predicted;
    b.LastOpActualMs = 1e3 * actual;
    if (kind == 1) { b.RoundTrips++; b.MaxTripMs = std::max(b.MaxTripMs, 1e3 * actual); }
    if (predicted > 0) b.WorstUnderPredictionMs = std::max(b.WorstUnderPredictionMs, 1e3 * (actual - predicted));
}

// ------------------------------------------------------------------------------------------------ open list
void GpuAStarPlanner::pushVertexQueue(int vi) {   // SamplingBasedPlanner.cpp:7-19
    Node& v = m_Nodes[vi];
    if (v.parent >= 0 && v.infeasible) return;
    if (v.h == -1) throw std::runtime_error("Fetching unset approx to go (h)");
    if (m_Best >= 0 && m_Nodes[m_Best].f() < v.f()) return;
    if (m_Best >= 0 && m_Nodes[m_Best].f() == v.f() && goalCondition(v)) return;
    m_Queue.push_back(QEntry{v.f(), vi});
    std::push_heap(m_Queue.begin(), m_Queue.end(), [](const QEntry& a, const QEntry& b) { return a.f > b.f; });
    visualizeVertex(vi, "vertex", false);
    m_Stats.Generated++;
}

// ------------------------------------------------------------------------------------------------ search dump
// "Generated|Expanded State: (x y heading speed time), f: F, g: G, h: H tag id id ... " — Vertex::toString (Vertex.cpp:114-119)
// plus the chain of vertex identities from the root (the reference prints object addresses, Vertex.cpp:140-143; here the
// identity is the node's index + 1: visualizer.py only needs a distinct integer per vertex).
void GpuAStarPlanner::visualizeVertex(int vi, const char* tag, bool expanded) {
    if (!m_Config.visualizations()) return;
    const Node& v = m_Nodes[vi];
    std::vector<int> chain;
    for (int cur = vi; cur >= 0; cur = m_Nodes[cur].parent) chain.push_back(cur);
    std::ostream& o = m_Config.visualizationStream();
    o << (expanded ? "Expanded " : "Generated ") << "State: (" << v.state.toStringRad() << "), f: " << v.g + v.h << ", g: " << v.g
      << ", h: " << v.h << " " << tag << " ";
    for (auto it = chain.rbegin(); it != chain.rend(); ++it) o << (*it + 1) << " ";
    o << std::endl;
}

// What Edge::computeTrueCost streams while it sweeps (Edge.cpp:122-143): "Trajectory:" and every int(1/increment)+1-th sampled
// state with the cost accrued so far (start g + time so far + collision penalty of the steps before it) and the START vertex's
// h.  Rebuilt here from the child's curve, the same time grid (:114-120) and the host obstacle manager.
void GpuAStarPlanner::visualizeTrajectory(const Node& child) {
    if (!m_Config.visualizations() || child.parent < 0) return;
    const Node& src = m_Nodes[child.parent];
    std::ostream& o = m_Config.visualizationStream();
    o << "Trajectory:" << std::endl;
    const double timeIncrement = m_Config.collisionCheckingIncrement() / m_Config.maxSpeed();
    State intermediate(src.state);
    intermediate.time() += std::fmod(intermediate.time() - m_StartStateTime, timeIncrement);
    int visCount = int(1.0 / m_Config.collisionCheckingIncrement());
    double collisionPenalty = 0;
    for (int k = 0; k < child.steps; k++) {
        try {
            child.wrapper.sample(intermediate);
        } catch (std::runtime_error&) {
            break;
        }
        if (visCount-- <= 0) {
            visCount = int(1.0 / m_Config.collisionCheckingIncrement());
            const double gSoFar = src.g + (intermediate.time() - src.state.time()) + collisionPenalty;
            o << "State: (" << intermediate.toStringRad() << "), f: " << gSoFar + src.h << ", g: " << gSoFar << ", h: " << src.h
              << " trajectory" << std::endl;
        }
        collisionPenalty += m_Config.obstaclesManager().collisionExists(intermediate, true) * kCollisionPenaltyFactor;
        intermediate.time() += timeIncrement;
    }
}

void GpuAStarPlanner::visualizePlan(const DubinsPlan& plan) {   // SamplingBasedPlanner.cpp:227-238: one state per second
    if (!m_Config.visualizations() || plan.empty()) return;
    State s;
    s.time() = plan.getStartTime();
    while (s.time() < plan.getEndTime()) {
        plan.sample(s);
        m_Config.visualizationStream() << "State: (" << s.toStringRad() << "), f: " << 0 << ", g: " << 0 << ", h: " << 0 << " plan" << std::endl;
        s.time() += 1;
    }
}

void GpuAStarPlanner::visualizeSamples() {   // AStarPlanner.cpp:103-108: every sample, every iteration (read back from the device)
    if (!m_Config.visualizations() || m_NumSamples <= 0) return;
    std::vector<double> s5((size_t)m_NumSamples * 5);
    check(ppgpu_get_samples(m_Ctx->handle(), 0, m_NumSamples, s5.data()), "ppgpu_get_samples");
    for (long i = 0; i < m_NumSamples; i++) {
        State s(s5[5 * i], s5[5 * i + 1], s5[5 * i + 2], s5[5 * i + 3], s5[5 * i + 4]);
        m_Config.visualizationStream() << "State: (" << s.toStringRad() << "), f: " << 0 << ", g: " << 0 << ", h: " << 0 << " sample" << std::endl;
    }
}

int GpuAStarPlanner::popVertexQueue() {   // :21-27
    if (m_Queue.empty()) throw std::out_of_range("Trying to pop an empty vertex queue");
    std::pop_heap(m_Queue.begin(), m_Queue.end(), [](const QEntry& a, const QEntry& b) { return a.f > b.f; });
    int r = m_Queue.back().v;
    m_Queue.pop_back();
    return r;
}

bool GpuAStarPlanner::goalCondition(const Node& v) const {   // :42-50
    double coverageDoneTime = v.ribbons.coverageCompletedTime() + m_Config.timeMinimum();
    if (v.ribbons.coverageCompletedTime() == -1 && v.ribbons.done())
        throw std::runtime_error("Unset coverage completed time but coverage is done");
    double nonCoverageDoneTime = m_StartStateTime + m_Config.timeHorizon();
    return v.state.time() >= nonCoverageDoneTime || (v.ribbons.done() && v.state.time() >= coverageDoneTime);
}

// ------------------------------------------------------------------------------------------------ sampling
void GpuAStarPlanner::addSamples(long n) {   // SamplingBasedPlanner::addSamples (:157-168)
    drainInFlight();                          // round trips still running were chosen among the old samples (and use the contexts)
    Lap lap(5);
    int64_t total = 0;
    // every device draws the same attempts from the same generator state itself (the stream is a pure function of seed and
    // position): nothing is copied between devices
    auto draw = [this, n](GpuContext& ctx, int64_t& kept) {
        long left = n;
        while (left > 0) {   // the device sampler takes at most 524288 attempts per call
            long chunk = std::min<long>(left, 524288);
            check(ppgpu_sampler_add(ctx.handle(), chunk, &kept), "ppgpu_sampler_add");
            left -= chunk;
        }
    };
    const double w0 = HostProfile::now();
    if (m_Ctxs.size() == 1) {
        draw(*m_Ctx, total);
    } else {
        std::vector<int64_t> kept(m_Ctxs.size(), 0);
        for (size_t d = 0; d < m_Ctxs.size(); d++) m_Ctxs[d]->run([&, d] { draw(*m_Ctxs[d], kept[d]); });
        // every job refers to `kept` and `draw`: wait for ALL of them before anything is thrown
        std::exception_ptr first;
        for (auto& ctx : m_Ctxs) {
            try { ctx->wait(); } catch (...) { if (!first) first = std::current_exception(); }
        }
        if (first) std::rethrow_exception(first);
        total = kept[0];
        for (int64_t t : kept) if (t != total) throw std::runtime_error("the devices disagree on the sample set");
    }
    if (n > 0) {
        const double took = HostProfile::now() - w0;
        noteOperation(2, w0, m_Stats.Iterations > 0 ? m_Ctx->predictDoubling((double)n) : 0.0, took);
        m_Ctx->doubling = {(double)n, took};
    }
    if (n > 0) m_NumSamples = (long)total;
    m_Speculated.clear();   // children costed ahead were chosen among the old samples
}

// ------------------------------------------------------------------------------------------------ edges
// The child Node of one costed edge (what Vertex::connect + Edge::computeTrueCost + Vertex::computeApproxToGo leave behind)
GpuAStarPlanner::Node GpuAStarPlanner::makeChild(int source, unsigned cfgBits, const ppgpu_edge_result& r, const double* childRibbons, int stride) {
    if (r.flags & PPGPU_F_THROWS) throw std::runtime_error("Edge cost evaluation failed: invalid time in sample for Dubins path");
    const int nChild = (int)((r.info >> 8) & 0xff);
    // PPGPU_F_RIBBON_OVF on a list that came back whole means only that the device's TSP enumeration stops at 8 (12) ribbons: the
    // reference enumerates any length (RibbonManager.cpp:53-140), so h is computed here, with the same arithmetic as the root's
    // (PPGPU_F_RIBBON_LOST: the sweep itself ran out of its 64 ribbons per vertex and dropped pieces — that record is not the
    // reference's and is refused, whatever the count says)
    const bool hostHeuristic = (r.flags & PPGPU_F_RIBBON_OVF) && !(r.flags & PPGPU_F_RIBBON_LOST) && nChild <= stride;
    if ((r.flags & (PPGPU_F_DUBINS_ERR | PPGPU_F_RIBBON_LOST)) || ((r.flags & PPGPU_F_RIBBON_OVF) && !hostHeuristic))
        throw std::runtime_error("Edge cost evaluation exceeded a device capacity (flags " + std::to_string(r.flags) + ", child ribbons " +
                                 std::to_string(nChild) + ", parent ribbons " + std::to_string(m_Nodes[source].ribbons.count()) + ")");
    Node c;
    fillChild(c, source, cfgBits, r, childRibbons);
    if (hostHeuristic) {       // Vertex::computeApproxToGo (Vertex.cpp:49-64): the child's heading goes where the callee says yaw
        c.h = c.ribbons.approximateDistanceUntilDone(c.state.x(), c.state.y(), c.state.heading()) / m_Config.maxSpeed() * kTimePenaltyFactor;
        m_Stats.HostHeuristics++;
    }
    return c;
}

// the child vertex of one costed edge, from its record (nothing here throws or counts: prebuildWhileWaiting uses it too)
void GpuAStarPlanner::fillChild(Node& c, int source, unsigned cfgBits, const ppgpu_edge_result& r, const double* childRibbons) const {
    const Node& src = m_Nodes[source];
    const int nChild = (int)((r.info >> 8) & 0xff);
    c.parent = source;
    c.state = State(r.end_x, r.end_y, r.end_heading, r.end_speed, r.end_time);
    c.coverageAllowed = (cfgBits & PPGPU_EDGE_COVERAGE) != 0;
    c.infeasible = (r.flags & PPGPU_F_INFEASIBLE) != 0;
    c.collisionPenalty = r.collision_penalty;
    c.steps = (int)(r.info >> 16);
    c.g = r.g;
    c.h = r.h;
    c.ribbons = RibbonManager(src.ribbons.heuristic(), src.ribbons.turningRadius(), src.ribbons.k());   // the parent's settings; its list is replaced below
    c.ribbons.assign(childRibbons, nChild, r.coverage_completed_time);
    DubinsPath p;
    p.qi[0] = src.state.x(); p.qi[1] = src.state.y(); p.qi[2] = src.state.yaw();
    p.param[0] = r.param[0]; p.param[1] = r.param[1]; p.param[2] = r.param[2];
    p.rho = c.coverageAllowed ? m_Config.coverageTurningRadius() : m_Config.turningRadius();
    p.type = (DubinsPathType)(r.info & 0xff);
    c.wrapper.fill(p, r.end_speed, src.state.time());
    if (!c.infeasible && r.end_time < c.wrapper.getEndTime()) c.wrapper.updateEndTime(r.end_time);   // Edge.cpp:179
}

// While the planner's thread waits for a round trip it has nothing to do: it builds, ahead of their parents' expansion, the children
// of the open vertices whose edges are already costed, best f first — the ones the search pops next.  expand() then only moves them
// into the tree and pushes them.  (Round 4: a cycle was 47 ms of such waits and 28 ms of building and pushing.  Building ALL costed
// children on the context threads was measured first and lost: five times the children, on threads the round trips need.)
void GpuAStarPlanner::prebuildWhileWaiting(GpuContext& busy) {
    static const bool off = std::getenv("PPAMD_NO_PREBUILD") != nullptr;      // (A/B switch)
    if (off || m_Config.visualizations() || m_Queue.empty()) return;
    // the open vertices with costed children, in pop order: the same best-first walk of the heap array as pickBatch
    typedef std::pair<double, size_t> Entry;
    auto worse = [](const Entry& a, const Entry& b) { return a.first > b.first; };
    std::vector<Entry> frontier;
    frontier.emplace_back(m_Queue[0].f, 0);
    size_t visited = 0;
    while (!frontier.empty() && visited < 256) {
        if (busy.idle()) return;
        std::pop_heap(frontier.begin(), frontier.end(), worse);
        const Entry e = frontier.back();
        frontier.pop_back();
        visited++;
        for (size_t c = 2 * e.second + 1; c <= 2 * e.second + 2 && c < m_Queue.size(); c++) {
            frontier.emplace_back(m_Queue[c].f, c);
            std::push_heap(frontier.begin(), frontier.end(), worse);
        }
        const int v = m_Queue[e.second].v;
        auto it = m_Speculated.find(v);
        if (it == m_Speculated.end() || !it->second.ready.empty() || it->second.count == 0) continue;
        Costed& cs = it->second;
        const TripBlock& blk = *cs.block;
        const ppgpu_edge_result* records = reinterpret_cast<const ppgpu_edge_result*>(blk.records.get());
        cs.kids.resize(cs.count);
        cs.ready.assign(cs.count, 0);
        for (size_t i = 0; i < cs.count; i++) {
            const size_t eidx = cs.first + i;
            const ppgpu_edge_result& r = records[eidx];
            // left to expand(): a record that is never pushed, one the planner throws on, one whose heuristic it computes itself
            if (r.flags & (PPGPU_F_INFEASIBLE | PPGPU_F_THROWS | PPGPU_F_DUBINS_ERR | PPGPU_F_RIBBON_LOST | PPGPU_F_RIBBON_OVF)) continue;
            fillChild(cs.kids[i], v, (unsigned)(blk.edges[eidx] >> 56), r, blk.child.get() + eidx * (size_t)blk.stride * 4);
            cs.ready[i] = 1;
        }
    }
}

// ppgpu_cost_edges_host with a child-ribbon stride sized for the parents at hand: children rarely carry more than a few
// ribbons more than their parent, and the child buffer is what crosses PCIe (a 64-ribbon stride is 2 KB per edge).  If some
// child does not fit, the batch is costed again at the device's full per-vertex capacity.
int GpuAStarPlanner::costEdgeList(const std::vector<uint64_t>& edges, int maxParentRibbons, std::vector<ppgpu_edge_result>& res,
                                  std::vector<double>& child) {
    const size_t n = edges.size();
    int stride = std::min(kRibbonStride, maxParentRibbons + 6);
    for (;;) {
        res.resize(n);
        child.assign(n * (size_t)stride * 4, 0.0);
        check(ppgpu_cost_edges_host(m_Ctx->handle(), (int64_t)n, edges.data(), res.data(), child.data(), stride), "ppgpu_cost_edges_host");
        m_Stats.EdgesCosted += n;
        bool retry = false;
        if (stride < kRibbonStride)
            for (size_t i = 0; i < n && !retry; i++)
                retry = (res[i].flags & PPGPU_F_RIBBON_OVF) && (int)((res[i].info >> 8) & 0xff) > stride;
        if (!retry) return stride;
        stride = kRibbonStride;
    }
}

// Vertex::connect(source, state, radius, coverageAllowed) + Edge::computeTrueCost for a batch of targets, then
// pushVertexQueue in the given order.  sampleIndex[i] >= 0 uses a stored sample, otherwise targets[i] is uploaded.
int GpuAStarPlanner::costStateEdges(int source, const std::vector<State>& targets, const std::vector<unsigned>& cfgBits,
                                    const std::vector<long>& sampleIndex) {
    const size_t n = cfgBits.size();
    if (n == 0) return 0;
    ppgpu_ctx* h = m_Ctx->handle();
    std::vector<double> ex, ey, eh;
    std::vector<long> extraSlot(n, -1);
    for (size_t i = 0; i < n; i++) {
        if (sampleIndex[i] < 0) {
            extraSlot[i] = (long)ex.size();
            ex.push_back(targets[i].x()); ey.push_back(targets[i].y()); eh.push_back(targets[i].heading());
        }
    }
    int64_t first = m_NumSamples;
    check(ppgpu_set_extra_targets(h, (int32_t)ex.size(), ex.data(), ey.data(), eh.data(), &first), "ppgpu_set_extra_targets");
    std::vector<uint64_t> edges(n);
    for (size_t i = 0; i < n; i++) {
        uint32_t tgt = (uint32_t)(sampleIndex[i] >= 0 ? sampleIndex[i] : first + extraSlot[i]);
        edges[i] = ppgpu_edge_pack(0, tgt, cfgBits[i]);
    }
    std::vector<ppgpu_edge_result> res;
    std::vector<double> child;
    const int stride = costEdgeList(edges, (int)m_Nodes[source].ribbons.count(), res, child);
    for (size_t i = 0; i < n; i++) {
        {
            const bool cov = (cfgBits[i] & PPGPU_EDGE_COVERAGE) != 0;
            g_dump.write(m_Nodes[source].state, res[i], cov ? m_Config.coverageTurningRadius() : m_Config.turningRadius(), cov);
        }
        addNode(makeChild(source, cfgBits[i], res[i], child.data() + i * (size_t)stride * 4, stride));
        visualizeTrajectory(m_Nodes.back());
        pushVertexQueue((int)m_Nodes.size() - 1);
    }
    return (int)n;
}

// SamplingBasedPlanner::expand (:52-151) for several open vertices in ONE device round trip: every vertex's edges are what
// expand() would build for it alone (nearest ribbon endpoint at each speed and radius, then the k nearest samples per radius
// at each speed, in the order of the reference's heap array), in that order; the children are kept aside, not pushed.
//
// A round trip is a Batch.  It is PACKED on the planning thread (reads the search tree), RUN on its context — inline when the
// planner has one context, on the context's own host thread when it has several, so that the planning thread goes on building and
// pushing children (and packing the next batch) meanwhile — and HARVESTED on the planning thread again.  With several contexts the
// planner keeps the others busy with PREFETCH batches: the next-best open vertices, costed on the chance that the search pops them
// before anything cheaper turns up (it nearly always does: config 5 costs 1.1 edges ahead for every edge it consumes).  Two contexts
// on ONE device are two HIP streams: a round trip is a chain of a dozen latency-bound launches that leave most of the chip idle, and
// two of them overlap almost for free.  What is pushed, and in which order, never depends on any of this.
struct GpuAStarPlanner::Batch {
    std::vector<int> sources;
    std::vector<ppgpu_vertex> verts;
    std::vector<double> pool, nearest;
    int maxParent = 0;
    double heavy = 0;                      // heavyListAllowance of the batch (part of `predicted`)
    int strideFloor = 0, strideUsed = 0, retries = 0;      // (the child stride earlier round trips of this plan() needed; what this one ended with)
    GpuContext* ctx = nullptr;
    bool threaded = false;                 // running on ctx's thread (wait() before anything else touches ctx)
    // filled by run()
    std::shared_ptr<TripBlock> block;
    int64_t n = 0;
    unsigned long edgesCosted = 0;
    double started = 0, took = 0, predicted = 0, samples = 0;
    double posted = 0;                     // when the planner's thread handed it over (started - posted: how long the context's thread took to wake)
};

void GpuAStarPlanner::packBatch(Batch& b) const {
    Lap lap(1);
    const int M = (int)b.sources.size();
    b.verts.resize((size_t)M);
    b.pool.clear();
    b.maxParent = 0;
    for (int i = 0; i < M; i++) {                     // these vertices become the device's open-vertex array
        const Node& n = m_Nodes[b.sources[i]];
        b.verts[i] = makeVertex(n);
        b.verts[i].ribbon_offset = (int32_t)(b.pool.size() / 4);
        b.pool.insert(b.pool.end(), n.ribbons.rows(), n.ribbons.rows() + 4 * (size_t)n.ribbons.count());
        b.maxParent = std::max(b.maxParent, (int)b.verts[i].ribbon_count);
    }
    // nearest point to cover (:64-81): one explicit target per vertex that has one (computed on the host, as in the reference)
    b.nearest.assign((size_t)M * 3, std::nan(""));
    for (int i = 0; i < M; i++) {
        const Node& n = m_Nodes[b.sources[i]];
        if (n.ribbons.done()) continue;
        State s = n.ribbons.getNearestEndpointAsState(n.state);
        if (n.state.distanceTo(s) > m_Config.collisionCheckingIncrement()) {
            b.nearest[3 * i] = s.x(); b.nearest[3 * i + 1] = s.y(); b.nearest[3 * i + 2] = s.heading();
        }
    }
}

// everything else of expand() — k nearest samples per radius, edge list in push order, costing — is one device round trip.
// Touches only the batch and its context (runs on the context's thread when the planner has several).
void GpuAStarPlanner::runBatch(Batch& b, int k) {
    GpuContext& ctx = *b.ctx;
    ppgpu_ctx* h = ctx.handle();
    const int M = (int)b.sources.size();
    const int64_t cap = ppgpu_expand_capacity(M, k);
    int stride = std::min(kRibbonStride, std::max(b.maxParent + 6, b.strideFloor));
    b.started = HostProfile::now();
    for (;;) {
        b.block = ctx.takeTripBlock((size_t)cap, (size_t)cap * stride * 4);      // (the call fills the first n entries; nothing is read beyond them)
        b.block->stride = stride;
        ppgpu_edge_result* res = reinterpret_cast<ppgpu_edge_result*>(b.block->records.get());
        if (ppgpu_expand_host(h, M, b.verts.data(), (int32_t)(b.pool.size() / 4), b.pool.empty() ? nullptr : b.pool.data(), b.nearest.data(), k, &b.n,
                              b.block->edges.get(), res, b.block->child.get(), stride) != PPGPU_OK)
            throw std::runtime_error(std::string("ppgpu_expand_host: ") + ppgpu_last_error());
        b.edgesCosted += (unsigned long)b.n;
        // some child does not fit: again with room for the longest list the records report (round 4: the second pass used to go straight
        // to the device's per-vertex capacity of 64 — 5 MB of child slots to download for 2 500 edges — and such a trip took 5-9 ms where
        // its neighbours took 0.6: the largest under-predictions of the deadline guard)
        int need = 0;
        if (stride < kRibbonStride)
            for (int64_t i = 0; i < b.n; i++)
                if (res[i].flags & PPGPU_F_RIBBON_OVF) need = std::max(need, (int)((res[i].info >> 8) & 0xff));
        if (need <= stride) break;
        stride = std::min(kRibbonStride, need + 2);
        b.retries++;
    }
    b.strideUsed = stride;
    b.took = HostProfile::now() - b.started;
}

// pack + start: inline (one context) or on the context's thread
void GpuAStarPlanner::submitBatch(std::shared_ptr<Batch> bp, GpuContext& ctx) {
    Batch& b = *bp;
    b.ctx = &ctx;
    b.samples = (double)m_NumSamples;
    b.strideFloor = m_StrideFloor;
    b.predicted = ctx.predictTrip(b.samples) + b.heavy;
    packBatch(b);
    const int k = m_Config.branchingFactor();
    for (int v : b.sources) m_InFlightOf[v] = &b;
    b.threaded = m_Ctxs.size() > 1;
    m_InFlight.push_back(std::move(bp));
    g_prof.trips++;
    b.posted = HostProfile::now();
    if (b.threaded) {
        Batch* raw = &b;
        ctx.run([raw, k] { runBatch(*raw, k); });
    } else {
        Lap lap(2);
        try { runBatch(b, k); } catch (...) { dropBatch(&b); throw; }
    }
}

// the batch is over (wait for its thread if it has one): its children go to m_Speculated, its figures to the statistics
void GpuAStarPlanner::harvestBatch(Batch* bp, bool keep) {
    std::shared_ptr<Batch> own;
    for (auto it = m_InFlight.begin(); it != m_InFlight.end(); ++it)
        if (it->get() == bp) { own = std::move(*it); m_InFlight.erase(it); break; }
    Batch& b = *own;
    for (int v : b.sources) m_InFlightOf.erase(v);
    if (b.threaded) {
        Lap lap(2);
        if (keep) prebuildWhileWaiting(*b.ctx);
        b.ctx->wait();                         // rethrows what the round trip threw
    }
    noteOperation(1, b.started, b.predicted, b.took);
    if (b.retries > 0) { m_StrideFloor = std::max(m_StrideFloor, b.strideUsed); m_Stats.Budget.StrideRetries += (unsigned long)b.retries; }
    if (b.threaded) m_Stats.Budget.MaxWakeMs = std::max(m_Stats.Budget.MaxWakeMs, 1e3 * (b.started - b.posted));
    if (b.heavy == 0) b.ctx->noteTrip(b.samples, b.took);      // (a trip with long lists says nothing about the ordinary ones)
    if (b.predicted > 0) m_Ctx->noteExcess(b.took - b.predicted);     // (the guard's margin is the first context's)
    m_Stats.EdgesCosted += b.edgesCosted;
    if (!keep) return;                         // costed against a sample set that is no longer the search's
    Lap lap(3);
    const int M = (int)b.sources.size();
    std::vector<Costed> out((size_t)M);
    for (int i = 0; i < M; i++) out[i].block = b.block;         // an entry even when a vertex has no edges at all
    // the list comes back compacted vertex by vertex, in the order the vertices went in: a vertex's edges are one run of it
    for (int64_t e = 0; e < b.n; e++) {
        const int owner = (int)((b.block->edges[e] >> 32) & 0xffffffu);
        Costed& c = out[(size_t)owner];
        if (c.count == 0) c.first = (size_t)e;
        else if (c.first + c.count != (size_t)e) throw std::runtime_error("ppgpu_expand_host: a vertex's edges are not contiguous in the returned list");
        c.count++;
    }
    for (int i = 0; i < M; i++) m_Speculated[b.sources[i]] = std::move(out[i]);
}

// Round trips that are over but that nobody has asked for yet (a prefetch batch none of whose vertices the search has popped): their
// children join m_Speculated and their contexts are free again.  Without this such a batch held its context for the rest of the
// cycle (round 4: late-mission cycles ran on one of their two contexts from 2 ms on).
void GpuAStarPlanner::harvestFinished() {
    for (size_t i = 0; i < m_InFlight.size();) {
        Batch* b = m_InFlight[i].get();
        if (b->threaded && b->ctx->idle()) harvestBatch(b, true);     // (erases it from m_InFlight)
        else i++;
    }
}

void GpuAStarPlanner::dropBatch(Batch* bp) {
    for (int v : bp->sources) m_InFlightOf.erase(v);
    for (auto it = m_InFlight.begin(); it != m_InFlight.end(); ++it)
        if (it->get() == bp) { m_InFlight.erase(it); break; }
}

// every round trip still running is waited for and thrown away (the sample set changes, or plan() is about to return: nothing of
// this planner may still be running on a context when it does)
void GpuAStarPlanner::drainInFlight() {
    std::exception_ptr first;
    while (!m_InFlight.empty()) {
        try { harvestBatch(m_InFlight.front().get(), false); } catch (...) { if (!first) first = std::current_exception(); }
    }
    m_InFlightOf.clear();
    if (first) std::rethrow_exception(first);
}

// a context with no round trip of this planner on it; when all are busy, the one whose batch was started first is harvested
GpuContext& GpuAStarPlanner::freeContext() {
    for (auto& c : m_Ctxs) {
        bool busy = false;
        for (auto& b : m_InFlight) busy = busy || b->ctx == c.get();
        if (!busy) return *c;
    }
    GpuContext* c = m_InFlight.front()->ctx;
    harvestBatch(m_InFlight.front().get(), true);
    return *c;
}

// expand(source) as the search sees it.  The device answers for `source` and, speculatively, for the open vertices the
// search is most likely to pop next (smallest f first), so that most later calls find their children already costed.  What
// is pushed, and in which order, is exactly what expanding one vertex at a time would push: speculation only changes when
// the arithmetic happens.  Nothing survives a change of the sample set (m_Speculated is cleared by addSamples).
// The open vertices a round trip costs besides `source`: the up to speculation() - 1 entries of smallest f among those whose
// children are not costed yet and that are not goals.  The open list is a binary min-heap on f: walked best-first from its root
// (a small heap of heap positions) the entries come out in non-decreasing f, so the walk touches about as many entries as it
// returns — round 3 scanned and partially sorted the whole list (100 000+ entries late in a cycle: up to 40 of a cycle's 100 ms).
// Ties in f are taken smallest node index first, as before: the walk goes on through every entry that ties with the last one taken.
// Two cuts bound the walk (round 4: one cycle in a few hundred spent 76 ms of its 100 in a single walk — late in a search tens of
// thousands of goals sit in the list at exactly f = horizon, few open vertices are better, and the walk went through the whole
// plateau looking for the rest of its batch): an entry worse than a goal the walk has already passed is never expanded in this pass
// (aStar returns when that goal is popped), so the walk ends there; and it visits at most 16 x the batch + 1 024 entries.  Either way
// the batch is smaller, which changes when arithmetic happens, never what is pushed.
void GpuAStarPlanner::pickBatch(int source, std::vector<int>& batch) {
    const double pick0 = HostProfile::now();
    struct PickTimer {
        Stats::BudgetTrace& b; double t0;
        ~PickTimer() { const double ms = 1e3 * (HostProfile::now() - t0); b.PickMs += ms; b.MaxPickMs = std::max(b.MaxPickMs, ms); }
    } pickTimer{m_Stats.Budget, pick0};
    batch.clear();
    if (source >= 0) batch.push_back(source);          // -1: a prefetch batch, nobody is waiting for any of it
    const size_t want = (size_t)std::max(0, m_Config.speculation() - (source >= 0 ? 1 : 0));
    if (want == 0 || m_Queue.empty()) return;
    typedef std::pair<double, size_t> Entry;                  // (f, position in m_Queue)
    auto worse = [](const Entry& a, const Entry& b) { return a.first > b.first; };
    std::vector<Entry> frontier;
    frontier.emplace_back(m_Queue[0].f, 0);
    std::vector<std::pair<double, int>> cand;                 // (f, node)
    double lastF = 0, goalF = INFINITY;
    size_t visited = 0;
    const size_t maxVisited = 16 * want + 1024;
    while (!frontier.empty()) {
        std::pop_heap(frontier.begin(), frontier.end(), worse);
        const Entry e = frontier.back();
        frontier.pop_back();
        if (cand.size() >= want && e.first > lastF) break;    // everything left is worse than what is already taken
        if (e.first > goalF || ++visited > maxVisited) break; // ... or than a goal already in the list; or the walk has gone far enough
        const int v = m_Queue[e.second].v;
        const bool goal = goalCondition(m_Nodes[v]);
        if (goal) goalF = std::min(goalF, e.first);
        if (!goal && !m_Speculated.count(v) && !m_InFlightOf.count(v)) { cand.emplace_back(e.first, v); lastF = std::max(lastF, e.first); }
        for (size_t c = 2 * e.second + 1; c <= 2 * e.second + 2 && c < m_Queue.size(); c++) {
            frontier.emplace_back(m_Queue[c].f, c);
            std::push_heap(frontier.begin(), frontier.end(), worse);
        }
    }
    std::sort(cand.begin(), cand.end());                      // (f, node index): the order the whole-list scan used to produce
    for (size_t i = 0; i < cand.size() && i < want; i++) batch.push_back(cand[i].second);
}

bool GpuAStarPlanner::expand(int source) {
    visualizeVertex(source, "vertex", true);
    auto it = m_Speculated.find(source);
    if (it == m_Speculated.end()) {
        harvestFinished();
        auto flying = m_Speculated.count(source) ? m_InFlightOf.end() : m_InFlightOf.find(source);
        if (m_Speculated.count(source)) {
            // (a finished prefetch batch just brought it)
        } else if (flying == m_InFlightOf.end()) {
            GpuContext& ctx = freeContext();           // (may harvest a batch: the source can be among its vertices now)
            if (!m_Speculated.count(source)) {
                std::shared_ptr<Batch> b(new Batch());
                {
                    Lap lap(0);
                    pickBatch(source, b->sources);
                }
                // the deadline guard once more, with the clock as it stands now that the batch is chosen (only when the guard is on: a
                // counting clock must see the reference's call sequence)
                {
                    int mp = 0;
                    for (int v : b->sources) mp = std::max(mp, m_Nodes[v].ribbons.count());
                    b->heavy = heavyListAllowance(mp);
                }
                if (m_Config.deadlineGuard() && now() + ctx.predictTrip((double)m_NumSamples) + b->heavy >= m_EndTime - m_Ctx->guardMargin()) return false;
                Batch* mine = b.get();
                submitBatch(std::move(b), ctx);
                // while that one runs: the other contexts take the next-best open vertices (prefetch)
                while (m_InFlight.size() < m_Ctxs.size()) {
                    std::shared_ptr<Batch> p(new Batch());
                    {
                        Lap lap(0);
                        pickBatch(-1, p->sources);
                    }
                    if (p->sources.empty()) break;
                    {
                        int mp = 0;
                        for (int v : p->sources) mp = std::max(mp, m_Nodes[v].ribbons.count());
                        p->heavy = heavyListAllowance(mp);
                    }
                    GpuContext& pc = freeContext();
                    // (a prefetch shares the device with the round trips already in flight: at worst it ends after all of them, one
                    // predicted round trip each — with eight contexts and no such allowance a cycle's last prefetches queued up behind one
                    // another and the drain at the end of the loop took 6 ms)
                    if (m_Config.deadlineGuard() &&
                        now() + (pc.predictTrip((double)m_NumSamples) + p->heavy) * (double)(m_InFlight.size() + 1) >= m_EndTime - m_Ctx->guardMargin()) break;
                    submitBatch(std::move(p), pc);
                }
                harvestBatch(mine, true);
            }
        } else {
            harvestBatch(flying->second, true);        // costed ahead and still running (or just finished): wait for it
        }
        it = m_Speculated.find(source);
        if (it == m_Speculated.end()) throw std::runtime_error("GpuAStarPlanner: a round trip came back without the vertex it was started for");
    }
    Costed costed = std::move(it->second);
    m_Speculated.erase(it);
    Lap lapPush(4);
    const bool watch = m_Config.visualizations();
    const TripBlock& blk = *costed.block;
    const ppgpu_edge_result* records = reinterpret_cast<const ppgpu_edge_result*>(blk.records.get());
    for (size_t e = costed.first; e < costed.first + costed.count; e++) {
        const ppgpu_edge_result& r = records[e];
        const unsigned cfgBits = (unsigned)(blk.edges[e] >> 56);
        // an infeasible edge is never pushed (SamplingBasedPlanner.cpp:8): no vertex is made for it, unless the search is being
        // watched (its sweep is streamed all the same) or the record carries an error (makeChild throws what the reference throws)
        {
            const bool cov = (cfgBits & PPGPU_EDGE_COVERAGE) != 0;
            g_dump.write(m_Nodes[source].state, r, cov ? m_Config.coverageTurningRadius() : m_Config.turningRadius(), cov);
        }
        const bool truncated = (r.flags & PPGPU_F_RIBBON_OVF) && (int)((r.info >> 8) & 0xff) > blk.stride;
        const bool plainInfeasible = (r.flags & PPGPU_F_INFEASIBLE) && !(r.flags & (PPGPU_F_THROWS | PPGPU_F_DUBINS_ERR | PPGPU_F_RIBBON_LOST)) && !truncated;
        if (plainInfeasible && !watch) continue;
        const size_t ki = e - costed.first;
        if (ki < costed.ready.size() && costed.ready[ki]) addNode(std::move(costed.kids[ki]));      // built while the planner waited
        else addNode(makeChild(source, cfgBits, r, blk.child.get() + e * (size_t)blk.stride * 4, blk.stride));
        visualizeTrajectory(m_Nodes.back());   // in the reference each edge streams its sweep, then its vertex is pushed
        pushVertexQueue((int)m_Nodes.size() - 1);
    }
    m_Stats.Expanded++;
    return true;
}

int GpuAStarPlanner::aStar(double endTime) {   // AStarPlanner.cpp:134-148
    int vertex = popVertexQueue();
    const bool guard = m_Config.deadlineGuard();
    // with the guard on even the host-only part of the loop (an expansion whose children were costed ahead: a few microseconds of
    // pushes) stops a little short of the deadline, so that plan() is back BEFORE it
    const double loopEnd = guard ? endTime - kHostMargin : endTime;
    double t;
    while ((t = now()) < loopEnd) {
        if (goalCondition(m_Nodes[vertex])) {
            visualizeVertex(vertex, "vertex", false);
            return vertex;
        }
        // Planner.h:42 "guaranteed to return before timeRemaining has elapsed": an expansion whose children are not costed yet is a
        // device round trip; one that, going by the last ones, would end after the deadline is not started
        // (aimed at the deadline minus GpuContext::guardMargin(): "before", not "at")
        if (guard && !m_Speculated.count(vertex) && t + m_Ctx->predictTrip((double)m_NumSamples) >= endTime - m_Ctx->guardMargin()) {
            m_Stats.DeadlineStops++;
            m_DeadlineStop = true;
            return -1;
        }
        if (!expand(vertex)) {                // (the guard again, after the batch was picked)
            m_Stats.DeadlineStops++;
            m_DeadlineStop = true;
            return -1;
        }
        if (m_Queue.empty()) return -1;
        vertex = popVertexQueue();
    }
    return -1;
}

DubinsPlan GpuAStarPlanner::tracePlan(int v, bool addToStats) {   // Planner.cpp:12-32
    DubinsPlan plan;
    if (v < 0) return plan;
    std::vector<int> branch;
    bool dangerous = false;
    for (int cur = v; m_Nodes[cur].parent >= 0; cur = m_Nodes[cur].parent) {
        branch.push_back(cur);
        if (m_Nodes[cur].collisionPenalty > 0) {
            dangerous = true;
            if (addToStats) m_Stats.PlanCollisionPenalty += m_Nodes[cur].collisionPenalty;
        }
    }
    plan.setDangerous(dangerous);
    for (auto it = branch.rbegin(); it != branch.rend(); it++) plan.append(m_Nodes[*it].wrapper);
    return plan;
}

// ------------------------------------------------------------------------------------------------ plan()
Planner::Stats GpuAStarPlanner::plan(const RibbonManager& ribbonManager, const State& start, PlannerConfig config,
                                     const DubinsPlan& previousPlan, double timeRemaining) {   // AStarPlanner.cpp:12-132
    m_PlanEntry = HostProfile::now();
    m_Config = std::move(config);
    double endTime = timeRemaining + now();
    m_EndTime = endTime;
    m_Config.setStartStateTime(start.time());
    m_RibbonManager = ribbonManager;
    m_RibbonManager.changeHeuristicIfTooManyRibbons();
    if (m_RibbonManager.done()) m_RibbonManager.setCoverageCompletedTime(start.time());
    m_Stats = Stats();
    m_StartStateTime = start.time();
    m_Nodes.clear();
    // the node array the context keeps between cycles (capacity and pages survive; a regrowth that happens all the same — the
    // first cycles of a process — is counted in Stats::Budget)
    if (m_Nodes.capacity() < m_Ctx->nodeArena.capacity()) { m_Ctx->nodeArena.clear(); m_Nodes.swap(m_Ctx->nodeArena); }
    if (m_Nodes.capacity() < kNodeArenaMin) m_Nodes.reserve(kNodeArenaMin);
    m_Queue.clear();
    m_Speculated.clear();
    drainInFlight();
    m_NumSamples = 0;
    m_StrideFloor = 0;
    m_DeadlineStop = false;
    ppgpu_ctx* h = m_Ctx->handle();
    uint64_t growthsBefore = 0;
    double growthSecondsBefore = 0;
    (void)ppgpu_growth_stats(h, &growthsBefore, &growthSecondsBefore);
    unsigned long orderFallbacksBefore = 0;
    for (const auto& ctx : m_Ctxs) orderFallbacksBefore += (unsigned long)ppgpu_order_fallbacks(ctx->handle());
    uploadWorld(start);

    double minSpeed = m_Config.maxSpeed(), maxSpeed = m_Config.maxSpeed();
    double magnitude = m_Config.maxSpeed() * m_Config.timeHorizon();
    const double* mapExtremes = m_Config.map() ? m_Config.map()->extremes() : Map().extremes();
    double bounds[6];
    bounds[0] = std::fmax(start.x() - magnitude, mapExtremes[0]);
    bounds[1] = std::fmin(start.x() + magnitude, mapExtremes[1]);
    bounds[2] = std::fmax(start.y() - magnitude, mapExtremes[2]);
    bounds[3] = std::fmin(start.y() + magnitude, mapExtremes[3]);
    bounds[4] = minSpeed; bounds[5] = maxSpeed;
    unsigned long seed = (unsigned long)endTime;   // :33
    {
        std::vector<double> rib;
        ribbonsToArray(m_RibbonManager, rib);
        for (const auto& ctx : m_Ctxs)
            check(ppgpu_sampler_init(ctx->handle(), bounds, seed, (int32_t)(rib.size() / 4), rib.empty() ? nullptr : rib.data()), "ppgpu_sampler_init");
    }
    // root (:35-37)
    Node root;
    root.state = start;
    root.state.speed() = m_Config.maxSpeed();
    root.g = 0;
    root.ribbons = m_RibbonManager;
    root.h = root.ribbons.approximateDistanceUntilDone(root.state.x(), root.state.y(), root.state.heading()) / m_Config.maxSpeed() *
             kTimePenaltyFactor;   // Vertex::computeApproxToGo (Vertex.cpp:49-64), once per plan, on the host
    m_Nodes.push_back(root);
    const int startV = 0;
    m_Best = -1;
    std::vector<State> brownPathSamples;
    if (m_Config.useBrownPaths()) brownPathSamples = m_RibbonManager.findNearStatesOnRibbons(start, m_Config.coverageTurningRadius());

    // collision check old plan (:46-59)
    int lastPlanEnd = startV;
    std::vector<int> previousPlanNodes;   // the vertices made from it, for the search dump
    if (!previousPlan.empty()) {
        for (const auto& p : previousPlan.get()) {
            if (p.getEndTime() <= start.time()) continue;
            if (p.getNetTime() == 0) continue;
            const bool cov = p.getRho() == m_Config.coverageTurningRadius();
            const double expectRho = cov ? m_Config.coverageTurningRadius() : m_Config.turningRadius();
            ppgpu_vertex v = makeVertex(m_Nodes[lastPlanEnd]);
            std::vector<double> rib;
            ribbonsToArray(m_Nodes[lastPlanEnd].ribbons, rib);
            check(ppgpu_set_vertices(h, 1, &v, v.ribbon_count, rib.empty() ? nullptr : rib.data()), "ppgpu_set_vertices");
            const size_t before = m_Nodes.size();
            if (p.getRho() != expectRho) {
                // Edge.cpp:78-80: a curve at a radius the configuration no longer has is re-solved to the wrapper's end state
                State s;
                s.time() = p.getEndTime();
                p.sample(s);
                std::vector<State> t{s};
                std::vector<unsigned> c{(s.speed() == m_Config.maxSpeed() ? 0u : PPGPU_EDGE_SLOW)};
                std::vector<long> si{-1};
                const size_t q = m_Queue.size();
                const unsigned long gen = m_Stats.Generated;
                costStateEdges(lastPlanEnd, t, c, si);
                m_Queue.resize(q);   // connect + computeTrueCost only: the reference does not push here
                std::make_heap(m_Queue.begin(), m_Queue.end(), [](const QEntry& a, const QEntry& b) { return a.f > b.f; });
                m_Stats.Generated = gen;
            } else {
                ppgpu_wrapper_edge we{};
                we.vertex = 0;
                we.coverage_allowed = cov ? 1 : 0;
                const DubinsPath& dp = p.unwrap();
                for (int i = 0; i < 3; i++) { we.qi[i] = dp.qi[i]; we.param[i] = dp.param[i]; }
                we.rho = dp.rho; we.type = (int32_t)dp.type;
                we.speed = p.getSpeed(); we.start_time = p.curveStartTime(); we.end_time = p.getEndTime();
                ppgpu_edge_result r;
                std::vector<double> child((size_t)kRibbonStride * 4);
                check(ppgpu_cost_wrapper_edges_host(h, 1, &we, &r, child.data(), kRibbonStride), "ppgpu_cost_wrapper_edges_host");
                m_Stats.EdgesCosted++;
                g_dump.write(m_Nodes[lastPlanEnd].state, r, dp.rho, cov);
                if (r.flags & PPGPU_F_THROWS) throw std::runtime_error("Invalid time in sample for Dubins path (previous plan)");
                const int nChild = (int)((r.info >> 8) & 0xff);
                const bool hostHeuristic = (r.flags & PPGPU_F_RIBBON_OVF) && !(r.flags & PPGPU_F_RIBBON_LOST) && nChild <= kRibbonStride;
                if ((r.flags & (PPGPU_F_DUBINS_ERR | PPGPU_F_RIBBON_LOST)) || ((r.flags & PPGPU_F_RIBBON_OVF) && !hostHeuristic))
                    throw std::runtime_error("Edge cost evaluation exceeded a device capacity");
                Node c;
                c.parent = lastPlanEnd;
                c.state = State(r.end_x, r.end_y, r.end_heading, r.end_speed, r.end_time);
                c.coverageAllowed = cov;
                c.infeasible = (r.flags & PPGPU_F_INFEASIBLE) != 0;
                c.collisionPenalty = r.collision_penalty;
                c.steps = (int)(r.info >> 16);
                c.g = r.g; c.h = r.h;
                c.ribbons = m_Nodes[lastPlanEnd].ribbons;
                c.ribbons.assign(child.data(), nChild, r.coverage_completed_time);
                if (hostHeuristic) {
                    c.h = c.ribbons.approximateDistanceUntilDone(c.state.x(), c.state.y(), c.state.heading()) / m_Config.maxSpeed() * kTimePenaltyFactor;
                    m_Stats.HostHeuristics++;
                }
                c.wrapper = p;
                if (!c.infeasible && r.end_time < c.wrapper.getEndTime()) c.wrapper.updateEndTime(r.end_time);
                addNode(std::move(c));
            }
            if (m_Nodes.size() == before) break;
            lastPlanEnd = (int)m_Nodes.size() - 1;
            previousPlanNodes.push_back(lastPlanEnd);
            if (m_Nodes[lastPlanEnd].infeasible) {
                lastPlanEnd = startV;
                break;
            }
            if (goalCondition(m_Nodes[lastPlanEnd])) break;
        }
    }

    // big loop (:61-119)
    m_Stats.Budget.PrologueMs = 1e3 * (HostProfile::now() - m_PlanEntry);
    const bool guard = m_Config.deadlineGuard();
    double tPoll;
    while ((tPoll = now()) < (guard ? endTime - kHostMargin : endTime)) {
        m_Queue.clear();
        if (m_Best >= 0 && m_Nodes[m_Best].f() <= m_Nodes[startV].f()) {
            *m_Config.output() << "Found best possible plan, assuming heuristic admissibility" << std::endl;
            break;
        }
        if (m_Config.visualizations()) {
            // :67-92.  The reference re-costs the previous plan here when it visualises (same arithmetic, fresh vertices); the
            // vertices costed above are shown instead.
            visualizeVertex(startV, "start", false);
            for (int v : previousPlanNodes) {
                visualizeTrajectory(m_Nodes[v]);
                visualizeVertex(v, "lastPlanEnd", false);
            }
            m_Config.visualizationStream() << "Incumbent f-value: " << (m_Best >= 0 ? m_Nodes[m_Best].f() : 0) << std::endl;
            m_Config.visualizationStream() << m_RibbonManager.dumpRibbons() << "End Ribbons" << std::endl;
        }
        pushVertexQueue(startV);
        if (lastPlanEnd != startV) pushVertexQueue(lastPlanEnd);
        // expandToCoverSpecificSamples(startV, brownPathSamples, ..., true) (:150-162)
        if (!brownPathSamples.empty() && m_Config.coverageTurningRadius() > 0) {
            // (a device round trip like any other: after the first pass it is not started when it cannot end in time)
            const double brownPredicted = m_Ctx->predictTrip((double)m_NumSamples);
            if (guard && m_Stats.Iterations > 0 && tPoll + brownPredicted >= endTime - m_Ctx->guardMargin()) {
                m_Stats.DeadlineStops++;
                break;
            }
            const double brown0 = HostProfile::now();
            ppgpu_vertex v = makeVertex(m_Nodes[startV]);
            std::vector<double> rib;
            ribbonsToArray(m_Nodes[startV].ribbons, rib);
            check(ppgpu_set_vertices(h, 1, &v, v.ribbon_count, rib.empty() ? nullptr : rib.data()), "ppgpu_set_vertices");
            std::vector<State> t;
            std::vector<unsigned> c;
            std::vector<long> si;
            for (const State& s : brownPathSamples) {
                t.push_back(s); c.push_back(PPGPU_EDGE_COVERAGE); si.push_back(-1);
                t.push_back(s); c.push_back(PPGPU_EDGE_COVERAGE | PPGPU_EDGE_SLOW); si.push_back(-1);
            }
            costStateEdges(startV, t, c, si);
            noteOperation(3, brown0, m_Stats.Iterations > 0 ? brownPredicted : 0.0, HostProfile::now() - brown0);
        }
        // first iteration: initialSamples; afterwards double them (:101-102)
        const long moreSamples = m_NumSamples < m_Config.initialSamples() ? m_Config.initialSamples() : m_NumSamples;
        // the deadline guard: a doubling that, with the one round trip that makes it worth anything, cannot end in time is not started
        if (guard && m_Stats.Iterations > 0 &&
            tPoll + m_Ctx->predictDoubling((double)moreSamples) + m_Ctx->predictTrip((double)(m_NumSamples + moreSamples)) >= endTime - m_Ctx->guardMargin()) {
            m_Stats.DeadlineStops++;
            break;
        }
        addSamples(moreSamples);
        visualizeSamples();
        int v = aStar(endTime);
        drainInFlight();                       // prefetched round trips still running: the next pass changes the sample set (and uses the contexts)
        if (m_Best < 0 || (v >= 0 && m_Nodes[v].f() + 0.0 < m_Nodes[m_Best].f())) {
            m_Best = v;
            if (v >= 0 && m_Config.visualizations()) {   // :113-116
                visualizePlan(tracePlan(v, false));
                visualizeVertex(v, "goal", false);
            }
        }
        if (v >= 0 && m_Stats.FirstGoalIteration < 0) m_Stats.FirstGoalIteration = (long)m_Stats.Iterations;
        m_Stats.Iterations++;
        // aStar() left because its next round trip could not end in time: neither can another pass of this loop (which would begin with
        // the Brown-path round trip and a sample doubling) — the reference's loop would find the clock past endTime at this point
        if (m_DeadlineStop) break;
    }
    drainInFlight();                           // nothing of this planner runs on a context once plan() has returned
    m_Stats.Budget.LoopEndMs = 1e3 * (HostProfile::now() - m_PlanEntry);
    m_Stats.Budget.MarginMs = guard ? 1e3 * m_Ctx->guardMargin() : 0.0;
    m_Stats.Samples = (unsigned long)m_NumSamples;
    if (m_Best < 0) {
        *m_Config.output() << "Failed to find a plan" << std::endl;
    } else {
        m_Stats.PlanFValue = m_Nodes[m_Best].f();
        m_Stats.PlanDepth = (unsigned long)depth(m_Best);
        m_Stats.PlanTimePenalty = (m_Nodes[m_Best].state.time() - m_StartStateTime) * kTimePenaltyFactor;
        m_Stats.PlanHValue = m_Nodes[m_Best].h;
        m_Stats.Plan = tracePlan(m_Best);
    }
    for (const auto& ctx : m_Ctxs) m_Stats.OrderFallbacks += (unsigned long)ppgpu_order_fallbacks(ctx->handle());
    m_Stats.OrderFallbacks -= orderFallbacksBefore;
    g_prof.report("plan()");
    {
        uint64_t growths = 0;
        double growthSeconds = 0;
        if (ppgpu_growth_stats(h, &growths, &growthSeconds) == PPGPU_OK) {
            m_Stats.Budget.DeviceGrowths = (unsigned long)(growths - growthsBefore);
            m_Stats.Budget.DeviceGrowthMs = 1e3 * (growthSeconds - growthSecondsBefore);
        }
    }
    m_Stats.Budget.TotalMs = 1e3 * (HostProfile::now() - m_PlanEntry);
    return m_Stats;
}

// ------------------------------------------------------------------------------------------------ ShardedIteration
ShardedIteration::ShardedIteration(std::vector<std::shared_ptr<GpuContext>> ctxs) : m_Ctxs(std::move(ctxs)) {
    if (m_Ctxs.empty()) m_Ctxs.push_back(GpuContext::shared(0));
    const size_t D = m_Ctxs.size();
    m_Keys.assign(D, nullptr); m_Records.assign(D, nullptr); m_RecordCap.assign(D, 0);
    bool distinct = true;
    for (size_t i = 0; i < D; i++)
        for (size_t j = 0; j < i; j++) distinct = distinct && m_Ctxs[i]->device() != m_Ctxs[j]->device();
    try {
        // a context that already carries a communicator belongs to someone else's collective (GpuContext::shared handles are
        // process-level): refuse rather than tear it down under its owner
        if (distinct)
            for (auto& c : m_Ctxs) {
                int32_t world = 0, rank = 0;
                if (ppgpu_comm_info(c->handle(), &world, &rank) == PPGPU_OK)
                    throw std::runtime_error("ShardedIteration: device " + std::to_string(c->device()) + " already has a communicator (" +
                                             std::to_string(world) + " ranks) that this object did not create");
            }
        for (size_t i = 0; i < D; i++)
            if (ppgpu_device_alloc(m_Ctxs[i]->handle(), 16, &m_Keys[i]) != PPGPU_OK) throw std::runtime_error(std::string("ppgpu_device_alloc: ") + ppgpu_last_error());
        if (distinct) {
            std::vector<ppgpu_ctx*> hs;
            for (auto& c : m_Ctxs) hs.push_back(c->handle());
            // (all handles get a communicator or none does: ppgpu_comm_init_all cleans up after itself)
            if (ppgpu_comm_init_all(hs.data(), (int32_t)D) != PPGPU_OK) throw std::runtime_error(std::string("ppgpu_comm_init_all: ") + ppgpu_last_error());
            m_Rccl = true;
        }
    } catch (...) {
        release();        // the destructor does not run for a constructor that throws
        throw;
    }
}

void ShardedIteration::release() {
    for (size_t i = 0; i < m_Ctxs.size(); i++) {
        if (m_Rccl) ppgpu_comm_destroy(m_Ctxs[i]->handle());      // only the communicators this object created
        ppgpu_device_free(m_Ctxs[i]->handle(), m_Keys[i]);
        ppgpu_device_free(m_Ctxs[i]->handle(), m_Records[i]);
        m_Keys[i] = nullptr; m_Records[i] = nullptr; m_RecordCap[i] = 0;
    }
    m_Rccl = false;
}

ShardedIteration::~ShardedIteration() { release(); }

ShardedIteration::Result ShardedIteration::run(const RibbonManager& ribbonManager, const State& start, const PlannerConfig& config, unsigned long seed,
                                               int64_t attempts) {
    const size_t D = m_Ctxs.size();
    RibbonManager rm = ribbonManager;
    rm.changeHeuristicIfTooManyRibbons();                                  // AStarPlanner.cpp:18
    uploadSnapshot(m_Ctxs, config, rm, start.time());
    // the sampling box and the root vertex of AStarPlanner::plan (:27-37)
    const double magnitude = config.maxSpeed() * config.timeHorizon();
    const double* ext = config.map() ? config.map()->extremes() : Map().extremes();
    const double bounds[6] = {std::fmax(start.x() - magnitude, ext[0]), std::fmin(start.x() + magnitude, ext[1]), std::fmax(start.y() - magnitude, ext[2]),
                              std::fmin(start.y() + magnitude, ext[3]), config.maxSpeed(), config.maxSpeed()};
    std::vector<double> rib(rm.rows(), rm.rows() + 4 * (size_t)rm.count());
    ppgpu_vertex root{};
    root.x = start.x(); root.y = start.y(); root.heading = start.heading(); root.speed = config.maxSpeed(); root.time = start.time();
    root.g = 0; root.coverage_completed_time = rm.done() ? start.time() : rm.coverageCompletedTime();
    root.ribbon_offset = 0; root.ribbon_count = (int32_t)rm.count();
    // contiguous slices of the batch, sizes differing by at most one (path_planner_amd/sharding.py: shard_attempts)
    const int64_t base = attempts / (int64_t)D, extra = attempts % (int64_t)D;
    const int64_t edgesPerShard = 4 * (base + (extra ? 1 : 0));            // global edge ids: shard * edgesPerShard + local index
    Result out;
    out.kept.assign(D, 0);
    std::vector<uint64_t> keys(2 * D, ~0ull);
    auto check = [](int rc, const char* what) { if (rc != PPGPU_OK) throw std::runtime_error(std::string(what) + ": " + ppgpu_last_error()); };
    // Phase 1, per shard, no collective in it: skip, draw, cost, reduce to the shard's best key (left on the device).
    auto local = [&](size_t d) {
        ppgpu_ctx* h = m_Ctxs[d]->handle();
        if ((int)d == failShardForTest) throw std::runtime_error("ShardedIteration: injected failure in shard " + std::to_string(d));
        const int64_t lo = (int64_t)d * base + std::min<int64_t>((int64_t)d, extra), n = base + ((int64_t)d < extra ? 1 : 0);
        check(ppgpu_set_vertices(h, 1, &root, root.ribbon_count, rib.empty() ? nullptr : rib.data()), "ppgpu_set_vertices");
        check(ppgpu_sampler_init(h, bounds, seed, (int32_t)(rib.size() / 4), rib.empty() ? nullptr : rib.data()), "ppgpu_sampler_init");
        if (lo > 0) check(ppgpu_sampler_skip(h, lo), "ppgpu_sampler_skip");   // the slices of the lower ranks
        int64_t kept = 0;
        for (int64_t left = n; left > 0; left -= 524288) check(ppgpu_sampler_add(h, std::min<int64_t>(left, 524288), &kept), "ppgpu_sampler_add");
        out.kept[d] = kept;
        const size_t need = (size_t)std::max<int64_t>(4 * kept, 1) * sizeof(ppgpu_edge_result);
        if (need > m_RecordCap[d]) {
            check(ppgpu_device_free(h, m_Records[d]), "ppgpu_device_free");
            m_Records[d] = nullptr; m_RecordCap[d] = 0;
            check(ppgpu_device_alloc(h, need, &m_Records[d]), "ppgpu_device_alloc");
            m_RecordCap[d] = need;
        }
        if (kept > 0) check(ppgpu_cost_edges_dense(h, 0, 1, 0, kept, 0xF, (ppgpu_edge_result*)m_Records[d], nullptr, 0), "ppgpu_cost_edges_dense");
        check(ppgpu_best_edge(h, 4 * kept, (const ppgpu_edge_result*)m_Records[d], 0, (uint64_t)d * (uint64_t)edgesPerShard, (uint64_t*)m_Keys[d]), "ppgpu_best_edge");
    };
    // Phase 2: the one collective of the iteration, entered only when EVERY shard got through phase 1 — a rank that never joins
    // an all-gather leaves the others waiting on their streams for ever, with no error anywhere — then the key comes home.
    auto combine = [&](size_t d) {
        ppgpu_ctx* h = m_Ctxs[d]->handle();
        if (m_Rccl) check(ppgpu_allreduce_best(h, nullptr, (uint64_t*)m_Keys[d]), "ppgpu_allreduce_best");
        check(ppgpu_device_read(h, &keys[2 * d], m_Keys[d], 16), "ppgpu_device_read");
    };
    // each phase on every context's own host thread; all of them joined before anything is thrown
    auto onAll = [&](const std::function<void(size_t)>& phase, bool abortOthersOnFailure) {
        if (D == 1) { phase(0); return; }
        for (size_t d = 0; d < D; d++) m_Ctxs[d]->run([&, d] { phase(d); });
        std::exception_ptr first;
        for (size_t d = 0; d < D; d++) {
            try { m_Ctxs[d]->wait(); } catch (...) {
                if (!first) {
                    first = std::current_exception();
                    // a rank failed INSIDE the collective phase: the ranks still waiting for it are released by aborting their
                    // communicators (ncclCommAbort is made for exactly this); the object is unusable for RCCL afterwards
                    if (abortOthersOnFailure && m_Rccl) {
                        for (size_t o = 0; o < D; o++) if (o != d) (void)ppgpu_comm_abort(m_Ctxs[o]->handle());
                        (void)ppgpu_comm_abort(m_Ctxs[d]->handle());
                        m_Rccl = false;
                    }
                }
            }
        }
        if (first) std::rethrow_exception(first);
    };
    onAll(local, false);
    onAll(combine, true);
    for (size_t d = 0; d < D; d++) out.edges += 4 * out.kept[d];
    if (m_Rccl) {
        int32_t world = 0, rank = 0;
        if (ppgpu_comm_info(m_Ctxs[0]->handle(), &world, &rank) == PPGPU_OK) out.rcclRanks = world;
        out.agreed = true;
        for (size_t d = 1; d < D; d++) out.agreed = out.agreed && keys[2 * d] == keys[0] && keys[2 * d + 1] == keys[1];
        out.fBits = keys[0]; out.edge = keys[1];
    } else {
        // contexts that share a device: the lexicographic min ppgpu_key_min computes, on the host
        out.agreed = true;
        for (size_t d = 0; d < D; d++)
            if (keys[2 * d] < out.fBits || (keys[2 * d] == out.fBits && keys[2 * d + 1] < out.edge)) { out.fBits = keys[2 * d]; out.edge = keys[2 * d + 1]; }
    }
    if (out.edge != ~0ull) {
        std::memcpy(&out.f, &out.fBits, sizeof(double));
        out.shard = (int)(out.edge / (uint64_t)edgesPerShard);
    }
    return out;
}

}  // namespace ppamd
