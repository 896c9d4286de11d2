// Planner / GpuAStarPlanner — the drop-in seam.
//
// Planner mirrors /root/reference/path_planner/src/planner/Planner.h:17-80 (same Stats fields, same plan() signature);
// GpuAStarPlanner is AStarPlanner (AStarPlanner.cpp:12-148) + SamplingBasedPlanner::expand (SamplingBasedPlanner.cpp:52-151)
// with every sample drawn, every Dubins length computed and every edge costed on the device through include/ppgpu.h.
// The A* control flow, the open list and the clock polling stay on the host in the reference's order, so that with the same
// injected clock and seed the same vertices are expanded in the same order.
#pragma once
#include <algorithm>
#include <condition_variable>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>
#include <unordered_map>
#include <vector>

#include "DubinsWrapper.h"
#include "RibbonManager.h"
#include "World.h"

struct ppgpu_ctx;

struct ppgpu_edge_result;   // include/ppgpu.h

namespace ppamd {

class Planner {
public:
    struct Stats {
        unsigned long Samples = 0;
        unsigned long Generated = 0;
        unsigned long Expanded = 0;
        unsigned long Iterations = 0;
        double PlanFValue = 0;
        double PlanCollisionPenalty = 0;
        double PlanTimePenalty = 0;
        double PlanHValue = 0;
        unsigned long PlanDepth = 0;
        DubinsPlan Plan;
        // extra, for parity checks against the CPU oracle
        long FirstGoalIteration = -1;
        unsigned long EdgesCosted = 0;
        unsigned long HostHeuristics = 0;     // children whose ribbon list exceeded the device's TSP enumeration: h computed on the host
        unsigned long DeadlineStops = 0;      // round trips / sample doublings not started because they could not end before the deadline
        unsigned long OrderFallbacks = 0;     // (vertex, radius) lists whose push order the device could not replay (ppgpu_order_fallbacks)
    };
    Planner();
    virtual ~Planner() = default;
    virtual Stats plan(const RibbonManager& ribbonManager, const State& start, PlannerConfig config,
                       const DubinsPlan& previousPlan, double timeRemaining);
    void setConfig(PlannerConfig config) { m_Config = std::move(config); }

protected:
    double now() const { return m_Config.now(); }
    PlannerConfig m_Config;
    Stats m_Stats;
};

// Process-level device handle: stream, persistent buffers (the reference builds a new planner every cycle,
// executive.cpp:85-90, so nothing device-side may live in the planner object).  Throws std::runtime_error.
// shared(device) hands out ONE context per device for the life of the process: the cache holds a strong reference, so the
// buffers a cycle grew (samples, workspace, pinned staging) are there for the next cycle's planner; releaseShared() drops them.
// Each context owns one host thread bound to its device: a planner that is given several contexts runs its per-device work
// there (run()), so the devices of a node cost their parts of a batch at the same time.
class GpuContext {
public:
    explicit GpuContext(int device = 0);
    ~GpuContext();
    GpuContext(const GpuContext&) = delete;
    GpuContext& operator=(const GpuContext&) = delete;
    ppgpu_ctx* handle() const { return m_Handle; }
    int device() const { return m_Device; }
    static std::shared_ptr<GpuContext> shared(int device = 0);   // one per device per process, kept until releaseShared()
    static std::vector<std::shared_ptr<GpuContext>> shared(const std::vector<int>& devices);
    static void releaseShared();
    // run `job` on this context's thread; wait() blocks until it has finished and rethrows what it threw
    void run(std::function<void()> job);
    void wait();

    // What this device's recent round trips and sample doublings took (wall seconds), kept with the context so that the first
    // cycle of a new planner already knows: the deadline guard (PlannerConfig::deadlineGuard) predicts the next one from them.
    // Written and read by the planning thread only.
    struct Observed { double samples = 0, seconds = 0; };
    Observed trips[4];        // the last round trips of expandBatch: sample count, duration
    int tripSlot = 0;
    Observed doubling;        // the last addSamples: attempts, duration
    void noteTrip(double samples, double seconds) { trips[tripSlot] = {samples, seconds}; tripSlot = (tripSlot + 1) % 4; }
    // a round trip over `samples` samples: no longer than the recent ones scaled up to that sample count (their cost grows
    // less than linearly), plus a margin
    double predictTrip(double samples) const {
        double worst = 0;
        for (const Observed& o : trips)
            if (o.seconds > 0) worst = std::max(worst, o.seconds * std::max(1.0, samples / std::max(1.0, o.samples)));
        return worst * 1.15 + 1e-4;
    }
    double predictDoubling(double attempts) const {
        if (doubling.seconds <= 0) return 0;
        return doubling.seconds * std::max(1.0, attempts / std::max(1.0, doubling.samples)) * 1.15 + 1e-4;
    }

private:
    ppgpu_ctx* m_Handle = nullptr;
    int m_Device = 0;
    std::thread m_Thread;
    std::mutex m_Mutex;
    std::condition_variable m_Wake;
    std::function<void()> m_Job;
    bool m_Busy = false, m_Quit = false;
    std::exception_ptr m_Error;
    void serve();
};

class GpuAStarPlanner : public Planner {
public:
    explicit GpuAStarPlanner(std::shared_ptr<GpuContext> ctx = GpuContext::shared()) : m_Ctx(ctx), m_Ctxs{ctx} {}
    // several devices of one node: world and samples are replicated on each (the sampler stream is deterministic: every device
    // draws the same samples itself), the open vertices of a batch are dealt across them, records come back to the host search
    explicit GpuAStarPlanner(std::vector<std::shared_ptr<GpuContext>> ctxs) : m_Ctx(ctxs.at(0)), m_Ctxs(std::move(ctxs)) {}
    Stats plan(const RibbonManager& ribbonManager, const State& start, PlannerConfig config, const DubinsPlan& previousPlan,
               double timeRemaining) override;

    // one search-tree node: what Vertex + its parent Edge hold in the reference (Vertex.h:180-187, Edge.h:133-143)
    struct Node {
        State state;
        int parent = -1;
        RibbonManager ribbons;
        double g = -1, h = -1;
        bool coverageAllowed = false;
        bool infeasible = false;
        double collisionPenalty = 0;
        int steps = 0;             // collision-check steps the edge's sweep executed (visualisation only)
        DubinsWrapper wrapper;     // parent edge's curve
        double f() const { return g + h; }
    };

private:
    std::shared_ptr<GpuContext> m_Ctx;                    // device 0 of this planner: sampling read-back, wrapper edges, explicit targets
    std::vector<std::shared_ptr<GpuContext>> m_Ctxs;      // all of them (m_Ctxs[0] == m_Ctx)
    std::vector<Node> m_Nodes;
    std::vector<int> m_Queue;      // binary heap of node indices, min f (AStarPlanner.cpp:6-10)
    int m_Best = -1;
    double m_StartStateTime = 0;
    RibbonManager m_RibbonManager;
    long m_NumSamples = 0;
    // Edges of open vertices costed ahead of their expansion, as the device returned them: a child Node is built only when the
    // search expands the parent, and only for edges it would push (an infeasible edge never becomes a vertex).
    struct Costed {
        std::vector<unsigned> cfgBits;
        std::vector<unsigned char> records;   // ppgpu_edge_result each
        std::vector<double> childRibbons;     // `stride` x 4 doubles per edge
        int stride = 0;
    };
    std::unordered_map<int, Costed> m_Speculated;

    void uploadWorld(const State& start);
    void pushVertexQueue(int v);
    int popVertexQueue();
    bool goalCondition(const Node& v) const;
    void expand(int source);
    void expandBatch(const std::vector<int>& sources);
    void expandOn(GpuContext& ctx, const std::vector<int>& sources, std::vector<std::pair<int, Costed>>& out, unsigned long& edgesCosted) const;
    int costEdgeList(const std::vector<uint64_t>& edges, int maxParentRibbons, std::vector<::ppgpu_edge_result>& res, std::vector<double>& child);
    Node makeChild(int source, unsigned cfgBits, const ::ppgpu_edge_result& r, const double* childRibbons, int stride);
    int aStar(double endTime);
    void addSamples(long n);
    int depth(int v) const;
    DubinsPlan tracePlan(int v, bool addToStats = true);
    void check(int rc, const char* what) const;
    // the reference's search dump (SamplingBasedPlanner.cpp:210-238, Edge.cpp:122-143); no-ops unless the config enables it
    void visualizeVertex(int v, const char* tag, bool expanded);
    void visualizeTrajectory(const Node& child);
    void visualizePlan(const DubinsPlan& plan);
    void visualizeSamples();
    int costStateEdges(int source, const std::vector<State>& targets, const std::vector<unsigned>& cfgBits,
                       const std::vector<long>& sampleIndex);
};

}  // namespace ppamd
