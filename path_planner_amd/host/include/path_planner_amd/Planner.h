// Planner / GpuAStarPlanner — the drop-in seam.
//
// Planner mirrors /root/reference/path_planner/src/planner/Planner.h:17-80 (same Stats fields, same plan() signature);
// GpuAStarPlanner is AStarPlanner (AStarPlanner.cpp:12-148) + SamplingBasedPlanner::expand (SamplingBasedPlanner.cpp:52-151)
// with every sample drawn, every Dubins length computed and every edge costed on the device through include/ppgpu.h.
// The A* control flow, the open list and the clock polling stay on the host in the reference's order, so that with the same
// injected clock and seed the same vertices are expanded in the same order.
#pragma once
#include <algorithm>
#include <cmath>
#include <atomic>
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
        // Where the budget of this call went, for the time contract ("guaranteed to return before timeRemaining has elapsed",
        // Planner.h:42): wall milliseconds from plan()'s entry, measured with the steady clock whatever clock is injected.
        struct BudgetTrace {
            double PrologueMs = 0;            // entry -> first pass of the big loop (world upload, sampler, root, previous plan)
            double LoopEndMs = 0;             // when the big loop was left
            double TotalMs = 0;               // when plan() returned
            int LastOpKind = 0;               // the last device operation started: 0 none, 1 round trip, 2 sample doubling, 3 Brown-path re-cost
            double LastOpStartMs = 0, LastOpPredictedMs = 0, LastOpActualMs = 0;
            double MarginMs = 0;              // what the deadline guard kept clear of the deadline
            double MaxTripMs = 0;             // the slowest round trip of the call
            double WorstUnderPredictionMs = 0;   // max over guarded operations of (actual - predicted)
            unsigned long NodeRegrowths = 0;  // reallocations of the search tree's node array inside the call ...
            double NodeRegrowthMs = 0;        // ... and what they took
            unsigned long DeviceGrowths = 0;  // device / pinned buffers the device library grew inside the call (ppgpu_growth_stats) ...
            double DeviceGrowthMs = 0;        // ... and what that took
            unsigned long RoundTrips = 0;
            unsigned long StrideRetries = 0;  // round trips repeated because a child's ribbon list did not fit the stride they were costed with
            double MaxWakeMs = 0;             // the longest a context's thread took to start a round trip handed to it
            double PickMs = 0, MaxPickMs = 0;  // choosing the open vertices of the round trips (pickBatch): total and the longest single walk
            bool GridUploaded = false;        // the occupancy grid went to the device in this call (false: the device already held this map)
        } Budget;
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

// One search-tree node: what Vertex + its parent Edge hold in the reference (Vertex.h:180-187, Edge.h:133-143)
struct SearchNode {
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

// What one device round trip of GpuAStarPlanner brought back — edge descriptors, records, child ribbons, compacted vertex by
// vertex — kept whole: the children of an open vertex costed ahead of its expansion are a RANGE of one block, not copies.  Blocks
// are recycled through their context's pool (a 300 KB buffer from malloc is a fresh mmap, page faults and all, every time).
struct TripBlock {
    std::unique_ptr<uint64_t[]> edges;
    std::unique_ptr<unsigned char[]> records;     // ppgpu_edge_result (128 bytes) each
    std::unique_ptr<double[]> child;              // `stride` x 4 doubles per edge
    size_t edgeCap = 0, childCap = 0;
    int stride = 0;
    void reserve(size_t nEdges, size_t childDoubles);
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
    // one per (device, lane) per process, kept until releaseShared().  Lane 0 is "the" context of a device; further lanes are further
    // streams (with their own buffers) on the same device: a planner given {shared(0, 0), shared(0, 1)} keeps two round trips in flight.
    static std::shared_ptr<GpuContext> shared(int device = 0, int lane = 0);
    static std::vector<std::shared_ptr<GpuContext>> shared(const std::vector<int>& devices);
    static void releaseShared();
    // run `job` on this context's thread; wait() blocks until it has finished and rethrows what it threw
    void run(std::function<void()> job);
    void wait();
    bool idle();               // no job of run() is running (its result, or what it threw, is waiting for wait())

    // What this device's recent round trips and sample doublings took (wall seconds), kept with the context so that the first
    // cycle of a new planner already knows: the deadline guard (PlannerConfig::deadlineGuard) predicts the next one from them.
    // Written and read by the planning thread only.
    struct Observed { double samples = 0, seconds = 0; };
    static constexpr int kTrips = 8;
    Observed trips[kTrips];   // the last round trips of expandBatch: sample count, duration
    int tripSlot = 0;
    Observed doubling;        // the last addSamples: attempts, duration
    void noteTrip(double samples, double seconds) { trips[tripSlot] = {samples, seconds}; tripSlot = (tripSlot + 1) % kTrips; }
    // a round trip over `samples` samples: no longer than the recent ones scaled up to that sample count (their cost grows
    // less than linearly), plus 15 %
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
    // How far a round trip overran its prediction, remembered for a while (round 4: one trip in a few hundred takes 0.8-3 ms longer
    // than any of the eight before it scaled to its sample count would suggest — children with long ribbon lists whose heuristic one
    // wavefront enumerates, a second pass at a larger child stride — and when it happened to be a cycle's last the cycle ended
    // 0.05-0.4 ms late): the largest excess seen, decaying by 0.1 % per round trip (half of it is left four cycles later), at most 3 ms.
    double tailExcess = 0;
    void noteExcess(double seconds) { tailExcess = std::max(tailExcess * 0.999, std::min(seconds, 3e-3)); }
    // What the guard keeps clear of the deadline on top of its predictions, so that "before" holds and not "at": half a
    // millisecond (the return path: tracing the plan, the caller's clock read), or twice the spread of the recent round trips,
    // or the remembered excess above, whichever is largest.
    double guardMargin() const {
        double n = 0, sum = 0, sq = 0;
        for (const Observed& o : trips)
            if (o.seconds > 0) { n += 1; sum += o.seconds; sq += o.seconds * o.seconds; }
        double sigma = 0;
        if (n >= 2) { const double mean = sum / n; sigma = std::sqrt(std::max(0.0, sq / n - mean * mean)); }
        return std::max(std::max(5e-4, 2 * sigma), tailExcess);
    }
    // The search tree's node array lives HERE between plan() calls (a planner takes it at the start of plan() and hands it back,
    // emptied, when it is destroyed): its capacity — and its pages — survive the cycle.  Measured in round 4 (Stats::Budget): a
    // std::vector that doubles at 65 536 nodes moves 16 MB and faults in 31 MB of fresh pages, 6.9 ms inside a 100 ms budget —
    // round 3's unexplained +6 ms cycles.
    std::vector<SearchNode> nodeArena;
    // Which occupancy map the device holds (identity + Map::version): the planner of the next cycle uploads the grid only when
    // the Executive has been given another map.  nullptr / 0: nothing cached.
    const void* gridOf = nullptr;
    unsigned long gridVersion = 0;
    // round-trip result blocks not referred to by any planner any more are handed out again (used by this context's thread only)
    std::vector<std::shared_ptr<TripBlock>> tripPool;
    std::shared_ptr<TripBlock> takeTripBlock(size_t nEdges, size_t childDoubles);

private:
    ppgpu_ctx* m_Handle = nullptr;
    int m_Device = 0;
    std::thread m_Thread;
    std::mutex m_Mutex;
    std::condition_variable m_Wake;
    std::function<void()> m_Job;
    bool m_Busy = false, m_Quit = false;
    std::atomic<bool> m_Running{false};    // run() .. the end of its job (wait() polls it for a while before it sleeps)
    std::atomic<bool> m_Posted{false};     // a job is waiting (what the thread polls for a millisecond after each job before it sleeps)
    std::exception_ptr m_Error;
    void serve();
};

class GpuAStarPlanner : public Planner {
public:
    explicit GpuAStarPlanner(std::shared_ptr<GpuContext> ctx = GpuContext::shared()) : m_Ctx(ctx), m_Ctxs{ctx} {}
    // several devices of one node: world and samples are replicated on each (the sampler stream is deterministic: every device
    // draws the same samples itself), the open vertices of a batch are dealt across them, records come back to the host search
    explicit GpuAStarPlanner(std::vector<std::shared_ptr<GpuContext>> ctxs) : m_Ctx(ctxs.at(0)) {
        for (auto& c : ctxs)                 // (a context named twice is one context)
            if (std::find(m_Ctxs.begin(), m_Ctxs.end(), c) == m_Ctxs.end()) m_Ctxs.push_back(c);
    }
    ~GpuAStarPlanner() override;      // the search tree is torn down here, as the reference's is (not inside plan()'s budget)
    Stats plan(const RibbonManager& ribbonManager, const State& start, PlannerConfig config, const DubinsPlan& previousPlan,
               double timeRemaining) override;

    typedef SearchNode Node;

private:
    std::shared_ptr<GpuContext> m_Ctx;                    // device 0 of this planner: sampling read-back, wrapper edges, explicit targets
    std::vector<std::shared_ptr<GpuContext>> m_Ctxs;      // all of them (m_Ctxs[0] == m_Ctx)
    std::vector<Node> m_Nodes;
    // binary heap of node indices, min f (AStarPlanner.cpp:6-10).  Each entry carries its node's f: the heap's comparisons then read
    // 16-byte neighbours instead of two 400-byte nodes somewhere in a 100 MB tree (round 4: a push's cache misses were most of the
    // 0.3 us a child cost the planner's thread).  Same comparisons, same results, same heap.
    struct QEntry { double f; int v; };
    std::vector<QEntry> m_Queue;
    int m_Best = -1;
    double m_StartStateTime = 0;
    RibbonManager m_RibbonManager;
    long m_NumSamples = 0;
    // Edges of open vertices costed ahead of their expansion, as the device returned them: a child Node is built only when the
    // search expands the parent, and only for edges it would push (an infeasible edge never becomes a vertex).
    struct Costed {
        std::shared_ptr<TripBlock> block;     // the round trip that costed them
        size_t first = 0, count = 0;          // this vertex's edges: [first, first + count) of the block, in push order
        // children built ahead while the planner's thread waited for a round trip (prebuildWhileWaiting): kids[i] is edge first + i's
        // child vertex, complete, where ready[i] is set; the others are left to expand() (never pushed, or the planner's own judgement)
        std::vector<Node> kids;
        std::vector<unsigned char> ready;
    };
    std::unordered_map<int, Costed> m_Speculated;

    void uploadWorld(const State& start);
    void pushVertexQueue(int v);
    int popVertexQueue();
    bool goalCondition(const Node& v) const;
    bool expand(int source);               // false: the deadline guard did not start the round trip it needed
    struct Batch;                          // one device round trip, from the pick to the harvest (planner.cpp)
    std::vector<std::shared_ptr<Batch>> m_InFlight;        // started, not yet harvested: at most one per context
    std::unordered_map<int, Batch*> m_InFlightOf;          // open vertex -> the round trip that is costing its children
    void packBatch(Batch& b) const;
    static void runBatch(Batch& b, int k);
    void harvestFinished();
    void submitBatch(std::shared_ptr<Batch> b, GpuContext& ctx);
    void harvestBatch(Batch* b, bool keep);
    void dropBatch(Batch* b);
    void drainInFlight();
    GpuContext& freeContext();
    void pickBatch(int source, std::vector<int>& batch);
    double m_EndTime = 0;                  // the deadline of this plan() call, on the injected clock
    int costEdgeList(const std::vector<uint64_t>& edges, int maxParentRibbons, std::vector<::ppgpu_edge_result>& res, std::vector<double>& child);
    Node makeChild(int source, unsigned cfgBits, const ::ppgpu_edge_result& r, const double* childRibbons, int stride);
    void fillChild(Node& c, int source, unsigned cfgBits, const ::ppgpu_edge_result& r, const double* childRibbons) const;
    void prebuildWhileWaiting(GpuContext& busy);
    void addNode(Node&& n);                // m_Nodes.push_back that counts and times reallocations (Stats::Budget)
    void noteOperation(int kind, double startedAt, double predicted, double actual);
    double m_PlanEntry = 0;                // steady-clock time of plan()'s entry
    bool m_DeadlineStop = false;           // aStar() stopped on the deadline guard: the big loop ends too
    int m_StrideFloor = 0;                 // child-ribbon stride a round trip of this plan() had to be repeated with: later ones start there
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

// One planner iteration in its BATCH form, sample-sharded over the GPUs of a node (SURVEY.md 8 e; BASELINE's north_star: "the
// sample batch shards naturally across the 8 GPUs of one node with a single RCCL all-reduce of the per-shard best cost/vertex
// each iteration").  The iteration's batch of `attempts` draws from the StateGenerator stream is cut into contiguous slices, one
// per device context; every device skips the slices below its own (ppgpu_sampler_skip), draws and filters its slice, costs the
// edges from the open vertex to its samples under the four (radius, speed) configurations and reduces its best (f, edge); then
// ONE collective — ppgpu_allreduce_best on the communicator ppgpu_comm_init_all made of the contexts — leaves the global
// incumbent on every device.  No sample, edge or record crosses between devices.  Each context's work runs on its own host
// thread (GpuContext::run), which is also what the collective needs: every rank must have joined before any returns.
// The reference has no counterpart (its incumbent update is AStarPlanner.cpp:109-117 on one thread); GpuAStarPlanner's anytime
// search keeps dealing open VERTICES to the devices (the host pops goals in f order, no collective needed there).
class ShardedIteration {
public:
    struct Result {
        double f = 0;                          // the global incumbent: smallest f over every shard's feasible edges ...
        uint64_t fBits = ~0ull, edge = ~0ull;  // ... its bit pattern and global edge id (shard * edgesPerShard + local index); ~0: none
        int shard = -1;                        // which device found it
        std::vector<int64_t> kept;             // samples each shard kept (map filter)
        int64_t edges = 0;                     // edges costed over all shards
        int rcclRanks = 0;                     // ncclCommCount of the communicator (0: host combine, see below)
        bool agreed = false;                   // every device ended with the same key
    };
    // One context per device: the communicator is RCCL's.  Two contexts on ONE device (a one-GPU box rehearsing the split) cannot
    // form one — RCCL takes one rank per device — and the keys are then combined on the host instead; Result::rcclRanks says which.
    explicit ShardedIteration(std::vector<std::shared_ptr<GpuContext>> ctxs);
    ~ShardedIteration();
    // Throws what a shard threw.  A shard that fails before the collective keeps every shard out of it (nothing is left waiting);
    // a failure inside the collective aborts the communicators (ppgpu_comm_abort) so that the other ranks return.
    Result run(const RibbonManager& ribbonManager, const State& start, const PlannerConfig& config, unsigned long seed, int64_t attempts);
    int failShardForTest = -1;                 // tests only: this shard throws at the start of its work

private:
    void release();
    std::vector<std::shared_ptr<GpuContext>> m_Ctxs;
    std::vector<void*> m_Keys, m_Records;      // per device: the 16-byte key and the record buffer (device memory)
    std::vector<size_t> m_RecordCap;
    bool m_Rccl = false;
};

}  // namespace ppamd
