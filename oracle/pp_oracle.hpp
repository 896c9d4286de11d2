// pp_oracle.hpp — TEST INFRASTRUCTURE ONLY.
//
// Sequential CPU restatement of the reference's sampling / Dubins-edge / edge-cost
// hot path (afb2001/path_planner).  It exists to CHECK the HIP implementation:
// only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may build,
// load or call it.  Nothing under path_planner_amd/ includes or links this file.
//
// Every function cites the reference file:line it follows (pp = path_planner,
// ppc = path_planner_common, both under /root/reference).
//
// Pinning status (see DESIGN.md "Oracle"):
//   * Ribbon, State, GridWorldMap, BinaryDynamicObstaclesManager: checked against the
//     reference's own objects compiled in place (oracle/_ref, oracle/Makefile) and
//     against the reference tests' known answers (tests/test_known_answers.py).
//   * StateGenerator stream: pinned by the survey's probe of the reference object
//     (SURVEY.md section 8 a-1 golden states) — libstdc++ minstd_rand0 semantics.
//   * Dubins solver: the third-party `dubins_curves` package is absent and unpinned;
//     general CSC/CCC geometry is "parity unpinned" against the original binary and is
//     anchored on the reference tests' straight-line / half-turn answers plus
//     geometric property tests.  Everything built on top of it (Edge, Vertex,
//     RibbonManager, planners) needs dubins.h to compile, so those reference files are
//     unbuildable here and are restated from their text.
#pragma once
#include <cstdint>
#include <vector>
#include <functional>
#include <string>

namespace ppo {

// ---------------------------------------------------------------- State
// ppc/include/path_planner_common/State.h:13-213, ppc/src/state/State.cpp
struct State {
    double x = 0, y = 0, heading = 0, speed = 0;  // State.h:201
    double time = -1;                             // State.h:202
    State() = default;
    State(double x_, double y_, double h_, double s_, double t_) : x(x_), y(y_), heading(h_), speed(s_), time(t_) {}
    double yaw() const;                 // State.h:51-55
    void setYaw(double yaw1);           // State.h:62-65
    void move(double distance);         // State.cpp:22-25
    State push(double dt) const;        // State.cpp:11-20
    double headingTo(double x1, double y1) const;     // State.cpp:51-57
    void setHeadingTowards(double x1, double y1);     // State.cpp:64-67
    double distanceTo(double x1, double y1) const;    // State.cpp:91-93
    double distanceTo(const State& o) const { return distanceTo(o.x, o.y); }
    bool isCoLocated(const State& o) const { return x == o.x && y == o.y && heading == o.heading; }  // State.cpp:81-85
};

// ---------------------------------------------------------------- Dubins (third-party API restated)
struct DubinsPath { double qi[3]; double param[3]; double rho; int type; };
enum { EDUBOK = 0, EDUBCOCONFIGS = 1, EDUBPARAM = 2, EDUBBADRHO = 3, EDUBNOPATH = 4 };
int dubins_shortest_path(DubinsPath* path, const double q0[3], const double q1[3], double rho);
double dubins_path_length(const DubinsPath* path);
int dubins_path_sample(const DubinsPath* path, double t, double q[3]);
int dubins_extract_subpath(const DubinsPath* path, double t, DubinsPath* out);
int dubins_word(int type, const double q0[3], const double q1[3], double rho, double out[3]);

// ---------------------------------------------------------------- DubinsWrapper
// ppc/include/path_planner_common/DubinsWrapper.h, ppc/src/dubinsPlan/DubinsWrapper.cpp
struct SampleError { std::string what; };
struct DubinsWrapper {
    DubinsPath path{};
    double speed = 0;
    double startTime = -1, endTime = -1, updatedStartTime = -1;  // DubinsWrapper.h:120
    void set(const State& s1, const State& s2, double rho);      // DubinsWrapper.cpp:9-17
    void fill(const DubinsPath& p, double speed, double startTime); // :85-90
    double length() const;                                       // :19-22 (throws if unset)
    bool isInitialized() const { return startTime >= 0; }        // :51-53
    bool containsTime(double t) const;                           // :24-27
    void sample(State& s) const;                                 // :29-49 (throws SampleError)
    void setSpeed(double s);                                     // :121-124
    void setEndTime();                                           // :92-94
    void updateEndTime(double t);                                // :100-104
    double getRho() const { return path.rho; }
};

// ---------------------------------------------------------------- Ribbon / RibbonManager
// pp/src/planner/utilities/Ribbon.{h,cpp}, RibbonManager.{h,cpp}
struct Ribbon {
    double sx, sy, ex, ey;   // Ribbon.h:126
    double squaredLength() const;
    double length() const;
    bool covered(bool strict, double w) const;                   // Ribbon.cpp:23-25
    void projection(double x, double y, double& px, double& py) const; // Ribbon.cpp:72-78
    bool containsProjection(double px, double py) const;         // Ribbon.cpp:90-95
    double distance(double x, double y) const;                   // Ribbon.h:118-121
    bool contains(double x, double y, double px, double py, bool strict, double w) const; // Ribbon.cpp:39-43
    Ribbon split(double x, double y, bool strict, double w);     // Ribbon.cpp:9-17
    State startAsState() const;                                  // Ribbon.cpp:60-64
    State endAsState() const;                                    // Ribbon.cpp:66-70
};

enum Heuristic { MaxDistance = 0, TspPointRobotNoSplitAllRibbons, TspPointRobotNoSplitKRibbons,
                 TspDubinsNoSplitAllRibbons, TspDubinsNoSplitKRibbons };

struct RibbonManager {
    int heuristic = MaxDistance;
    double turningRadius = -1;
    int K = 0;
    double coverageCompletedTime = -1;
    std::vector<Ribbon> ribbons;      // std::list in the reference (RibbonManager.h:200); order preserved
    static double RibbonWidth;        // Ribbon::RibbonWidth (Ribbon.cpp:4)
    static double minLength() { return 2 * RibbonWidth; }         // Ribbon.cpp:52-58

    void add(double x1, double y1, double x2, double y2);         // RibbonManager.cpp:7-12
    void cover(double x, double y, bool strict);                  // :14-22
    void coverBetween(double x1, double y1, double x2, double y2, bool strict); // :391-403
    bool done() const { return ribbons.empty(); }                 // :24-26
    double approximateDistanceUntilDone(double x, double y, double yaw) const; // :28-51
    double minDistanceFrom(double x, double y) const;             // :142-152
    double maxDistance(double x, double y) const;                 // :234-248
    State getNearestEndpointAsState(const State& s) const;        // :160-195
    void projectOntoNearestRibbon(State& s) const;                // :220-232
    void changeHeuristicIfTooManyRibbons();                       // :381-385
    void setCoverageCompletedTime(double t) { if (coverageCompletedTime == -1) coverageCompletedTime = t; } // :409-412
    std::vector<State> findNearStatesOnRibbons(const State& start, double radius) const; // :296-379
    double dubinsDistance(double x, double y, double h, const State& s) const; // RibbonManager.h:210-216
};

// ---------------------------------------------------------------- Map / obstacles
struct GridMap {
    // rows == 0: base Map (pp/src/common/map/Map.cpp:4-6); else GridWorldMap (GridWorldMap.cpp)
    int rows = 0, cols = 0;
    double resolution = 0;
    std::vector<uint8_t> cells;  // row 0 = y in [0,res)
    double extremes[4] = {-1.7976931348623157e308, 1.7976931348623157e308, -1.7976931348623157e308, 1.7976931348623157e308};
    bool isBlocked(double x, double y) const;           // GridWorldMap.cpp:84-93
    bool loadText(const std::string& text);             // GridWorldMap.cpp:10-82 (from a string instead of a path)
    void setCells(const uint8_t* c, int rows, int cols, double res);
};

struct BinaryObstacle { double X, Y, Yaw, Speed, Time, Width, Length; };  // BinaryDynamicObstaclesManager.h:15-16
// GaussianDynamicObstaclesManager::Obstacle (GaussianDynamicObstaclesManager.h:19-49).  The reference uses Eigen (absent here):
// the 2x2 inverse / determinant / quadratic form are written out in Eigen's fixed-size evaluation order.  PARITY
// UNPINNED for this model: the reference's only test of it prints values without asserting (test_planner.cpp:230-238).
struct GaussianObstacle { double X, Y, Yaw, Speed, Time; double cov[2][2]; };
struct Obstacles {
    int model = 0;   // 0 = base DynamicObstaclesManager (returns 0), 1 = binary, 2 = Gaussian
    std::vector<BinaryObstacle> list;
    std::vector<GaussianObstacle> gauss;
    void update(double x, double y, double heading, double speed, double time, double width, double length); // .h:17-19
    void updateGaussian(double x, double y, double heading, double speed, double time, const double* cov4);  // Gaussian .cpp:16-26,37-47
    double collisionExists(double x, double y, double time, bool strict) const; // Binary .cpp:4-22 / Gaussian .cpp:3-13
};

// ---------------------------------------------------------------- Config
// pp/src/planner/PlannerConfig.h:179-207, Edge.h:151-152
struct Config {
    int branchingFactor = 9;
    double maxSpeed = 2.5, slowSpeedRaw = 0.5, turningRadius = 8, coverageTurningRadius = 16;
    double timeHorizon = 30, timeMinimum = 5;
    double collisionCheckingIncrement = 0.05;
    int initialSamples = 100;
    bool useBrownPaths = false;
    double startStateTime = 0;
    double collisionPenaltyFactor = 600, timePenaltyFactor = 1;
    const GridMap* map = nullptr;
    const Obstacles* obstacles = nullptr;
    std::function<double()> now;
    // Checker-only knob mirroring the DEVICE contract (include/ppgpu.h, PPGPU_F_RIBBON_OVF): with a TSP heuristic
    // and more ribbons than this, h is reported as 0 instead of enumerating 2^n n! tours.  0 = no limit (reference).
    int tspRibbonLimit = 0;
    // Checker-only: leave the heuristic's VALUE out (h = 0) where the comparison discards it anyway; flags are decided as usual.
    // The reference's Dubins-TSP recursion (RibbonManager.cpp:97-140) solves a Dubins problem per tree node: 2^n n! leaves, 6 s per
    // edge at 8 ribbons, minutes at 9 — what made a randomized round look hung.
    bool skipHeuristicValue = false;
    double slowSpeed() const { return slowSpeedRaw <= 0 ? maxSpeed : slowSpeedRaw; }  // PlannerConfig.h:168-171
};

// ---------------------------------------------------------------- Vertex / Edge
// pp/src/planner/search/{Vertex,Edge}.{h,cpp}; arena-allocated, parent by index.
struct Vertex {
    State state;
    int parent = -1;                 // index in the arena, -1 = root
    RibbonManager ribbons;
    double currentCost = -1;         // g
    double approxToGo = -1;          // h
    double turningRadius = 0;
    bool coverageAllowed = false;
    // parent edge
    DubinsWrapper wrapper;
    bool infeasible = false;
    bool threw = false;              // computeTrueCost would have thrown
    double edgeApproxCost = -1, edgeTrueCost = -1, collisionPenalty = 0;
    int steps = 0;                   // sweep iterations executed (diagnostic)
    bool heuristicSkipped = false;   // tspRibbonLimit applied
    int events = 0, mutations = 0;
    int mutKinds[4] = {0, 0, 0, 0};   // [0] one piece's start moved, [1] two pieces changed, same count, [2] count changed, [3] other   // coverage events / events that changed the ribbon list (diagnostic)
    double f() const { return currentCost + approxToGo; }
    int depth(const std::vector<Vertex>& arena) const;
};

// Edge::computeApproxCost(maxSpeed, turningRadius) (Edge.cpp:11-20)
double computeApproxCost(const Vertex& start, Vertex& end, double maxSpeed, double turningRadius);
// Edge::computeTrueCost (Edge.cpp:68-206) incl. Vertex::setCurrentCost / computeApproxToGo
double computeTrueCost(const Vertex& start, Vertex& end, const Config& cfg);
// Vertex::connect overloads (Vertex.cpp:21-43,125-130)
Vertex connectState(const Vertex& start, int startIndex, const State& next, double turningRadius, bool coverageAllowed);
Vertex connectWrapper(const Vertex& start, int startIndex, const DubinsWrapper& w, bool coverageAllowed);
Vertex makeRoot(const State& s, const RibbonManager& r);    // Vertex.cpp:38-43
double computeApproxToGo(Vertex& v, const Config& cfg);      // Vertex.cpp:49-64

// ---------------------------------------------------------------- StateGenerator
// pp/src/planner/utilities/StateGenerator.{h,cpp}; libstdc++ default_random_engine = minstd_rand0
struct StateGenerator {
    double minX, maxX, minY, maxY, minSpeed, maxSpeed;
    uint32_t engine = 1;
    bool sampleOnRibbons = false;
    RibbonManager ribbons;
    uint64_t draws = 0;  // engine invocations so far (diagnostic / jump-ahead tests)
    StateGenerator(double minX, double maxX, double minY, double maxY, double minSpeed, double maxSpeed, unsigned long seed);
    StateGenerator(double minX, double maxX, double minY, double maxY, double minSpeed, double maxSpeed, unsigned long seed,
                   const RibbonManager& r);
    uint32_t next();                       // minstd_rand0 step
    double canonical();                    // std::generate_canonical<double,53>
    double uniform(double a, double b);    // std::uniform_real_distribution<double>
    State generate();                      // StateGenerator.cpp:15-31
};

// ---------------------------------------------------------------- Planner
// pp/src/planner/{SamplingBasedPlanner,AStarPlanner,Planner}.{h,cpp}
struct Stats {
    unsigned long Samples = 0, Generated = 0, Expanded = 0, Iterations = 0;
    double PlanFValue = 0, PlanCollisionPenalty = 0, PlanTimePenalty = 0, PlanHValue = 0;
    unsigned long PlanDepth = 0;
    std::vector<DubinsWrapper> Plan;
    // extras for parity tests
    long firstGoalIteration = -1;            // index of the first iteration whose aStar() returned a goal
    std::vector<double> iterationBestF;      // incumbent f after each iteration (NaN if none)
};

struct AStarPlanner {
    Config cfg;
    double startStateTime = 0;
    std::vector<State> samples;
    unsigned long attemptedSamples = 0;
    std::vector<Vertex> arena;
    std::vector<int> queue;     // heap of arena indices (std::push_heap/pop_heap with the A* comparator)
    int best = -1;
    RibbonManager ribbonManager;
    Stats stats;
    // hook: called for every true-costed edge (start index, end vertex) — used to dump goldens
    std::function<void(const Vertex&, const Vertex&)> onEdge;

    void pushVertexQueue(int v);                 // SamplingBasedPlanner.cpp:7-19
    int popVertexQueue();                        // :21-27 (returns -1 when empty instead of throwing)
    bool goalCondition(const Vertex& v) const;   // :42-50
    void expand(int source);                     // :52-151
    void addSamples(StateGenerator& g, int n);   // :157-164
    void addSamples(StateGenerator& g);          // :166-168
    int aStar(double endTime);                   // AStarPlanner.cpp:134-148
    void expandToCoverSpecificSamples(int root, const std::vector<State>& s, bool coverageAllowed); // :150-162
    Stats plan(const RibbonManager& ribbons, const State& start, Config config,
               const std::vector<DubinsWrapper>& previousPlan, double timeRemaining); // AStarPlanner.cpp:12-132
    std::vector<DubinsWrapper> tracePlan(int v);  // Planner.cpp:12-32
};

}  // namespace ppo
