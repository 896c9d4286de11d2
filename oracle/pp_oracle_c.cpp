// pp_oracle_c.cpp — TEST INFRASTRUCTURE ONLY: flat C entry points over pp_oracle.cpp so that
// tests/ (ctypes) and bench.py's cpu_baseline leg can drive the CPU restatement with the same
// records (include/ppgpu.h) the HIP library produces.  Not linked into any product library.
#include <cmath>
#include <chrono>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../include/ppgpu.h"
#include "pp_oracle.hpp"

using namespace ppo;

namespace {
struct World {
    GridMap map;
    Obstacles obstacles;
    ppgpu_config pc{};
    bool have_cfg = false;
    int tsp_limit = 8;   // PP_TSP_MAX of the device path; ppo_world_set_tsp_limit(0) restores the unbounded reference
    bool skip_heuristic_value = false;   // Config::skipHeuristicValue
};

// No C++ exception may cross the ctypes boundary (it would end the test process with std::terminate): anything other than the
// SampleError the records have a flag for is kept here and turns the call's return code into -2.
std::mutex g_error_mutex;
std::string g_error;
void note_error(const char* where, const std::exception& e) {
    std::lock_guard<std::mutex> lock(g_error_mutex);
    if (g_error.empty()) g_error = std::string(where) + ": " + e.what();
}
bool take_error() {
    std::lock_guard<std::mutex> lock(g_error_mutex);
    return !g_error.empty();
}

Config make_config(const World& w) {
    Config c;
    const ppgpu_config& p = w.pc;
    c.branchingFactor = p.branching_factor;
    c.maxSpeed = p.max_speed;
    c.slowSpeedRaw = p.slow_speed;
    c.turningRadius = p.turning_radius;
    c.coverageTurningRadius = p.coverage_turning_radius;
    c.timeHorizon = p.time_horizon;
    c.timeMinimum = p.time_minimum;
    c.collisionCheckingIncrement = p.collision_checking_increment;
    c.startStateTime = p.start_state_time;
    c.collisionPenaltyFactor = p.collision_penalty_factor;
    c.timePenaltyFactor = p.time_penalty_factor;
    c.map = &w.map;
    c.obstacles = &w.obstacles;
    c.tspRibbonLimit = w.tsp_limit;
    c.skipHeuristicValue = w.skip_heuristic_value;
    RibbonManager::RibbonWidth = p.ribbon_width;
    return c;
}

RibbonManager make_rm(const World& w, const double* ribbons4, int n, double cct) {
    RibbonManager rm;
    rm.heuristic = w.pc.heuristic;
    rm.K = w.pc.tsp_k;
    rm.turningRadius = w.pc.heuristic_turning_radius;
    rm.coverageCompletedTime = cct;
    for (int i = 0; i < n; i++) rm.ribbons.push_back(Ribbon{ribbons4[4 * i], ribbons4[4 * i + 1], ribbons4[4 * i + 2], ribbons4[4 * i + 3]});
    return rm;
}

void path_to8(const DubinsPath& p, double* o) {
    o[0] = p.qi[0]; o[1] = p.qi[1]; o[2] = p.qi[2];
    o[3] = p.param[0]; o[4] = p.param[1]; o[5] = p.param[2];
    o[6] = p.rho; o[7] = (double)p.type;
}
DubinsPath path_from8(const double* o) {
    DubinsPath p;
    p.qi[0] = o[0]; p.qi[1] = o[1]; p.qi[2] = o[2];
    p.param[0] = o[3]; p.param[1] = o[4]; p.param[2] = o[5];
    p.rho = o[6]; p.type = (int)o[7];
    return p;
}
}  // namespace

extern "C" {

// ------------------------------------------------------------------ dubins
int ppo_dubins_shortest_path(const double* q0, const double* q1, double rho, double* out8) {
    DubinsPath p{};
    int e = dubins_shortest_path(&p, q0, q1, rho);
    path_to8(p, out8);
    return e;
}
int ppo_dubins_word(int type, const double* q0, const double* q1, double rho, double* out3) {
    return dubins_word(type, q0, q1, rho, out3);
}
double ppo_dubins_path_length(const double* path8) {
    DubinsPath p = path_from8(path8);
    return dubins_path_length(&p);
}
int ppo_dubins_path_sample(const double* path8, double t, double* q3) {
    DubinsPath p = path_from8(path8);
    return dubins_path_sample(&p, t, q3);
}
int ppo_dubins_extract_subpath(const double* path8, double t, double* out8) {
    DubinsPath p = path_from8(path8), o{};
    int e = dubins_extract_subpath(&p, t, &o);
    if (e == EDUBOK) path_to8(o, out8);
    return e;
}

// DubinsWrapper::fill(path, speed, startTime) then getEndTime() (DubinsWrapper.cpp:85-94)
double ppo_wrapper_fill_end_time(const double* path8, double speed, double start_time) {
    DubinsWrapper w;
    w.fill(path_from8(path8), speed, start_time);
    return w.endTime;
}

// DubinsWrapper(s1, s2, rho) then sample(time): returns 0 ok, 1 = would throw.  out5 = sampled State.
int ppo_wrapper_sample(const double* s1_5, const double* s2_5, double rho, double new_speed, double time, double* out5,
                       double* end_time) {
    State a(s1_5[0], s1_5[1], s1_5[2], s1_5[3], s1_5[4]), b(s2_5[0], s2_5[1], s2_5[2], s2_5[3], s2_5[4]);
    DubinsWrapper w;
    w.set(a, b, rho);
    if (new_speed > 0) w.setSpeed(new_speed);
    if (end_time) *end_time = w.endTime;
    State s;
    s.time = time;
    try {
        w.sample(s);
    } catch (SampleError&) {
        return 1;
    }
    out5[0] = s.x; out5[1] = s.y; out5[2] = s.heading; out5[3] = s.speed; out5[4] = s.time;
    return 0;
}

// ------------------------------------------------------------------ State helpers
double ppo_state_yaw(double heading) { State s(0, 0, heading, 0, 0); return s.yaw(); }
double ppo_state_heading_to(double x, double y, double x1, double y1) { State s(x, y, 0, 0, 0); return s.headingTo(x1, y1); }
void ppo_state_move(double* s5, double d) {
    State s(s5[0], s5[1], s5[2], s5[3], s5[4]);
    s.move(d);
    s5[0] = s.x; s5[1] = s.y;
}
void ppo_state_push(const double* s5, double dt, double* out5) {
    State s(s5[0], s5[1], s5[2], s5[3], s5[4]);
    State o = s.push(dt);
    out5[0] = o.x; out5[1] = o.y; out5[2] = o.heading; out5[3] = o.speed; out5[4] = o.time;
}

// ------------------------------------------------------------------ ribbons (world-less)
void ppo_set_ribbon_width(double w) { RibbonManager::RibbonWidth = w; }
double ppo_get_ribbon_width() { return RibbonManager::RibbonWidth; }

// RibbonManager::add with its covered() filter; returns new count
int ppo_ribbons_add(double* ribbons4, int n, int cap, double x1, double y1, double x2, double y2) {
    World w;
    RibbonManager rm = make_rm(w, ribbons4, n, -1);
    rm.add(x1, y1, x2, y2);
    int m = (int)rm.ribbons.size();
    if (m > cap) return -1;
    for (int i = 0; i < m; i++) { ribbons4[4 * i] = rm.ribbons[i].sx; ribbons4[4 * i + 1] = rm.ribbons[i].sy; ribbons4[4 * i + 2] = rm.ribbons[i].ex; ribbons4[4 * i + 3] = rm.ribbons[i].ey; }
    return m;
}
static int store(const RibbonManager& rm, double* ribbons4, int cap) {
    int m = (int)rm.ribbons.size();
    if (m > cap) return -1;
    for (int i = 0; i < m; i++) { ribbons4[4 * i] = rm.ribbons[i].sx; ribbons4[4 * i + 1] = rm.ribbons[i].sy; ribbons4[4 * i + 2] = rm.ribbons[i].ex; ribbons4[4 * i + 3] = rm.ribbons[i].ey; }
    return m;
}
int ppo_ribbons_cover(double* ribbons4, int n, int cap, double x, double y, int strict) {
    World w;
    RibbonManager rm = make_rm(w, ribbons4, n, -1);
    rm.cover(x, y, strict != 0);
    return store(rm, ribbons4, cap);
}
int ppo_ribbons_cover_between(double* ribbons4, int n, int cap, double x1, double y1, double x2, double y2, int strict) {
    World w;
    RibbonManager rm = make_rm(w, ribbons4, n, -1);
    rm.coverBetween(x1, y1, x2, y2, strict != 0);
    return store(rm, ribbons4, cap);
}
double ppo_ribbons_min_distance(const double* ribbons4, int n, double x, double y) {
    World w;
    return make_rm(w, ribbons4, n, -1).minDistanceFrom(x, y);
}
double ppo_ribbons_heuristic(const double* ribbons4, int n, int heuristic, int K, double turningRadius, double x, double y,
                             double yaw) {
    World w;
    RibbonManager rm = make_rm(w, ribbons4, n, -1);
    rm.heuristic = heuristic; rm.K = K; rm.turningRadius = turningRadius;
    return rm.approximateDistanceUntilDone(x, y, yaw);
}
int ppo_ribbons_nearest_endpoint(const double* ribbons4, int n, const double* s5, double* out5) {
    World w;
    RibbonManager rm = make_rm(w, ribbons4, n, -1);
    if (rm.done()) return 1;
    State s = rm.getNearestEndpointAsState(State(s5[0], s5[1], s5[2], s5[3], s5[4]));
    out5[0] = s.x; out5[1] = s.y; out5[2] = s.heading; out5[3] = s.speed; out5[4] = s.time;
    return 0;
}
// RibbonManager::findNearStatesOnRibbons (RibbonManager.cpp:296-379): the Brown-path seeds of AStarPlanner.cpp:40-43
int ppo_ribbons_near_states(const double* ribbons4, int n, const double* start5, double radius, double* out5, int cap) {
    World w;
    RibbonManager rm = make_rm(w, ribbons4, n, -1);
    std::vector<State> v = rm.findNearStatesOnRibbons(State(start5[0], start5[1], start5[2], start5[3], start5[4]), radius);
    for (size_t i = 0; i < v.size() && (int)i < cap; i++) {
        out5[5 * i] = v[i].x; out5[5 * i + 1] = v[i].y; out5[5 * i + 2] = v[i].heading; out5[5 * i + 3] = v[i].speed; out5[5 * i + 4] = v[i].time;
    }
    return (int)v.size();
}
void ppo_ribbons_project(const double* ribbons4, int n, double* s5) {
    World w;
    RibbonManager rm = make_rm(w, ribbons4, n, -1);
    State s(s5[0], s5[1], s5[2], s5[3], s5[4]);
    rm.projectOntoNearestRibbon(s);
    s5[0] = s.x; s5[1] = s.y; s5[2] = s.heading; s5[3] = s.speed; s5[4] = s.time;
}
// single-ribbon primitives, for comparison with the reference's Ribbon class (oracle/_ref)
void ppo_ribbon_projection(const double* r4, double x, double y, double* out2) {
    Ribbon r{r4[0], r4[1], r4[2], r4[3]};
    r.projection(x, y, out2[0], out2[1]);
}
int ppo_ribbon_contains(const double* r4, double x, double y, int strict) {
    Ribbon r{r4[0], r4[1], r4[2], r4[3]};
    double px, py;
    r.projection(x, y, px, py);
    return r.contains(x, y, px, py, strict != 0, RibbonManager::RibbonWidth) ? 1 : 0;
}
int ppo_ribbon_contains_projection(const double* r4, double px, double py) {
    Ribbon r{r4[0], r4[1], r4[2], r4[3]};
    return r.containsProjection(px, py) ? 1 : 0;
}
double ppo_ribbon_distance(const double* r4, double x, double y) {
    Ribbon r{r4[0], r4[1], r4[2], r4[3]};
    return r.distance(x, y);
}
int ppo_ribbon_covered(const double* r4, int strict) {
    Ribbon r{r4[0], r4[1], r4[2], r4[3]};
    return r.covered(strict != 0, RibbonManager::RibbonWidth) ? 1 : 0;
}
// split: r4 updated in place, front part to out4
void ppo_ribbon_split(double* r4, double x, double y, int strict, double* out4) {
    Ribbon r{r4[0], r4[1], r4[2], r4[3]};
    Ribbon f = r.split(x, y, strict != 0, RibbonManager::RibbonWidth);
    r4[0] = r.sx; r4[1] = r.sy; r4[2] = r.ex; r4[3] = r.ey;
    out4[0] = f.sx; out4[1] = f.sy; out4[2] = f.ex; out4[3] = f.ey;
}
void ppo_ribbon_end_states(const double* r4, double* start5, double* end5) {
    Ribbon r{r4[0], r4[1], r4[2], r4[3]};
    State a = r.startAsState(), b = r.endAsState();
    start5[0] = a.x; start5[1] = a.y; start5[2] = a.heading; start5[3] = a.speed; start5[4] = a.time;
    end5[0] = b.x; end5[1] = b.y; end5[2] = b.heading; end5[3] = b.speed; end5[4] = b.time;
}

// ------------------------------------------------------------------ world
void* ppo_world_create() { return new World(); }
void ppo_world_destroy(void* w) { delete (World*)w; }
void ppo_world_set_config(void* w, const ppgpu_config* c) { ((World*)w)->pc = *c; ((World*)w)->have_cfg = true; RibbonManager::RibbonWidth = c->ribbon_width; }
void ppo_world_set_tsp_limit(void* w, int limit) { ((World*)w)->tsp_limit = limit; }
void ppo_world_set_skip_heuristic_value(void* w, int on) { ((World*)w)->skip_heuristic_value = on != 0; }
// message of the first exception a call swallowed (empty string: none); reading clears it
const char* ppo_last_error() {
    static thread_local std::string out;
    std::lock_guard<std::mutex> lock(g_error_mutex);
    out = g_error;
    g_error.clear();
    return out.c_str();
}
void ppo_world_set_grid(void* w, const uint8_t* cells, int rows, int cols, double res) {
    World* W = (World*)w;
    if (rows == 0) { W->map = GridMap(); return; }
    W->map.setCells(cells, rows, cols, res);
}
// GridWorldMap text format; returns rows (<0 on failure); cols/res via out params
int ppo_world_load_grid_text(void* w, const char* text, int* cols, double* res) {
    World* W = (World*)w;
    if (!W->map.loadText(text)) return -1;
    if (cols) *cols = W->map.cols;
    if (res) *res = W->map.resolution;
    return W->map.rows;
}
void ppo_world_get_cells(void* w, uint8_t* out) {
    World* W = (World*)w;
    memcpy(out, W->map.cells.data(), W->map.cells.size());
}
void ppo_world_extremes(void* w, double* out4) { memcpy(out4, ((World*)w)->map.extremes, 4 * sizeof(double)); }
int ppo_world_is_blocked(void* w, double x, double y) { return ((World*)w)->map.isBlocked(x, y) ? 1 : 0; }
void ppo_world_is_blocked_many(void* w, long n, const double* x, const double* y, uint8_t* out) {
    World* W = (World*)w;
    for (long i = 0; i < n; i++) out[i] = W->map.isBlocked(x[i], y[i]) ? 1 : 0;
}
void ppo_world_set_obstacles(void* w, int model, int n, const double* o7) {
    World* W = (World*)w;
    W->obstacles.model = model;
    W->obstacles.list.clear();
    for (int i = 0; i < n; i++) W->obstacles.update(o7[7 * i], o7[7 * i + 1], o7[7 * i + 2], o7[7 * i + 3], o7[7 * i + 4], o7[7 * i + 5], o7[7 * i + 6]);
}
// Gaussian model: rows {X, Y, heading, Speed, Time, c00, c01, c10, c11}; cov_given = 0 uses the default covariance and rows of 5
void ppo_world_set_gaussian_obstacles(void* w, int n, const double* o, int cov_given) {
    World* W = (World*)w;
    W->obstacles.model = 2;
    W->obstacles.list.clear();
    W->obstacles.gauss.clear();
    const int stride = cov_given ? 9 : 5;
    for (int i = 0; i < n; i++) {
        const double* r = o + (size_t)stride * i;
        W->obstacles.updateGaussian(r[0], r[1], r[2], r[3], r[4], cov_given ? r + 5 : nullptr);
    }
}
double ppo_world_collision_exists(void* w, double x, double y, double t, int strict) {
    return ((World*)w)->obstacles.collisionExists(x, y, t, strict != 0);
}

// ------------------------------------------------------------------ sampler
// Raw generator stream: n states from a fresh StateGenerator, after skipping `skip` states.
// n_ribbons < 0: ribbon-less constructor.  draws_out = engine invocations consumed in total.
void ppo_sampler_generate(const double* b6, uint64_t seed, int n_ribbons, const double* ribbons4, long skip, long n,
                          double* out5, uint64_t* draws_out) {
    World w;
    StateGenerator g = n_ribbons < 0 ? StateGenerator(b6[0], b6[1], b6[2], b6[3], b6[4], b6[5], seed)
                                     : StateGenerator(b6[0], b6[1], b6[2], b6[3], b6[4], b6[5], seed, make_rm(w, ribbons4, n_ribbons, -1));
    for (long i = 0; i < skip; i++) g.generate();
    for (long i = 0; i < n; i++) {
        State s = g.generate();
        out5[5 * i] = s.x; out5[5 * i + 1] = s.y; out5[5 * i + 2] = s.heading; out5[5 * i + 3] = s.speed; out5[5 * i + 4] = s.time;
    }
    if (draws_out) *draws_out = g.draws;
}

// SamplingBasedPlanner::addSamples: attempts [skip, skip+n) of the stream, keeping unblocked ones.
long ppo_add_samples(void* w, const double* b6, uint64_t seed, int n_ribbons, const double* ribbons4, long skip, long n,
                     double* out5) {
    World* W = (World*)w;
    World tmp;
    StateGenerator g = n_ribbons < 0 ? StateGenerator(b6[0], b6[1], b6[2], b6[3], b6[4], b6[5], seed)
                                     : StateGenerator(b6[0], b6[1], b6[2], b6[3], b6[4], b6[5], seed, make_rm(tmp, ribbons4, n_ribbons, -1));
    for (long i = 0; i < skip; i++) g.generate();
    long kept = 0;
    for (long i = 0; i < n; i++) {
        State s = g.generate();
        if (!W->map.isBlocked(s.x, s.y)) {
            out5[5 * kept] = s.x; out5[5 * kept + 1] = s.y; out5[5 * kept + 2] = s.heading; out5[5 * kept + 3] = s.speed; out5[5 * kept + 4] = s.time;
            kept++;
        }
    }
    return kept;
}

// ------------------------------------------------------------------ edges
static thread_local int g_last_events = 0, g_last_mutations = 0, g_last_kinds[4] = {0, 0, 0, 0};
// computeTrueCost + the record, for an edge whose child vertex has been connected (state edge or wrapper edge)
static void finish_edge(const World& W, const Config& cfg, Vertex& src, Vertex& end, ppgpu_edge_result* out, double* child, int stride) {
    memset(out, 0, sizeof(*out));
    bool threw = false;
    try {
        computeTrueCost(src, end, cfg);
    } catch (SampleError&) {
        threw = true;
    }
    g_last_events = end.events; g_last_mutations = end.mutations;
    for (int q = 0; q < 4; q++) g_last_kinds[q] = end.mutKinds[q];
    uint32_t flags = 0;
    if (end.infeasible) flags |= PPGPU_F_INFEASIBLE;
    if (threw) flags |= PPGPU_F_THROWS | PPGPU_F_INFEASIBLE;
    int nr = (int)end.ribbons.ribbons.size();
    if (!threw) {
        if (end.ribbons.done()) flags |= PPGPU_F_DONE;
        double coverageDoneTime = end.ribbons.coverageCompletedTime + cfg.timeMinimum;  // goalCondition
        double nonCoverageDoneTime = cfg.startStateTime + cfg.timeHorizon;
        if (end.state.time >= nonCoverageDoneTime || (end.ribbons.done() && end.state.time >= coverageDoneTime)) flags |= PPGPU_F_GOAL;
        out->true_cost = end.edgeTrueCost;
        out->collision_penalty = end.collisionPenalty;
        out->approx_cost = end.edgeApproxCost;
        out->end_x = end.state.x; out->end_y = end.state.y; out->end_heading = end.state.heading;
        out->end_speed = end.state.speed; out->end_time = end.state.time;
        out->g = end.currentCost; out->h = end.approxToGo; out->f = end.currentCost + end.approxToGo;
        out->coverage_completed_time = end.ribbons.coverageCompletedTime;
        out->param[0] = end.wrapper.path.param[0]; out->param[1] = end.wrapper.path.param[1]; out->param[2] = end.wrapper.path.param[2];
        if (end.heuristicSkipped) flags |= PPGPU_F_RIBBON_OVF;
        if (child) {
            if (nr > stride) {   // the device contract: a truncated ribbon list carries no heuristic (h = 0, f = g)
                flags |= PPGPU_F_RIBBON_OVF;
                out->h = 0; out->f = out->g;
            }
            for (int i = 0; i < nr && i < stride; i++) {
                child[4 * i] = end.ribbons.ribbons[i].sx; child[4 * i + 1] = end.ribbons.ribbons[i].sy;
                child[4 * i + 2] = end.ribbons.ribbons[i].ex; child[4 * i + 3] = end.ribbons.ribbons[i].ey;
            }
        }
    }
    out->flags = flags;
    out->info = (uint32_t)(end.wrapper.path.type & 0xff) | ((uint32_t)(nr & 0xff) << 8) | ((uint32_t)(end.steps & 0xffff) << 16);
}

// Vertex::connect(state) + Edge::computeTrueCost for a list of packed edge descriptors.
// n_threads > 1: static chunking over std::thread (edges are independent).

static void cost_one(const World& W, const Config& cfg, const ppgpu_vertex* verts, const double* pool,
                     const double* sx, const double* sy, const double* sh, uint64_t desc, ppgpu_edge_result* out,
                     double* child, int stride) {
    uint32_t target = (uint32_t)(desc & 0xffffffffu);
    uint32_t vi = (uint32_t)((desc >> 32) & 0xffffffu);
    uint32_t c = (uint32_t)(desc >> 56);
    const ppgpu_vertex& pv = verts[vi];
    Vertex src;
    src.state = State(pv.x, pv.y, pv.heading, pv.speed, pv.time);
    src.currentCost = pv.g;
    src.ribbons = make_rm(W, pool + 4 * (size_t)pv.ribbon_offset, pv.ribbon_count, pv.coverage_completed_time);
    bool cov = (c & PPGPU_EDGE_COVERAGE) != 0;
    double rho = cov ? cfg.coverageTurningRadius : cfg.turningRadius;
    double speed = (c & PPGPU_EDGE_SLOW) ? cfg.slowSpeed() : cfg.maxSpeed;
    State tgt(sx[target], sy[target], sh[target], speed, 0);
    try {
        Vertex end = connectState(src, 0, tgt, rho, cov);
        finish_edge(W, cfg, src, end, out, child, stride);
    } catch (const std::exception& e) {   // not a SampleError (finish_edge turns those into PPGPU_F_THROWS): a checker failure
        memset(out, 0, sizeof(*out));
        out->flags = PPGPU_F_THROWS | PPGPU_F_INFEASIBLE;
        note_error("ppo_cost_edges", e);
    }
}

int ppo_cost_edges(void* w, const ppgpu_vertex* verts, const double* pool, const double* sx, const double* sy,
                   const double* sh, long n, const uint64_t* edges, ppgpu_edge_result* out, double* child_ribbons,
                   int stride, int n_threads) {
    World* W = (World*)w;
    if (!W->have_cfg) return -1;
    Config cfg = make_config(*W);
    auto work = [&](long a, long b) {
        for (long e = a; e < b; e++)
            cost_one(*W, cfg, verts, pool, sx, sy, sh, edges[e], out + e, child_ribbons ? child_ribbons + (size_t)e * stride * 4 : nullptr, stride);
    };
    if (n_threads <= 1) {
        work(0, n);
    } else {
        std::vector<std::thread> th;
        for (int t = 0; t < n_threads; t++) th.emplace_back(work, n * t / n_threads, n * (t + 1) / n_threads);
        for (auto& t : th) t.join();
    }
    return take_error() ? -2 : 0;
}

// Vertex::connect(start, DubinsWrapper, coverageAllowed) + computeTrueCost (Vertex.cpp:28-36, Edge.cpp:208-216): what
// AStarPlanner::plan does with each segment of the previous plan (AStarPlanner.cpp:46-59).  Same records as ppo_cost_edges.
int ppo_cost_wrapper_edges(void* w, const ppgpu_vertex* verts, const double* pool, long n, const ppgpu_wrapper_edge* edges,
                           ppgpu_edge_result* out, double* child_ribbons, int stride) {
    World* W = (World*)w;
    if (!W->have_cfg) return -1;
    Config cfg = make_config(*W);
    for (long e = 0; e < n; e++) {
        const ppgpu_wrapper_edge& we = edges[e];
        const ppgpu_vertex& pv = verts[we.vertex];
        Vertex src;
        src.state = State(pv.x, pv.y, pv.heading, pv.speed, pv.time);
        src.currentCost = pv.g;
        src.ribbons = make_rm(*W, pool + 4 * (size_t)pv.ribbon_offset, pv.ribbon_count, pv.coverage_completed_time);
        DubinsPath dp;
        for (int i = 0; i < 3; i++) { dp.qi[i] = we.qi[i]; dp.param[i] = we.param[i]; }
        dp.rho = we.rho; dp.type = we.type;
        memset(out + e, 0, sizeof(*out));
        try {
            DubinsWrapper wr;
            wr.fill(dp, we.speed, we.start_time);   // a negative start time reads as "unset" and throws (DubinsWrapper.cpp:19-22)
            if (we.end_time < wr.endTime) wr.updateEndTime(we.end_time);
            Vertex end = connectWrapper(src, 0, wr, we.coverage_allowed != 0);
            finish_edge(*W, cfg, src, end, out + e, child_ribbons ? child_ribbons + (size_t)e * stride * 4 : nullptr, stride);
        } catch (SampleError&) {     // sampling the wrapper's end state threw while connecting
            out[e].flags = PPGPU_F_THROWS | PPGPU_F_INFEASIBLE;
        } catch (const std::exception& ex) {
            out[e].flags = PPGPU_F_THROWS | PPGPU_F_INFEASIBLE;
            note_error("ppo_cost_wrapper_edges", ex);
        }
    }
    return take_error() ? -2 : 0;
}

// Dubins lengths (Edge::computeApproxCost's wrapper.length()), layout as ppgpu_dubins_lengths.
int ppo_dubins_lengths(void* w, const ppgpu_vertex* verts, int v0, int nv, long ns, const double* sx, const double* sy,
                       const double* sh, double* out) {
    World* W = (World*)w;
    Config cfg = make_config(*W);
    for (int v = 0; v < nv; v++) {
        const ppgpu_vertex& pv = verts[v0 + v];
        State a(pv.x, pv.y, pv.heading, pv.speed, pv.time);
        for (long s = 0; s < ns; s++) {
            State b(sx[s], sy[s], sh[s], cfg.maxSpeed, 0);
            for (int r = 0; r < 2; r++) {
                double rho = r ? cfg.coverageTurningRadius : cfg.turningRadius;
                double len = -1;
                if (a.distanceTo(b) > cfg.collisionCheckingIncrement) {
                    DubinsWrapper wr;
                    wr.set(a, b, rho);
                    len = wr.length();
                }
                out[((size_t)v * ns + s) * 2 + r] = len;
            }
        }
    }
    return 0;
}

// ------------------------------------------------------------------ planner
struct ppo_plan_stats {
    uint64_t samples, generated, expanded, iterations;
    double plan_f, plan_collision_penalty, plan_time_penalty, plan_h;
    uint64_t plan_depth;
    int64_t first_goal_iteration;
    int32_t plan_len;   // number of Dubins segments
    int32_t threw;      // 1 = an exception would have left plan()
};

// One AStarPlanner::plan() call with an injected clock now() = clock_t0 + calls*clock_dt
// (PlannerConfig::setNowFunction, PlannerConfig.h:110-114).  Plan segments are returned as
// 11 doubles each: qi[3], param[3], rho, type, speed, start_time, end_time.
// edge_dump (optional): per true-costed edge, 16 doubles:
//   src x,y,heading,speed,time | target x,y,heading,speed | rho, coverageAllowed | infeasible, trueCost, g, h, end_time
int ppo_plan(void* w, int n_ribbons, const double* ribbons4, double cct, const double* start5, int initial_samples,
             int use_brown_paths, int n_prev, const double* prev11, double time_remaining, double clock_t0, double clock_dt,
             ppo_plan_stats* st, double* plan11, int plan_cap, double* iter_best_f, int iter_cap, double* edge_dump,
             long edge_cap, long* n_edges_out) {
    World* W = (World*)w;
    Config cfg = make_config(*W);
    cfg.initialSamples = initial_samples;
    cfg.useBrownPaths = use_brown_paths != 0;
    long calls = 0;
    // clock_dt < 0: the wall clock (clock_t0 + seconds since this call began) — bench.py's plan-level CPU baseline, where the
    // checker's planner is given the same WALL budget as the product's; every parity comparison uses the counting clock
    const std::chrono::steady_clock::time_point wall0 = std::chrono::steady_clock::now();
    if (clock_dt < 0) cfg.now = [&]() { return clock_t0 + std::chrono::duration<double>(std::chrono::steady_clock::now() - wall0).count(); };
    else cfg.now = [&]() { return clock_t0 + (double)(calls++) * clock_dt; };
    RibbonManager rm = make_rm(*W, ribbons4, n_ribbons, cct);
    State start(start5[0], start5[1], start5[2], start5[3], start5[4]);
    std::vector<DubinsWrapper> prev;
    for (int i = 0; i < n_prev; i++) {
        const double* p = prev11 + 11 * i;
        DubinsWrapper wr;
        wr.fill(path_from8(p), p[8], p[9]);
        if (wr.endTime > p[10]) wr.updateEndTime(p[10]);
        prev.push_back(wr);
    }
    AStarPlanner planner;
    long ne = 0;
    if (edge_dump) {
        planner.onEdge = [&](const Vertex& s, const Vertex& e) {
            if (ne < edge_cap) {
                double* d = edge_dump + 16 * ne;
                d[0] = s.state.x; d[1] = s.state.y; d[2] = s.state.heading; d[3] = s.state.speed; d[4] = s.state.time;
                d[5] = e.wrapper.path.param[0]; d[6] = e.wrapper.path.param[1]; d[7] = e.wrapper.path.param[2];
                d[8] = (double)e.wrapper.path.type;
                d[9] = e.turningRadius; d[10] = e.coverageAllowed ? 1 : 0;
                d[11] = e.infeasible ? 1 : 0; d[12] = e.edgeTrueCost; d[13] = e.currentCost; d[14] = e.approxToGo; d[15] = e.state.time;
            }
            ne++;
        };
    }
    memset(st, 0, sizeof(*st));
    Stats s;
    try {
        s = planner.plan(rm, start, cfg, prev, time_remaining);
    } catch (SampleError&) {
        st->threw = 1;
        return 1;
    } catch (std::exception&) {
        st->threw = 1;
        return 1;
    }
    st->samples = s.Samples; st->generated = s.Generated; st->expanded = s.Expanded; st->iterations = s.Iterations;
    st->plan_f = s.PlanFValue; st->plan_collision_penalty = s.PlanCollisionPenalty; st->plan_time_penalty = s.PlanTimePenalty;
    st->plan_h = s.PlanHValue; st->plan_depth = s.PlanDepth; st->first_goal_iteration = s.firstGoalIteration;
    st->plan_len = (int32_t)s.Plan.size();
    for (int i = 0; i < (int)s.Plan.size() && i < plan_cap; i++) {
        double* p = plan11 + 11 * i;
        path_to8(s.Plan[i].path, p);
        p[8] = s.Plan[i].speed; p[9] = s.Plan[i].updatedStartTime; p[10] = s.Plan[i].endTime;
    }
    for (int i = 0; i < (int)s.iterationBestF.size() && i < iter_cap; i++) iter_best_f[i] = s.iterationBestF[i];
    if (n_edges_out) *n_edges_out = ne;
    return 0;
}

// diagnostic: coverage events / ribbon-list mutations per edge (single-threaded)
void ppo_edge_event_stats(void* w, const ppgpu_vertex* verts, const double* pool, const double* sx, const double* sy,
                          const double* sh, long n, const uint64_t* edges, int* stats2) {
    World* W = (World*)w;
    Config cfg = make_config(*W);
    ppgpu_edge_result r;
    for (long e = 0; e < n; e++) {
        cost_one(*W, cfg, verts, pool, sx, sy, sh, edges[e], &r, nullptr, 0);
        stats2[2 * e] = g_last_events; stats2[2 * e + 1] = g_last_mutations;
    }
}
void ppo_edge_event_kinds(void* w, const ppgpu_vertex* verts, const double* pool, const double* sx, const double* sy,
                          const double* sh, long n, const uint64_t* edges, int* kinds6) {
    World* W = (World*)w;
    Config cfg = make_config(*W);
    ppgpu_edge_result r;
    for (long e = 0; e < n; e++) {
        cost_one(*W, cfg, verts, pool, sx, sy, sh, edges[e], &r, nullptr, 0);
        kinds6[6 * e] = g_last_events; kinds6[6 * e + 1] = g_last_mutations;
        for (int q = 0; q < 4; q++) kinds6[6 * e + 2 + q] = g_last_kinds[q];
    }
}

// The sample scan of SamplingBasedPlanner::expand (SamplingBasedPlanner.cpp:82-133) on its own, with sample indices kept: heapify
// the samples by Euclidean distance from the source, visit nearest first, per radius keep a max-heap of the k best by approximate
// cost, stop a radius once its heap is full and the worst kept LENGTH is not above the next distance.  out_idx[r * k + j] = sample
// index at position j of radius r's heap ARRAY when the scan ends (-1 beyond its size): the order in which expand() then costs
// and pushes the children (:134-149).  `samples` keeps the order it is given in (the reference's vector is permuted by the scan;
// with distinct distances the visit order does not depend on the starting permutation).
void ppo_expand_order(void* w, const double* src5, long n, const double* sx, const double* sy, const double* sh, int k, int* out_idx) {
    World* W = (World*)w;
    Config cfg = make_config(*W);
    struct Item { State s; int idx; };
    std::vector<Item> items((size_t)n);
    for (long i = 0; i < n; i++) { items[(size_t)i].s = State(sx[i], sy[i], sh[i], cfg.maxSpeed, 0); items[(size_t)i].idx = (int)i; }
    const State origin(src5[0], src5[1], src5[2], src5[3], src5[4]);
    auto comp = [&](const Item& a, const Item& b) { return a.s.distanceTo(origin) > b.s.distanceTo(origin); };
    const double radii[2] = {cfg.turningRadius, cfg.coverageTurningRadius == cfg.turningRadius ? -1 : cfg.coverageTurningRadius};
    struct Cand { double cost, length; int idx; };
    std::vector<Cand> temps;
    auto dubinsComp = [&](int a, int b) { return temps[(size_t)a].cost < temps[(size_t)b].cost; };
    std::vector<int> heaps[2];
    bool done[2] = {false, false};
    std::make_heap(items.begin(), items.end(), comp);
    for (size_t i = 0; i < items.size() && (!done[0] || !done[1]); i++) {
        Item sample = items.front();
        std::pop_heap(items.begin(), items.end() - (long)i, comp);
        for (int j = 0; j < 2; j++) {
            if (done[j]) continue;
            if (radii[j] <= 0) { done[j] = true; continue; }
            auto& best = heaps[j];
            if (best.size() < (size_t)k || temps[(size_t)best.front()].length > sample.s.distanceTo(origin)) {
                if (origin.distanceTo(sample.s) > cfg.collisionCheckingIncrement) {
                    sample.s.speed = cfg.maxSpeed;
                    DubinsWrapper wr;
                    wr.set(origin, sample.s, radii[j]);
                    temps.push_back(Cand{wr.length() / sample.s.speed * 1.0, wr.length(), sample.idx});
                    best.push_back((int)temps.size() - 1);
                    std::push_heap(best.begin(), best.end(), dubinsComp);
                    if (best.size() > (size_t)k) {
                        std::pop_heap(best.begin(), best.end(), dubinsComp);
                        best.pop_back();
                    }
                }
            } else {
                done[j] = true;
            }
        }
    }
    for (int j = 0; j < 2; j++)
        for (int q = 0; q < k; q++) out_idx[j * k + q] = q < (int)heaps[j].size() ? temps[(size_t)heaps[j][(size_t)q]].idx : -1;
}

// SamplingBasedPlanner::expand on a root vertex, the way the reference's ExpandTest1Ribbons drives it
// (test_planner.cpp:1061-1082): a ribbon-less StateGenerator whose first `gen_skip` states are consumed
// elsewhere, addSamples(generator, n_samples), expand(root), then pop the whole queue.
// out_f receives the popped f values in pop order; returns how many were popped (-1 if something threw).
int ppo_expand_once(void* w, int n_ribbons, const double* ribbons4, const double* root5, const double* b6, uint64_t seed,
                    long gen_skip, int n_samples, double* out_f, int cap, uint64_t* generated, uint64_t* expanded,
                    uint64_t* kept_samples) {
    World* W = (World*)w;
    AStarPlanner pl;
    pl.cfg = make_config(*W);
    pl.startStateTime = pl.cfg.startStateTime;
    StateGenerator g(b6[0], b6[1], b6[2], b6[3], b6[4], b6[5], seed);
    for (long i = 0; i < gen_skip; i++) g.generate();
    RibbonManager rm = make_rm(*W, ribbons4, n_ribbons, -1);
    State root(root5[0], root5[1], root5[2], root5[3], root5[4]);
    pl.arena.push_back(makeRoot(root, rm));
    computeApproxToGo(pl.arena[0], pl.cfg);
    pl.addSamples(g, n_samples);
    int n = 0;
    try {
        pl.expand(0);
        int v;
        while ((v = pl.popVertexQueue()) >= 0) {
            if (n < cap) out_f[n] = pl.arena[v].f();
            n++;
        }
    } catch (...) {
        return -1;
    }
    if (generated) *generated = pl.stats.Generated;
    if (expanded) *expanded = pl.stats.Expanded;
    if (kept_samples) *kept_samples = pl.samples.size();
    return n;
}

int ppo_hardware_threads() { return (int)std::thread::hardware_concurrency(); }

}  // extern "C"
