// C entry points over the host-side classes (State, Ribbon, RibbonManager, GridWorldMap, the obstacle managers, DubinsWrapper)
// so that the CPU test suite can hold them against the reference's golden vectors (tests/test_golden.py) and against the CPU
// oracle (tests/test_host_cpu.py) without a GPU.  Arrays are plain doubles: a state is {x, y, heading, speed, time}, a ribbon
// {startX, startY, endX, endY}.  Nothing here touches the device.
#include <cstring>
#include <sstream>

#include "path_planner_amd/DubinsWrapper.h"
#include "path_planner_amd/RibbonManager.h"
#include "path_planner_amd/World.h"

using namespace ppamd;

namespace {
State stateOf(const double* s5) { return State(s5[0], s5[1], s5[2], s5[3], s5[4]); }
void put(const State& s, double* out5) { out5[0] = s.x(); out5[1] = s.y(); out5[2] = s.heading(); out5[3] = s.speed(); out5[4] = s.time(); }
Ribbon ribbonOf(const double* r4) { return Ribbon(r4[0], r4[1], r4[2], r4[3]); }
void put(const Ribbon& r, double* out4) { std::memcpy(out4, r.row(), 4 * sizeof(double)); }
RibbonManager managerOf(const double* ribbons4, int n, int heuristic = 0, double turningRadius = -1, int k = 0) {
    RibbonManager m((RibbonManager::Heuristic)heuristic, turningRadius, k);
    m.assign(ribbons4, n, -1);
    return m;
}
int store(const RibbonManager& m, double* ribbons4, int cap) {
    const int n = m.count() < cap ? m.count() : cap;
    if (n > 0) std::memcpy(ribbons4, m.rows(), (size_t)n * 4 * sizeof(double));
    return m.count();
}
}  // namespace

extern "C" {

// ---- State
double pph_state_yaw(double heading) { return State(0, 0, heading, 0, 0).yaw(); }
double pph_state_heading_to(double x, double y, double x1, double y1) { return State(x, y, 0, 0, 0).headingTo(x1, y1); }
double pph_state_distance_to(const double* s5, double x1, double y1) { return stateOf(s5).distanceTo(x1, y1); }
double pph_state_heading_difference(const double* s5, double other) { return stateOf(s5).headingDifference(other); }
void pph_state_move(double* s5, double d) { State s = stateOf(s5); s.move(d); put(s, s5); }
void pph_state_push(const double* s5, double dt, double* out5) { put(stateOf(s5).push(dt), out5); }
void pph_state_interpolate(const double* a5, const double* b5, double t, double* out5) { put(stateOf(a5).interpolate(stateOf(b5), t), out5); }
int pph_state_to_string(const double* s5, int radians, char* out, int cap) {
    const std::string s = radians ? stateOf(s5).toStringRad() : stateOf(s5).toString();
    std::strncpy(out, s.c_str(), (size_t)cap);
    return (int)s.size();
}

// ---- one ribbon
void pph_set_ribbon_width(double w) { RibbonManager::setRibbonWidth(w); }
double pph_get_ribbon_width() { return Ribbon::RibbonWidth; }
void pph_ribbon_projection(const double* r4, double x, double y, double* out2) {
    const auto p = ribbonOf(r4).getProjection(x, y);
    out2[0] = p.first; out2[1] = p.second;
}
int pph_ribbon_contains(const double* r4, double x, double y, int strict) {
    const Ribbon r = ribbonOf(r4);
    return r.contains(x, y, r.getProjection(x, y), strict != 0) ? 1 : 0;
}
int pph_ribbon_contains_projection(const double* r4, double px, double py) { return ribbonOf(r4).containsProjection({px, py}) ? 1 : 0; }
double pph_ribbon_distance(const double* r4, double x, double y) { return ribbonOf(r4).distance(x, y); }
double pph_ribbon_length(const double* r4) { return ribbonOf(r4).length(); }
int pph_ribbon_covered(const double* r4, int strict) { return ribbonOf(r4).covered(strict != 0) ? 1 : 0; }
void pph_ribbon_split(double* r4, double x, double y, int strict, double* front4) {
    Ribbon r = ribbonOf(r4);
    put(r.split(x, y, strict != 0), front4);
    put(r, r4);
}
void pph_ribbon_end_states(const double* r4, double* start5, double* end5) {
    put(ribbonOf(r4).startAsState(), start5);
    put(ribbonOf(r4).endAsState(), end5);
}

// ---- a list of ribbons
int pph_ribbons_add(double* ribbons4, int n, int cap, double x1, double y1, double x2, double y2) {
    RibbonManager m = managerOf(ribbons4, n);
    m.add(x1, y1, x2, y2);
    return store(m, ribbons4, cap);
}
int pph_ribbons_cover(double* ribbons4, int n, int cap, double x, double y, int strict) {
    RibbonManager m = managerOf(ribbons4, n);
    m.cover(x, y, strict != 0);
    return store(m, ribbons4, cap);
}
int pph_ribbons_cover_between(double* ribbons4, int n, int cap, double x1, double y1, double x2, double y2, int strict) {
    RibbonManager m = managerOf(ribbons4, n);
    m.coverBetween(x1, y1, x2, y2, strict != 0);
    return store(m, ribbons4, cap);
}
double pph_ribbons_min_distance(const double* ribbons4, int n, double x, double y) { return managerOf(ribbons4, n).minDistanceFrom(x, y); }
double pph_ribbons_heuristic(const double* ribbons4, int n, int heuristic, int K, double turningRadius, double x, double y, double yaw) {
    return managerOf(ribbons4, n, heuristic, turningRadius, K).approximateDistanceUntilDone(x, y, yaw);
}
int pph_ribbons_nearest_endpoint(const double* ribbons4, int n, const double* s5, double* out5) {
    const RibbonManager m = managerOf(ribbons4, n);
    if (m.done()) return 1;
    put(m.getNearestEndpointAsState(stateOf(s5)), out5);
    return 0;
}
void pph_ribbons_project(const double* ribbons4, int n, double* s5) {
    State s = stateOf(s5);
    managerOf(ribbons4, n).projectOntoNearestRibbon(s);
    put(s, s5);
}
int pph_ribbons_near_states(const double* ribbons4, int n, const double* start5, double radius, double* out5, int cap) {
    const std::vector<State> v = managerOf(ribbons4, n).findNearStatesOnRibbons(stateOf(start5), radius);
    for (size_t i = 0; i < v.size() && (int)i < cap; i++) put(v[i], out5 + 5 * i);
    return (int)v.size();
}
double pph_ribbons_total_uncovered_length(const double* ribbons4, int n) { return managerOf(ribbons4, n).getTotalUncoveredLength(); }
int pph_ribbons_dump(const double* ribbons4, int n, char* out, int cap) {
    const std::string s = managerOf(ribbons4, n).dumpRibbons();
    std::strncpy(out, s.c_str(), (size_t)cap);
    return (int)s.size();
}

// ---- maps
void* pph_grid_load_text(const char* text, int* rows, int* cols, double* res) {
    try {
        auto* m = new std::shared_ptr<GridWorldMap>(GridWorldMap::fromText(text));
        *rows = (*m)->rows(); *cols = (*m)->cols(); *res = (*m)->resolution();
        return m;
    } catch (const std::exception&) {
        return nullptr;
    }
}
void pph_grid_free(void* h) { delete static_cast<std::shared_ptr<GridWorldMap>*>(h); }
void pph_grid_extremes(void* h, double* out4) { std::memcpy(out4, (*static_cast<std::shared_ptr<GridWorldMap>*>(h))->extremes(), 4 * sizeof(double)); }
void pph_grid_is_blocked_many(void* h, long n, const double* x, const double* y, unsigned char* out) {
    const GridWorldMap& m = **static_cast<std::shared_ptr<GridWorldMap>*>(h);
    for (long i = 0; i < n; i++) out[i] = m.isBlocked(x[i], y[i]) ? 1 : 0;
}
void pph_grid_cells(void* h, unsigned char* out) {
    std::vector<uint8_t> cells; int r, c; double res;
    (*static_cast<std::shared_ptr<GridWorldMap>*>(h))->rasterize(cells, r, c, res);
    std::memcpy(out, cells.data(), cells.size());
}
int pph_base_map_is_blocked(double x, double y) { return Map().isBlocked(x, y) ? 1 : 0; }
void pph_base_map_extremes(double* out4) { Map m; std::memcpy(out4, m.extremes(), 4 * sizeof(double)); }

// ---- contacts
void* pph_obst_create(int gaussian) {
    if (gaussian) return new std::shared_ptr<DynamicObstaclesManager>(std::make_shared<GaussianDynamicObstaclesManager>());
    return new std::shared_ptr<DynamicObstaclesManager>(std::make_shared<BinaryDynamicObstaclesManager>());
}
void pph_obst_free(void* h) { delete static_cast<std::shared_ptr<DynamicObstaclesManager>*>(h); }
void pph_obst_update(void* h, unsigned mmsi, double x, double y, double heading, double speed, double time, double width, double length) {
    auto& m = *static_cast<std::shared_ptr<DynamicObstaclesManager>*>(h);
    if (auto* b = dynamic_cast<BinaryDynamicObstaclesManager*>(m.get())) b->update(mmsi, x, y, heading, speed, time, width, length);
    else if (auto* g = dynamic_cast<GaussianDynamicObstaclesManager*>(m.get())) g->update(mmsi, x, y, heading, speed, time);
}
void pph_obst_update_gaussian(void* h, unsigned mmsi, double x, double y, double heading, double speed, double time, const double* cov4) {
    auto& m = *static_cast<std::shared_ptr<DynamicObstaclesManager>*>(h);
    if (auto* g = dynamic_cast<GaussianDynamicObstaclesManager*>(m.get())) g->update(mmsi, x, y, heading, speed, time, cov4);
}
void pph_obst_forget(void* h, unsigned mmsi) {
    auto& m = *static_cast<std::shared_ptr<DynamicObstaclesManager>*>(h);
    if (auto* b = dynamic_cast<BinaryDynamicObstaclesManager*>(m.get())) b->forget(mmsi);
    else if (auto* g = dynamic_cast<GaussianDynamicObstaclesManager*>(m.get())) g->forget(mmsi);
}
double pph_obst_collision_exists(void* h, double x, double y, double t, int strict) {
    return (*static_cast<std::shared_ptr<DynamicObstaclesManager>*>(h))->collisionExists(x, y, t, strict != 0);
}
int pph_obst_device_rows(void* h, double* out, int cap_doubles) {
    std::vector<double> rows;
    (*static_cast<std::shared_ptr<DynamicObstaclesManager>*>(h))->deviceRows(rows);
    for (size_t i = 0; i < rows.size() && (int)i < cap_doubles; i++) out[i] = rows[i];
    return (int)rows.size();
}

// ---- timed curves: path8 = {qi[3], param[3], rho, type}
int pph_wrapper_sample(const double* path8, double speed, double startTime, double endTime, double t, double* out5) {
    DubinsPath p;
    for (int i = 0; i < 3; i++) { p.qi[i] = path8[i]; p.param[i] = path8[3 + i]; }
    p.rho = path8[6]; p.type = (DubinsPathType)(int)path8[7];
    try {
        DubinsWrapper w;
        w.fill(p, speed, startTime);
        if (endTime < w.getEndTime()) w.updateEndTime(endTime);
        State s;
        s.time() = t;
        w.sample(s);
        put(s, out5);
        return 0;
    } catch (const std::exception&) {
        return 1;
    }
}
double pph_wrapper_solve(const double* from5, const double* to5, double rho, double* path8) {
    DubinsWrapper w(stateOf(from5), stateOf(to5), rho);
    const DubinsPath& p = w.unwrap();
    for (int i = 0; i < 3; i++) { path8[i] = p.qi[i]; path8[3 + i] = p.param[i]; }
    path8[6] = p.rho; path8[7] = (double)p.type;
    return w.getEndTime();
}

}  // extern "C"
