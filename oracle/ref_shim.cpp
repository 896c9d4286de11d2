// ref_shim.cpp — TEST INFRASTRUCTURE ONLY.  C entry points over the REFERENCE's own classes,
// compiled together with the reference sources where they lie (see Makefile target `ref`);
// the result (oracle/_ref/libpp_ref.so) is git-ignored and is used only to validate
// pp_oracle.cpp and to generate tests/golden/*.json (tests/golden/make_golden.py).
// Nothing from the reference is copied here: this file only #includes its headers.
#include <path_planner_common/State.h>
#include "planner/utilities/Ribbon.h"
#include "common/map/GridWorldMap.h"
#include "common/dynamic_obstacles/BinaryDynamicObstaclesManager.h"

#include <string>

static void put(const State& s, double* o) { o[0] = s.x(); o[1] = s.y(); o[2] = s.heading(); o[3] = s.speed(); o[4] = s.time(); }
static State get(const double* s) { return State(s[0], s[1], s[2], s[3], s[4]); }

extern "C" {
// ---- State
double ref_state_yaw(double heading) { return State(0, 0, heading, 0, 0).yaw(); }
double ref_state_heading_to(double x, double y, double x1, double y1) { return State(x, y, 0, 0, 0).headingTo(x1, y1); }
void ref_state_move(double* s5, double d) { State s = get(s5); s.move(d); put(s, s5); }
void ref_state_push(const double* s5, double dt, double* out5) { put(get(s5).push(dt), out5); }
double ref_state_distance_to(const double* s5, double x, double y) { return get(s5).distanceTo(x, y); }
double ref_state_heading_difference(const double* s5, double other) { return get(s5).headingDifference(other); }

// ---- Ribbon
void ref_set_ribbon_width(double w) { Ribbon::RibbonWidth = w; }
double ref_ribbon_min_length() { return Ribbon::minLength(); }
void ref_ribbon_projection(const double* r4, double x, double y, double* out2) {
    auto p = Ribbon(r4[0], r4[1], r4[2], r4[3]).getProjection(x, y);
    out2[0] = p.first; out2[1] = p.second;
}
int ref_ribbon_contains(const double* r4, double x, double y, int strict) {
    Ribbon r(r4[0], r4[1], r4[2], r4[3]);
    return r.contains(x, y, r.getProjection(x, y), strict != 0) ? 1 : 0;
}
int ref_ribbon_contains_projection(const double* r4, double px, double py) {
    return Ribbon(r4[0], r4[1], r4[2], r4[3]).containsProjection(std::make_pair(px, py)) ? 1 : 0;
}
double ref_ribbon_distance(const double* r4, double x, double y) { return Ribbon(r4[0], r4[1], r4[2], r4[3]).distance(x, y); }
double ref_ribbon_length(const double* r4) { return Ribbon(r4[0], r4[1], r4[2], r4[3]).length(); }
int ref_ribbon_covered(const double* r4, int strict) { return Ribbon(r4[0], r4[1], r4[2], r4[3]).covered(strict != 0) ? 1 : 0; }
void ref_ribbon_split(double* r4, double x, double y, int strict, double* out4) {
    Ribbon r(r4[0], r4[1], r4[2], r4[3]);
    Ribbon f = r.split(x, y, strict != 0);
    r4[0] = r.start().first; r4[1] = r.start().second; r4[2] = r.end().first; r4[3] = r.end().second;
    out4[0] = f.start().first; out4[1] = f.start().second; out4[2] = f.end().first; out4[3] = f.end().second;
}
void ref_ribbon_end_states(const double* r4, double* start5, double* end5) {
    Ribbon r(r4[0], r4[1], r4[2], r4[3]);
    put(r.startAsState(), start5);
    put(r.endAsState(), end5);
}
void ref_ribbon_projection_as_state(const double* r4, double x, double y, double* out5) {
    put(Ribbon(r4[0], r4[1], r4[2], r4[3]).getProjectionAsState(x, y), out5);
}

// ---- Map
void* ref_grid_load(const char* path) { return new GridWorldMap(std::string(path)); }
void ref_grid_free(void* g) { delete (GridWorldMap*)g; }
int ref_grid_is_blocked(void* g, double x, double y) { return ((GridWorldMap*)g)->isBlocked(x, y) ? 1 : 0; }
void ref_grid_is_blocked_many(void* g, long n, const double* x, const double* y, unsigned char* out) {
    for (long i = 0; i < n; i++) out[i] = ((GridWorldMap*)g)->isBlocked(x[i], y[i]) ? 1 : 0;
}
void ref_grid_extremes(void* g, double* out4) { const double* e = ((GridWorldMap*)g)->extremes(); for (int i = 0; i < 4; i++) out4[i] = e[i]; }
double ref_grid_resolution(void* g) { return ((GridWorldMap*)g)->resolution(); }
int ref_base_map_is_blocked(double x, double y) { Map m; return m.isBlocked(x, y) ? 1 : 0; }
void ref_base_map_extremes(double* out4) { Map m; const double* e = m.extremes(); for (int i = 0; i < 4; i++) out4[i] = e[i]; }

// ---- BinaryDynamicObstaclesManager
void* ref_obst_create() { return new BinaryDynamicObstaclesManager(); }
void ref_obst_free(void* m) { delete (BinaryDynamicObstaclesManager*)m; }
void ref_obst_update(void* m, unsigned mmsi, double x, double y, double heading, double speed, double time, double width, double length) {
    ((BinaryDynamicObstaclesManager*)m)->update(mmsi, x, y, heading, speed, time, width, length);
}
double ref_obst_collision_exists(void* m, double x, double y, double t, int strict) {
    return ((BinaryDynamicObstaclesManager*)m)->collisionExists(x, y, t, strict != 0);
}
}
