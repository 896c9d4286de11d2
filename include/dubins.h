/*
 * dubins.h — C API of the Dubins-curve library the planner compiles against.
 *
 * The reference includes this header from a third-party catkin package,
 * `dubins_curves` (no version pinned: /root/reference/path_planner/package.xml:31,
 * path_planner_common/package.xml:31), which is NOT part of the reference tree.
 * Call sites this header has to satisfy:
 *   path_planner_common/src/dubinsPlan/DubinsWrapper.cpp:13   dubins_shortest_path
 *   path_planner_common/src/dubinsPlan/DubinsWrapper.cpp:21   dubins_path_length  (const method)
 *   path_planner_common/src/dubinsPlan/DubinsWrapper.cpp:38,41 dubins_path_sample (const method), EDUBPARAM
 *   path_planner_common/src/dubinsPlan/DubinsWrapper.cpp:43   EDUBOK
 *   path_planner_common/src/dubinsPlan/DubinsWrapper.cpp:114  dubins_extract_subpath
 *   path_planner/src/planner/utilities/RibbonManager.h:212-215
 *   path_planner/src/NodeBase.h:205-212  (fields qi, param, rho, type serialised into DubinsPath.msg)
 *   path_planner_common/msg/DubinsPath.msg:17  (word order LSL, LSR, RSL, RSR, RLR, LRL = 0..5)
 *
 * The implementation (path_planner_amd/csrc/dubins.c) is written from scratch from the
 * published six-word Dubins classification (Shkel & Lumelsky 2001) in its usual
 * normalised form; see DESIGN.md "Dubins" for what pins it.
 */
#ifndef PATH_PLANNER_AMD_DUBINS_H
#define PATH_PLANNER_AMD_DUBINS_H

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    LSL = 0,
    LSR = 1,
    RSL = 2,
    RSR = 3,
    RLR = 4,
    LRL = 5
} DubinsPathType;

typedef struct {
    double qi[3];        /* initial configuration (x, y, yaw) */
    double param[3];     /* lengths of the three segments, in units of rho */
    double rho;          /* turning radius */
    DubinsPathType type; /* which of the six words */
} DubinsPath;

#define EDUBOK        (0) /* no error */
#define EDUBCOCONFIGS (1) /* colocated configurations */
#define EDUBPARAM     (2) /* path parameterisation error */
#define EDUBBADRHO    (3) /* rho is invalid */
#define EDUBNOPATH    (4) /* no connection between configurations with this word */

/* Shortest of the six words from q0 to q1 (x, y, yaw) at turning radius rho. */
int dubins_shortest_path(DubinsPath* path, double q0[3], double q1[3], double rho);

/* Total length of the path (same units as rho). */
double dubins_path_length(const DubinsPath* path);

/* Configuration at arc length t in [0, length]; EDUBPARAM outside it. */
int dubins_path_sample(const DubinsPath* path, double t, double q[3]);

/* The prefix [0, t] of path as a new path. */
int dubins_extract_subpath(const DubinsPath* path, double t, DubinsPath* newpath);

#ifdef __cplusplus
}
#endif

#endif
