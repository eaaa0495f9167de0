"""Host-side geometry / path utilities of the navigator -- the per-episode, O(path length)
quantities that ``NND_MB_agent.start_new_episode_plan`` computes once per SmartStart episode
(smartstart/RLAgents/NND_MB_agent.py:375-423) and uploads as kernel arguments (SURVEY.md
Appendix A).  Counterparts of smartstart/utilities/numerical.py; numpy fp64 like the reference.

These run on the host by design: they are tiny, sequential and executed once per episode; the
per-step work (forward simulation, scoring) is on the GPU (csrc/dyn_*.hip, csrc/mpc.hip).
"""
from __future__ import annotations

import bisect

import numpy as np


def path_deltas_stds_and_means_per_dim(path):
    """numerical.py:30-59: per state dimension, std and mean of |path[i+1] - path[i]|."""
    p = np.asarray(path, dtype=np.float64)
    if len(p) <= 1:
        raise ValueError("path needs at least two states")
    deltas = np.abs(np.diff(p, axis=0))
    return deltas.std(axis=0), deltas.mean(axis=0)


def radii_calc(means, stds, num_means, num_stds, num_steps):
    """numerical.py:61-62"""
    return ((num_means * np.asarray(means)) + (num_stds * np.asarray(stds))) * num_steps


def elliptical_euclidean_distance_function_generator(radii):
    """numerical.py:101-126: d(x, y) = || (x - y) / radii ||_2 over the last axis."""
    radii = np.asarray(radii, dtype=np.float64)
    if not (radii > 0).all():
        raise AssertionError("radii must be positive")

    def distance_func(state, other_state):
        state = np.asarray(state, dtype=np.float64)
        other_state = np.asarray(other_state, dtype=np.float64)
        return np.sqrt(np.sum(((state - other_state) / radii) ** 2, axis=max(state.ndim, other_state.ndim) - 1))

    distance_func.radii = radii          # lets path_shortcutter hand the whole job to libssc's host routine
    return distance_func


def volume_of_n_dimensional_hyperellipsoid(radii):
    """numerical.py:157-164 (the reference's ``np.product`` is gone in numpy 2)."""
    from math import gamma, pi
    dim = len(radii)
    return ((pi ** (dim / 2.0)) / gamma((dim / 2.0) + 1)) * float(np.prod(radii))


def _length_weighted_activities_loop(activities, sub_extra=0):
    """The reference's loop, interval by interval (numerical.py:189-222)."""
    acts = sorted((tuple(int(v) for v in a) for a in activities), key=lambda a: a[1])
    if not acts:
        return 0, []
    ends = [0, acts[0][1]]                       # row end times (strictly increasing)
    best = [0, acts[0][1] - acts[0][0]]          # best weight using intervals ending <= ends[i]
    take = [None, acts[0]]                       # interval taken at row i (or None)
    back = [0, 0]                                # previous row
    for a in acts[1:]:
        j = bisect.bisect_right(ends, a[0]) - 1  # last row with end <= start
        inc = best[j] + (a[1] - a[0] - sub_extra)
        if a[1] == ends[-1]:
            if inc >= best[-1]:
                best[-1], take[-1], back[-1] = inc, a, j
        else:
            ends.append(a[1])
            if inc >= best[-1]:
                best.append(inc); take.append(a); back.append(j)
            else:
                best.append(best[-1]); take.append(None); back.append(len(ends) - 2)
    chosen, i = [], len(ends) - 1
    while True:
        if take[i] is not None:
            chosen.append(list(take[i]))
        if back[i] == i:
            break
        i = back[i]
    chosen.reverse()
    return best[-1], chosen


def length_weighted_activities_solver(activities, sub_extra=0):
    """numerical.py:189-222: weighted interval scheduling, weight = end - start - sub_extra, touching
    intervals compatible, ties resolved in favour of taking the later-ending interval; the first
    interval (by end time) is seeded without ``sub_extra`` exactly like the reference.

    Same table as the reference's loop, filled one END TIME at a time: all intervals that share an end form one row,
    and within a row the reference's ``inc >= best`` rule keeps the LAST interval that reaches the row's maximum -- so a
    row is one vectorised max.  A 300-state smart-start path yields tens of thousands of candidate shortcuts; the
    interval-by-interval loop took 0.35-0.5 s per plan (the serial part of the vectorised SmartStart loop), this
    takes milliseconds.  Small inputs run the loop itself."""
    A = np.asarray(activities, dtype=np.int64).reshape(-1, 2)
    if A.shape[0] <= 64 or A.min() < 0 or bool((A[:, 0] >= A[:, 1]).any()):
        return _length_weighted_activities_loop(A.tolist(), sub_extra)
    A = A[np.argsort(A[:, 1], kind="stable")]            # stable like sorted(): equal ends keep their order
    ends_u, first = np.unique(A[:, 1], return_index=True)
    bounds = np.append(first, A.shape[0])
    n_rows = len(ends_u) + 1
    ends = np.zeros(n_rows, np.int64)
    best = np.zeros(n_rows, np.int64)
    take = np.full(n_rows, -1, np.int64)                  # start of the interval taken at this row, -1: none
    back = np.zeros(n_rows, np.int64)
    for g in range(len(ends_u)):
        e, S = int(ends_u[g]), A[bounds[g]:bounds[g + 1], 0]
        row = g + 1
        ends[row] = e
        if g == 0:                                        # the first interval seeds the table without sub_extra
            best[row], take[row], back[row] = e - int(S[0]), int(S[0]), 0
            S = S[1:]
            if S.size == 0:
                continue
            prev_best = best[row]
            j = np.searchsorted(ends[:row + 1], S, side="right") - 1
        else:
            best[row], take[row], back[row] = best[row - 1], -1, row - 1
            prev_best = best[row - 1]
            j = np.searchsorted(ends[:row], S, side="right") - 1
        inc = best[j] + (e - S - sub_extra)
        m = int(inc.max())
        if m >= prev_best:
            k = int(np.flatnonzero(inc == m)[-1])         # ">=" lets every later tie replace the earlier one
            best[row], take[row], back[row] = m, int(S[k]), int(j[k])
    chosen, i = [], n_rows - 1
    while True:
        if take[i] >= 0:
            chosen.append([int(take[i]), int(ends[i])])
        if back[i] == i:
            break
        i = int(back[i])
    chosen.reverse()
    return int(best[-1]), chosen


def _native_path_shortcut(a, radii, theta):
    """``ssc_path_shortcut`` (csrc/path_geometry.cpp): the same decisions in native host code -- a 300-state path takes
    0.1 ms instead of 1.5-6.5 ms, and eight plans per smart-start selection are the serial part of the vectorised loop."""
    import ctypes
    from . import _ffi
    keep = np.ones(a.shape[0], np.uint8)
    r = np.ascontiguousarray(radii, np.float64)
    _ffi.check(_ffi.lib().ssc_path_shortcut(a.ctypes.data_as(ctypes.c_void_p), a.shape[0], a.shape[1],
                                            r.ctypes.data_as(ctypes.c_void_p), float(theta),
                                            keep.ctypes.data_as(ctypes.c_void_p), None))
    return a[keep.astype(bool)]


def path_shortcutter(path, distance_func, theta, native=True):
    """numerical.py:226-246: drop interior states between any two states (>= 2 apart) that are within
    ``theta`` of each other, choosing the non-overlapping shortcuts that delete the most states.
    With the elliptical distance of this module the work is done by libssc's host routine (``native=False``: numpy)."""
    a = np.ascontiguousarray(path, dtype=np.float64)
    radii = getattr(distance_func, "radii", None)
    # (d < 8: the native routine sums the squared terms in index order, like np.sum over a last axis shorter than numpy's
    # 8-accumulator pairwise unrolling; at d == 8 numpy associates differently and a pair within an ulp of theta could flip)
    if native and radii is not None and a.ndim == 2 and 1 <= a.shape[1] < 8 and a.shape[1] == len(radii):
        return _native_path_shortcut(a, radii, theta)
    dist = distance_func(a[:, None, :], a[None, :, :])
    pairs = np.transpose(np.where(np.triu(dist <= theta, k=2)))
    _, chosen = length_weighted_activities_solver(pairs, sub_extra=1)
    drop = [k for i, j in chosen for k in range(i + 1, j)]
    return np.delete(a, drop, axis=0)


def get_start_waypoints_final_states_steps(path, steps_per_waypoint):
    """smartstart/utilities/utilities.py:59-71: path[:-1:steps] + [path[-1]]."""
    p = np.asarray(path, dtype=np.float64)
    return np.concatenate([p[:-1:steps_per_waypoint], p[-1:]], axis=0)


def distances_left(desired_states, distance_func):
    """NND_MB_agent.py:411-418: remaining path length from each waypoint (last entry 0)."""
    wp = np.asarray(desired_states, dtype=np.float64)
    if len(wp) < 2:
        return np.asarray([0.0])
    seg = distance_func(wp[:-1], wp[1:])
    return np.concatenate([np.cumsum(seg[::-1])[::-1], [0.0]])
