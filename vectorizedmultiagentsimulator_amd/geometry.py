"""Scenario-side geometric queries on the packed state: ``get_distance``,
``get_distance_from_point``, ``is_overlapping`` (reference: vmas/simulator/core.py:1788-1969,
closest-point routines vmas/simulator/physics.py:13-429).

These run between steps from ``reward/observation/done`` (balance: line/package on the floor,
package on the goal - balance.py:218-221,260-263; transport: package on goal - transport.py:145;
navigation: agent-agent distance - navigation.py:218-229).  They are written as plain batched
torch ops over ``[B, ...]`` views (a handful of launches per query, no host sync); the step
kernels do not use them.  Formulas follow SURVEY.md Appendix A.4 / A.9, candidate order and
strict-``<`` tie breaking included.
"""
from __future__ import annotations

from typing import Tuple

import torch
from torch import Tensor

LINE_MIN_DIST = 4 / 6e2


def _norm(v: Tensor) -> Tensor:
    return torch.linalg.vector_norm(v, dim=-1)


def closest_point_line(pos: Tensor, rot: Tensor, length, p: Tensor, limit: bool = True) -> Tensor:
    """Closest point to ``p`` on the segment (centre ``pos`` [..,2], angle ``rot`` [..,1])."""
    u = torch.cat([rot.cos(), rot.sin()], dim=-1)
    dot = ((pos - p) * u).sum(-1, keepdim=True)
    m = dot.abs()
    if limit:
        m = m.clamp(max=length / 2)  # python scalar: no host->device copy (HIP-graph capturable)
    return pos - torch.sign(dot) * m * u


def box_edges(pos: Tensor, rot: Tensor, length: float, width: float):
    """Four edge segments (centre, rot, length) of a box, order p1..p4 of physics.py:298-325."""
    u = torch.cat([rot.cos(), rot.sin()], dim=-1)
    rot2 = rot + torch.pi / 2
    u2 = torch.cat([rot2.cos(), rot2.sin()], dim=-1)
    hl, hw = length / 2, width / 2
    return [
        (pos + u * hl, rot2, width),
        (pos - u * hl, rot2, width),
        (pos + u2 * hw, rot, length),
        (pos - u2 * hw, rot, length),
    ]


def _keep_closest(cands):
    """cands: list of (p1, p2); strict '<' in order, starting from +inf."""
    best1 = torch.full_like(cands[0][0], float("inf"))
    best2 = torch.full_like(cands[0][1], float("inf"))
    dist = torch.full(cands[0][0].shape[:-1], float("inf"), dtype=best1.dtype, device=best1.device)
    for p1, p2 in cands:
        d = _norm(p1 - p2)
        c = d < dist
        ce = c.unsqueeze(-1)
        best1, best2, dist = torch.where(ce, p1, best1), torch.where(ce, p2, best2), torch.where(c, d, dist)
    return best1, best2


def closest_point_box(pos: Tensor, rot: Tensor, length: float, width: float, p: Tensor) -> Tensor:
    cands = [(closest_point_line(ep, er, el, p), p) for ep, er, el in box_edges(pos, rot, length, width)]
    return _keep_closest(cands)[0]


def _extrema(pos, rot, length):
    xy = torch.cat([rot.cos(), rot.sin()], dim=-1) * (length / 2)
    return pos + xy, pos - xy


def closest_points_seg_seg(pos1, rot1, len1, pos2, rot2, len2) -> Tuple[Tensor, Tensor]:
    a1, a2 = _extrema(pos1, rot1, len1)
    b1, b2 = _extrema(pos2, rot2, len2)
    r, s, qp = a2 - a1, b2 - b1, b1 - a1

    def cross(a, b):
        return (a[..., 0] * b[..., 1] - a[..., 1] * b[..., 0]).unsqueeze(-1)

    crs = cross(r, s)
    u, t = cross(qp, r) / crs, cross(qp, s) / crs
    hit = (crs != 0) & (0 <= u) & (u <= 1) & (0 <= t) & (t <= 1)
    pi = a1 + t * r
    q1, q2 = _keep_closest(
        [
            (a1, closest_point_line(pos2, rot2, len2, a1)),
            (a2, closest_point_line(pos2, rot2, len2, a2)),
            (closest_point_line(pos1, rot1, len1, b1), b1),
            (closest_point_line(pos1, rot1, len1, b2), b2),
        ]
    )
    return torch.where(hit, pi, q1), torch.where(hit, pi, q2)


def closest_seg_box(bpos, brot, blen, bwid, lpos, lrot, llen) -> Tuple[Tensor, Tensor]:
    """(on box, on line)"""
    return _keep_closest([closest_points_seg_seg(ep, er, el, lpos, lrot, llen) for ep, er, el in box_edges(bpos, brot, blen, bwid)])


def closest_box_box(pa, ra, la, wa, pb, rb, lb, wb) -> Tuple[Tensor, Tensor]:
    """(on A, on B)"""
    cands = []
    for ep, er, el in box_edges(pa, ra, la, wa):
        on_b, on_a = closest_seg_box(pb, rb, lb, wb, ep, er, el)
        cands.append((on_a, on_b))
    for ep, er, el in box_edges(pb, rb, lb, wb):
        on_a, on_b = closest_seg_box(pa, ra, la, wa, ep, er, el)
        cands.append((on_a, on_b))
    return _keep_closest(cands)


def _shape_name(e) -> str:
    return type(e.shape).__name__


def get_distance_from_point(entity, p: Tensor) -> Tensor:
    """core.py:1788-1820"""
    s = _shape_name(entity)
    pos, rot = entity.state.pos, entity.state.rot
    if s == "Sphere":
        return _norm(pos - p) - entity.shape.radius
    if s == "Box":
        cp = closest_point_box(pos, rot, entity.shape.length, entity.shape.width, p)
        return _norm(p - cp) - LINE_MIN_DIST
    if s == "Line":
        cp = closest_point_line(pos, rot, entity.shape.length, p)
        return _norm(p - cp) - LINE_MIN_DIST
    raise RuntimeError("Distance not computable for given entity")


def get_distance(a, b) -> Tensor:
    """core.py:1822-1905"""
    sa, sb = _shape_name(a), _shape_name(b)
    if sa == "Sphere" and sb == "Sphere":
        return get_distance_from_point(a, b.state.pos) - b.shape.radius
    if {sa, sb} == {"Box", "Sphere"}:
        box, sph = (a, b) if sa == "Box" else (b, a)
        d = get_distance_from_point(box, sph.state.pos) - sph.shape.radius
        return torch.where(is_overlapping(a, b), torch.full_like(d, -1.0), d)
    if {sa, sb} == {"Line", "Sphere"}:
        line, sph = (a, b) if sa == "Line" else (b, a)
        return get_distance_from_point(line, sph.state.pos) - sph.shape.radius
    if sa == "Line" and sb == "Line":
        p1, p2 = closest_points_seg_seg(a.state.pos, a.state.rot, a.shape.length, b.state.pos, b.state.rot, b.shape.length)
        return _norm(p1 - p2) - LINE_MIN_DIST
    if {sa, sb} == {"Box", "Line"}:
        box, line = (a, b) if sa == "Box" else (b, a)
        pb_, pl = closest_seg_box(box.state.pos, box.state.rot, box.shape.length, box.shape.width, line.state.pos,
                                  line.state.rot, line.shape.length)
        return _norm(pb_ - pl) - LINE_MIN_DIST
    if sa == "Box" and sb == "Box":
        p1, p2 = closest_box_box(a.state.pos, a.state.rot, a.shape.length, a.shape.width, b.state.pos, b.state.rot,
                                 b.shape.length, b.shape.width)
        return _norm(p1 - p2) - LINE_MIN_DIST
    raise RuntimeError("Distance not computable for given entities")


def is_overlapping(a, b) -> Tensor:
    """core.py:1907-1969"""
    sa, sb = _shape_name(a), _shape_name(b)
    if {sa, sb} == {"Box", "Sphere"}:
        box, sph = (a, b) if sa == "Box" else (b, a)
        cp = closest_point_box(box.state.pos, box.state.rot, box.shape.length, box.shape.width, sph.state.pos)
        d_sphere_cp = _norm(sph.state.pos - cp)
        d_sphere_box = _norm(sph.state.pos - box.state.pos)
        d_box_cp = _norm(box.state.pos - cp)
        return (d_sphere_box < d_box_cp) | (d_sphere_cp < sph.shape.radius + LINE_MIN_DIST)
    return get_distance(a, b) < 0
