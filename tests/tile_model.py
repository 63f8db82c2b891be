"""Executable specification of the GPU tile algorithm (host-side model, test only).

The HIP step kernel (egg_fluid_simulation_amd/csrc/eggsim_step.hip) does not run
the reference's sequential loops.  Per collision pass it
  (A) builds every particle's ordered *visit list* from cell adjacency alone,
  (B) cuts the global visit sequence at the collision budget,
  (C) executes the pair solves with a dependency-DAG scheduler in which a pair
      runs as soon as it is the next pending pair of BOTH its particles.
This file states (A)-(C) in plain Python so the rules can be checked against the
oracle bit for bit on the CPU (tests/test_tile_model.py) before/independently of
any GPU run.  It mirrors the kernel's data flow, not the reference's.

Reference semantics reproduced (L = /root/reference/simulation_handler.lua):
  * attempt order of particle i: 3x3 cells, x-offset outer, y-offset inner
    (L:1568-1569), inside a cell the list order, i.e. entries of the previous
    un-cleared pass first, then this pass's, each ascending in particle index
    (L:1509, L:1905-1912);
  * a pair is visited once: skipped when already in `collided` (L:1589), which
    still holds the previous pass's pairs in the first pass of sub-step >= 2;
  * the budget return after ceil(0.05 N^2) visits (L:1657-1658).
"""
import math

EPS = 1e-8


def slot_of(dcx, dcy):
    """index of cell offset (dcx, dcy) in the reference's loop order, or -1"""
    if -1 <= dcx <= 1 and -1 <= dcy <= 1:
        return (dcx + 1) * 3 + (dcy + 1)
    return -1


def build_visit_lists(n, newc, older):
    """(A): own[i] = ordered list of partners particle i visits as `self`.

    newc[i] = (cx, cy) of this pass; older = [(cells, own), ...], oldest first: the cells and the
    (cut) visit lists of every earlier pass whose hash entries and `collided` marks were not cleared
    (one such pass after a sub-step boundary, L:1905-1912; with a single collision pass per sub-step
    nothing is ever cleared inside a step, so every earlier sub-step of the step is in there).
    """
    gens = [c for c, _ in older] + [newc]  # a cell's list holds the oldest generation's entries first
    cells = []
    for g in gens:
        d = {}
        for i in range(n):
            d.setdefault(g[i], []).append(i)
        cells.append(d)

    def in_collided(a, b):
        return any(b in own[a] or a in own[b] for _, own in older)

    def cand_key(i, j):
        """first-occurrence key of j in i's attempt order, or None if j is not met"""
        best = None
        for gi, g in enumerate(gens):
            s = slot_of(g[j][0] - newc[i][0], g[j][1] - newc[i][1])
            if s >= 0 and (best is None or (s, gi, j) < best):
                best = (s, gi, j)
        return best

    own = []
    for i in range(n):
        lst = []
        seen = set()
        cx, cy = newc[i]
        for ox in (-1, 0, 1):
            for oy in (-1, 0, 1):
                cell = (cx + ox, cy + oy)
                for d in cells:
                    for j in d.get(cell, []):
                        if j == i or j in seen:
                            continue
                        seen.add(j)
                        if in_collided(i, j):
                            continue
                        if j < i and cand_key(j, i) is not None:
                            continue  # j's loop met i first
                        lst.append(cand_key(i, j))
        lst.sort()
        own.append([k[2] for k in lst])
    return own


def cut_at_budget(own, w, budget):
    """(B): keep the first m = max(1, ceil(budget)) counted visits; pairs failing the
    mass guard (L:1601) are marked but not counted.  Returns (own, was_cut)."""
    m = max(1, math.ceil(budget))
    counted = 0
    out = [[] for _ in own]
    for i, lst in enumerate(own):
        for j in lst:
            out[i].append(j)
            if w[i] + w[j] < EPS:
                continue
            counted += 1
            if counted >= m and counted >= budget:
                rest = sum(len(l) for l in own[i + 1:]) + (len(lst) - len(out[i]))
                return out, rest > 0
    return out, False


def enforce(ax, ay, bx, by, wa, wb, target, compliance):
    dx = bx - ax
    dy = by - ay
    d = math.sqrt(dx * dx + dy * dy)
    if d < EPS:
        nx, ny = 0.0, 0.0
    else:
        nx, ny = dx / d, dy / d
    cv = d - target
    div = (wa + wb) + compliance
    if div < EPS:
        return 0.0, 0.0, 0.0, 0.0
    corr = -cv / div
    mc = abs(cv)
    if corr < -mc:
        corr = -mc
    if corr > mc:
        corr = mc
    return -nx * corr * wa, -ny * corr * wa, nx * corr * wb, ny * corr * wb


def solve_pair(st, a, b, overlap, compliance):
    x, y, w, r = st["x"], st["y"], st["w"], st["r"]
    if w[a] + w[b] < EPS:
        return
    md = overlap * (r[a] + r[b])
    dx = x[b] - x[a]
    dy = y[b] - y[a]
    if dx * dx + dy * dy <= md * md:
        c = enforce(x[a], y[a], x[b], y[b], w[a], w[b], md, compliance)
        x[a] = x[a] + c[0]
        y[a] = y[a] + c[1]
        x[b] = x[b] + c[2]
        y[b] = y[b] + c[3]


def execute_dag(st, own, overlap, compliance, order="lifo"):
    """(C): run every pair of `own` respecting each particle's pair order only."""
    n = len(own)
    inc = [[] for _ in range(n)]
    for a in range(n):
        for b in own[a]:
            inc[b].append(a)  # ascending in a by construction
    seq = []
    for i in range(n):
        lo = [(c, 0) for c in inc[i] if c < i]
        hi = [(c, 0) for c in inc[i] if c > i]
        seq.append(lo + [(j, 1) for j in own[i]] + hi)  # (partner, i_is_self)
    ptr = [0] * n

    def nxt(i):
        return seq[i][ptr[i]] if ptr[i] < len(seq[i]) else None

    ready = []
    for i in range(n):
        e = nxt(i)
        if e is not None and e[1] == 1:
            o = nxt(e[0])
            if o is not None and o[0] == i and o[1] == 0:
                ready.append((i, e[0]))
    rounds = 0
    done = 0
    while ready:
        rounds += 1
        batch = ready if order != "lifo" else ready[::-1]
        ready = []
        touched = []
        for a, b in batch:
            solve_pair(st, a, b, overlap, compliance)
            ptr[a] += 1
            ptr[b] += 1
            done += 1
            touched += [a, b]
        stamp = set(touched)
        for p in touched:
            e = nxt(p)
            if e is None:
                continue
            q, p_is_self = e
            o = nxt(q)
            if o is None or o[0] != p:
                continue
            if q in stamp and not p_is_self:
                continue  # q's owner pushes it
            ready.append((p, q) if p_is_self else (q, p))
    assert done == sum(len(l) for l in own), "DAG executor stalled"
    return rounds


class TileModel:
    """One particle type, all particles in one tile, fused step as the kernel does it."""

    def __init__(self, cfg, x, y, w, r, atom_of, targets, follow_dist):
        self.cfg = cfg
        self.st = dict(x=list(x), y=list(y), w=list(w), r=list(r))
        self.vx = [0.0] * len(x)
        self.vy = [0.0] * len(x)
        self.atom_of = list(atom_of)
        self.targets = list(targets)
        self.follow_dist = list(follow_dist)
        self.log = []

    def step(self, delta, S, C):
        cfg, st = self.cfg, self.st
        n = len(st["x"])
        sub = max(delta / S, EPS)
        damping = 1 - min(max(cfg["damping"], 0), 1)

        def compl(s):
            return (1 - min(max(s, 0), 1)) / (sub * sub)

        fc, cc = compl(cfg["follow_strength"]), compl(cfg["collision_strength"])
        budget = 0.05 * (n * n)
        cell = max(1, cfg["max_radius"] * max(cfg["collision_overlap_factor"],
                                              cfg["cohesion_interaction_distance_factor"]))
        x, y, w = st["x"], st["y"], st["w"]
        older = []  # (cells, visit lists) of the passes not cleared since
        self.log = []
        for s in range(S):
            px, py = list(x), list(y)
            for i in range(n):
                self.vx[i] = self.vx[i] * damping
                self.vy[i] = self.vy[i] * damping
                x[i] = x[i] + sub * self.vx[i]
                y[i] = y[i] + sub * self.vy[i]
            for i in range(n):
                tx, ty = self.targets[self.atom_of[i]]
                dx, dy = tx - x[i], ty - y[i]
                d = math.sqrt(dx * dx + dy * dy)
                td = self.follow_dist[self.atom_of[i]]
                if w[i] > EPS and d > td:
                    nx, ny = (0.0, 0.0) if d < EPS else (dx / d, dy / d)
                    lam = (d - td) / (w[i] + fc)
                    x[i] = x[i] + nx * lam * w[i]
                    y[i] = y[i] + ny * lam * w[i]
            for c in range(C):
                newc = [(math.floor(x[i] / cell), math.floor(y[i] / cell)) for i in range(n)]
                own = build_visit_lists(n, newc, older)
                own, was_cut = cut_at_budget(own, w, budget)
                rounds = execute_dag(st, own, cfg["collision_overlap_factor"], cc)
                self.log.append((sum(len(l) for l in own), int(was_cut), rounds))
                if c + 1 < C:
                    older = []  # cleared (L:1905-1912)
                else:
                    older.append((newc, own))  # survives into the next sub-step (Q3), on top of what it saw
            for i in range(n):
                self.vx[i] = (x[i] - px[i]) / sub
                self.vy[i] = (y[i] - py[i]) / sub
