// eggsim_host_tiling.hip -- retile(): claims, islands, tiles, launch classes and the packed pipeline's groups.  See eggsim_host.h.
#include "eggsim_host.h"

namespace egghost {


// ------------------------------------------------------------------ tiling


// Groups atoms into tiles.  Atoms whose margin-padded cell boxes come within one cell of
// each other may interact during the step and must share a tile ("islands": connected
// components of that relation); independent islands may additionally be packed into one
// tile to fill a wave.  Each atom's claim box is its padded box: the kernel verifies that
// no particle leaves it, which proves that particles of different tiles never occupy
// adjacent cells (claims of different islands are separated by >= 1 empty cell).
// developer aid (EGGSIM_HOST_PROFILE=1): wall time of retile()'s sections, printed when the handle is destroyed
double g_retile_ms[8];
struct RetileLap {
    std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
    void operator()(int k) {
        const auto n = std::chrono::steady_clock::now();
        g_retile_ms[k] += std::chrono::duration<double, std::milli>(n - t).count();
        t = n;
    }
};

int retile(egg_handle *h, int which) {
    System &s = h->sys[which];
    RetileLap lap;
    int rc = upload_atoms(h, which);
    if (rc != EGG_OK) return rc;
    const size_t na = s.atoms.size();
    const double cell = cell_size_of(s.cfg);
    s.extra_margin.resize(h->batches.size(), 0);
    s.tile_atom_begin.assign(1, 0);
    s.tile_atoms.clear();
    s.classes.clear();
    h->stats.n_tiles[which] = 0;
    h->stats.max_tile_particles[which] = 0;
    if (na == 0) {
        s.tiling_dirty = false;
        s.tiled_cell_size = cell;
        return EGG_OK;
    }
    if (!s.aabb_valid && s.aabb_on_device && s.tiled_cell_size == cell) {
        rc = fetch_end_aabb(h, s);
        if (rc != EGG_OK) return rc;
    }
    s.tiled_cell_size = cell;
    if (!s.aabb_valid) {
        hipLaunchKernelGGL(egg_atom_bounds_kernel, dim3((unsigned)na), dim3(EGG_WAVE), 0, s.stream, s.x[s.cur].p,
                           s.y[s.cur].p, s.d_atom_offset.p, s.d_atom_count.p, (int)na, cell, d_aabb(s));
        HIP_TRY(h, hipGetLastError());
        s.aabb.resize(na);
        HIP_TRY(h, hipMemcpyAsync(s.aabb.data(), d_aabb(s), na * sizeof(Box), hipMemcpyDeviceToHost, s.stream));
        HIP_TRY(h, hipStreamSynchronize(s.stream));
        s.aabb_valid = true;
        s.disp_valid = false;
        h->stats.kernel_launches++;
    }
    for (const Box &b : s.aabb)
        if (b.lo_x < -2000000000 || b.hi_x > 2000000000 || b.lo_y < -2000000000 || b.hi_y > 2000000000)
            return fail(h, EGG_ERR_UNSUPPORTED, "particle coordinates are not finite or exceed +-2e9 cells");

    lap(0);
    std::vector<Box> claim(na);
    std::vector<int> comp(na);
    const bool single = s.single_tile || h->opt_force_single;
    if (single) {
        Box u{std::numeric_limits<int32_t>::max(), std::numeric_limits<int32_t>::max(),
              std::numeric_limits<int32_t>::min(), std::numeric_limits<int32_t>::min()};
        for (const Box &b : s.aabb) {
            u.lo_x = std::min(u.lo_x, b.lo_x);
            u.lo_y = std::min(u.lo_y, b.lo_y);
            u.hi_x = std::max(u.hi_x, b.hi_x);
            u.hi_y = std::max(u.hi_y, b.hi_y);
        }
        int64_t ext = std::max<int64_t>((int64_t)u.hi_x - u.lo_x, (int64_t)u.hi_y - u.lo_y);
        if (ext > 60000) return fail(h, EGG_ERR_UNSUPPORTED, "single-tile extent of %lld cells is too large", (long long)ext);
        int m = (int)std::min<int64_t>(std::max(s.margin, 64), (60000 - ext) / 2);
        if (m < 1) m = 1;
        u = Box{u.lo_x - m, u.lo_y - m, u.hi_x + m, u.hi_y + m};
        for (size_t k = 0; k < na; ++k) {
            claim[k] = u;
            comp[k] = 0;
        }
    } else {
        // Claim = occupied cells + margin, swept along the displacement the follow constraint is about
        // to cause: per sub-step a particle farther than the slack from its target moves towards it by
        // (d - slack) * w / (w + follow compliance) (L:1461-1468), so a blob far from its target moves
        // many cells in one step in a known direction.  Sweeping the claim that way (with 50 % head room)
        // keeps fast blobs inside their claims without widening every neighbour's margin.
        const int m = s.margin;
        const double w_max = 1.0 / std::max(s.cfg.min_mass, 1e-300);  // lightest particle moves most
        const double pull = w_max / (w_max + s.step_follow_compliance);
        // inertia: x += dt * v carries on what the particles did in the previous step (times the
        // damping); the step kernel reports every atom's largest particle travel per direction
        const bool have_motion = s.disp_valid && s.disp.size() == 4 * na;
        s.swept = false;
        s.padded = s.margin > h->opt_margin;
        s.since_tiling = 0;
        for (size_t k = 0; k < na; ++k) {
            const Box &b = s.aabb[k];
            const Batch &B = h->batches[(size_t)s.atoms[k].batch];
            const double cx = 0.5 * ((double)b.lo_x + b.hi_x + 1.0) * cell, cy = 0.5 * ((double)b.lo_y + b.hi_y + 1.0) * cell;
            const double dx = B.target_x - cx, dy = B.target_y - cy;
            const double dist = std::sqrt(dx * dx + dy * dy);
            const double slack = 2 * std::sqrt(which == EGG_WHITE ? B.white_radius : B.yolk_radius);
            // Predict the step like the solver runs it: per sub-step the displacement is
            // damping * (previous sub-step's displacement) [pre-solve, L:1411-1418] plus the follow pull
            // (remaining distance beyond the slack) * w / (w + compliance) [L:1461-1468], along the
            // direction to the target.  Collisions only redistribute that inside the blob.
            const double ux = dist > 0 ? dx / dist : 0, uy = dist > 0 ? dy / dist : 0;
            const int mk = m + s.extra_margin[(size_t)s.atoms[k].batch];
            int side[4];
            for (int q = 0; q < 4; ++q) {
                const double dir = (q == 0) ? ux : (q == 1) ? -ux : (q == 2) ? uy : -uy;  // +x, -x, +y, -y
                double d = have_motion ? s.disp[4 * k + q] / 16.0 : 0.0;  // last sub-step, towards this side
                double remaining = (dist > slack && std::isfinite(dist)) ? dist - slack : 0.0;
                double travel = 0;
                for (int sub = 0; sub < s.step_substeps; ++sub) {
                    const double p = remaining * pull;
                    remaining -= p;
                    d = s.step_damping * d + std::max(0.0, dir) * p;
                    travel += d;
                }
                side[q] = std::max(mk, (int)std::min(4096.0, std::ceil(1.25 * travel / cell)));
                s.swept |= side[q] > mk;
                s.padded |= side[q] > m;
            }
            claim[k] = Box{b.lo_x - side[1], b.lo_y - side[3], b.hi_x + side[0], b.hi_y + side[2]};
        }
        lap(1);
        // union-find over atoms
        std::vector<int> parent(na);
        std::iota(parent.begin(), parent.end(), 0);
        auto find = [&](int v) {
            while (parent[v] != v) {
                parent[v] = parent[parent[v]];
                v = parent[v];
            }
            return v;
        };
        auto touch = [&](int a, int b) {  // not separated by a full empty cell column or row
            const Box &A = claim[(size_t)a], &B = claim[(size_t)b];
            if ((int64_t)B.lo_x > (int64_t)A.hi_x + 1 || (int64_t)A.lo_x > (int64_t)B.hi_x + 1) return;
            if ((int64_t)B.lo_y > (int64_t)A.hi_y + 1 || (int64_t)A.lo_y > (int64_t)B.hi_y + 1) return;
            int ra = find(a), rb = find(b);
            if (ra != rb) parent[std::max(ra, rb)] = std::min(ra, rb);
        };
        // Candidate pairs from a grid of square buckets as wide as an ordinary claim (+ the empty cell): two
        // such claims that touch have their low corners in the same or in adjacent buckets.  The few claims wider
        // than a bucket (a blob flying towards a far target) are tested against everything.
        int64_t ext_sum = 0, ext_max = 0, min_x = std::numeric_limits<int64_t>::max(), min_y = min_x, max_x = -min_x, max_y = -min_x;
        auto ext_of = [](const Box &c) { return std::max<int64_t>((int64_t)c.hi_x - c.lo_x, (int64_t)c.hi_y - c.lo_y) + 2; };
        for (const Box &c : claim) {
            const int64_t e = ext_of(c);
            ext_sum += e;
            ext_max = std::max(ext_max, e);
            min_x = std::min<int64_t>(min_x, c.lo_x);
            min_y = std::min<int64_t>(min_y, c.lo_y);
            max_x = std::max<int64_t>(max_x, c.lo_x);
            max_y = std::max<int64_t>(max_y, c.lo_y);
        }
        const int64_t ext_mean = ext_sum / (int64_t)na + 1;
        const int64_t bucket = ext_max <= 4 * ext_mean ? ext_max : 2 * ext_mean;
        const int64_t nbx = (max_x - min_x) / bucket + 1, nby = (max_y - min_y) / bucket + 1;
        if (nbx * nby <= 8 * (int64_t)na + 1024) {
            std::vector<int32_t> cell_of(na), start((size_t)(nbx * nby) + 1, 0), member(na), wide;
            for (size_t k = 0; k < na; ++k) {
                const Box &c = claim[k];
                if (ext_of(c) > bucket) {
                    cell_of[k] = -1;
                    wide.push_back((int32_t)k);
                    continue;
                }
                cell_of[k] = (int32_t)(((int64_t)c.lo_x - min_x) / bucket * nby + ((int64_t)c.lo_y - min_y) / bucket);
                start[(size_t)cell_of[k] + 1]++;
            }
            for (size_t c = 0; c < (size_t)(nbx * nby); ++c) start[c + 1] += start[c];
            {
                std::vector<int32_t> fill(start.begin(), start.end() - 1);
                for (size_t k = 0; k < na; ++k)
                    if (cell_of[k] >= 0) member[(size_t)fill[(size_t)cell_of[k]]++] = (int32_t)k;
            }
            for (size_t k = 0; k < na; ++k) {
                const int32_t c = cell_of[k];
                if (c < 0) continue;
                const int64_t bx = c / nby, by = c % nby;
                // own bucket (later members) and the four forward neighbours: every pair once
                for (int32_t m = start[(size_t)c]; m < start[(size_t)c + 1]; ++m)
                    if (member[(size_t)m] > (int32_t)k) touch((int)k, member[(size_t)m]);
                const int64_t nb[4][2] = {{bx, by + 1}, {bx + 1, by - 1}, {bx + 1, by}, {bx + 1, by + 1}};
                for (const auto &q : nb) {
                    if (q[0] >= nbx || q[1] < 0 || q[1] >= nby) continue;
                    const size_t c2 = (size_t)(q[0] * nby + q[1]);
                    for (int32_t m = start[c2]; m < start[c2 + 1]; ++m) touch((int)k, member[(size_t)m]);
                }
            }
            for (int32_t wk : wide)
                for (size_t k = 0; k < na; ++k)
                    if ((int32_t)k != wk) touch(wk, (int)k);
        } else {
            // atoms scattered over far more buckets than there are atoms: sweep over lo_x
            std::vector<int> order(na);
            std::iota(order.begin(), order.end(), 0);
            std::sort(order.begin(), order.end(), [&](int a, int b) { return claim[a].lo_x < claim[b].lo_x; });
            for (size_t i = 0; i < na; ++i)
                for (size_t j = i + 1; j < na; ++j) {
                    if ((int64_t)claim[order[j]].lo_x > (int64_t)claim[order[i]].hi_x + 1) break;
                    touch(order[i], order[j]);
                }
        }
        for (size_t k = 0; k < na; ++k) comp[k] = find((int)k);
    }

    lap(2);
    // islands: atoms grouped by component (in the order of each component's first atom), ascending atom index
    // inside (keeps particle order); flat arrays, this runs every step while blobs move
    std::vector<int32_t> island_of(na, -1), isl_begin(1, 0), isl_atoms(na);
    {
        int32_t n_isl = 0;
        std::vector<int32_t> size;
        for (size_t k = 0; k < na; ++k) {
            const int r = comp[k];
            if (island_of[(size_t)r] < 0) {
                island_of[(size_t)r] = n_isl++;
                size.push_back(0);
            }
            size[(size_t)island_of[(size_t)r]]++;
        }
        isl_begin.resize((size_t)n_isl + 1);
        for (int32_t i = 0; i < n_isl; ++i) isl_begin[(size_t)i + 1] = isl_begin[(size_t)i] + size[(size_t)i];
        std::vector<int32_t> fill(isl_begin.begin(), isl_begin.end() - 1);
        for (size_t k = 0; k < na; ++k) isl_atoms[(size_t)fill[(size_t)island_of[(size_t)comp[k]]]++] = (int32_t)k;
    }
    struct TileTmp {  // atoms: isl_atoms[a_begin, a_end) -- islands that share a tile are neighbours in that array
        int32_t a_begin = 0, a_end = 0;
        int64_t particles = 0;
        Box box{std::numeric_limits<int32_t>::max(), std::numeric_limits<int32_t>::max(),
                std::numeric_limits<int32_t>::min(), std::numeric_limits<int32_t>::min()};
    };
    auto grow = [](Box a, const Box &b) {
        a.lo_x = std::min(a.lo_x, b.lo_x);
        a.lo_y = std::min(a.lo_y, b.lo_y);
        a.hi_x = std::max(a.hi_x, b.hi_x);
        a.hi_y = std::max(a.hi_y, b.hi_y);
        return a;
    };
    auto extent = [](const Box &b) { return std::max<int64_t>((int64_t)b.hi_x - b.lo_x, (int64_t)b.hi_y - b.lo_y); };
    std::vector<TileTmp> tiles;
    tiles.reserve(isl_begin.size());
    const int64_t target = single ? 0 : h->opt_tile_target;
    for (size_t i = 0; i + 1 < isl_begin.size(); ++i) {
        TileTmp one;
        one.a_begin = isl_begin[i];
        one.a_end = isl_begin[i + 1];
        for (int32_t k = one.a_begin; k < one.a_end; ++k) {
            const int32_t a = isl_atoms[(size_t)k];
            one.particles += s.atoms[(size_t)a].count;
            one.box = grow(one.box, claim[(size_t)a]);
        }
        // independent islands share a tile only while the joint claim box stays small enough for the dense cell
        // grid: a tile spanning the scene falls back to the hash table, which made such tiles (and with them
        // the whole launch) 7x slower
        auto grid_cells = [](const Box &b) { return ((int64_t)b.hi_x - b.lo_x + 4) * ((int64_t)b.hi_y - b.lo_y + 4); };
        if (target > 0 && !tiles.empty() && tiles.back().particles + one.particles <= target &&
            extent(grow(tiles.back().box, one.box)) <= 60000 && grid_cells(grow(tiles.back().box, one.box)) <= 2048) {
            // independent islands may share a tile (fills the wave's lanes in the pair executor)
            TileTmp &tt = tiles.back();
            tt.a_end = one.a_end;
            tt.particles += one.particles;
            tt.box = grow(tt.box, one.box);
        } else {
            tiles.push_back(one);
        }
    }
    for (auto &tt : tiles) {
        std::sort(isl_atoms.begin() + tt.a_begin, isl_atoms.begin() + tt.a_end);
        if (tt.particles > kMaxTileParticles)
            return fail(h, EGG_ERR_UNSUPPORTED,
                        "%lld particles of one type interact in one island; the LDS tile kernel handles at most %d",
                        (long long)tt.particles, kMaxTileParticles);
        // (tt.box is the union of the tile's claims)
        const int64_t ext_x = (int64_t)tt.box.hi_x - tt.box.lo_x, ext_y = (int64_t)tt.box.hi_y - tt.box.lo_y;
        if (ext_x > 65000 || ext_y > 65000)
            return fail(h, EGG_ERR_UNSUPPORTED, "a tile spans %lld x %lld cells; limit is 65000",
                        (long long)ext_x, (long long)ext_y);
    }
    // big tiles first, and similar sizes adjacent so that they share a launch class
    std::stable_sort(tiles.begin(), tiles.end(),
                     [](const TileTmp &a, const TileTmp &b) { return a.particles > b.particles; });

    s.tile_atoms.reserve(na);
    s.tile_atom_begin.reserve(tiles.size() + 1);
    for (auto &tt : tiles) {
        s.tile_atoms.insert(s.tile_atoms.end(), isl_atoms.begin() + tt.a_begin, isl_atoms.begin() + tt.a_end);
        s.tile_atom_begin.push_back((int32_t)s.tile_atoms.size());
    }
    lap(3);
    // launch classes: consecutive tiles whose particle count is within 2x
    size_t t0 = 0;
    size_t scratch_bytes = 0;
    while (t0 < tiles.size()) {
        size_t t1 = t0;
        int64_t nmax = tiles[t0].particles;
        int amax = 0;
        int64_t max_cells = 0;
        while (t1 < tiles.size() && tiles[t1].particles * 2 > nmax) {
            amax = std::max(amax, (int)(tiles[t1].a_end - tiles[t1].a_begin));
            // dense grid the kernel lays over the tile's claim box (see eggsim_step.hip, load tile)
            const Box &bx = tiles[t1].box;
            max_cells = std::max(max_cells, ((int64_t)bx.hi_x - bx.lo_x + 4) * ((int64_t)bx.hi_y - bx.lo_y + 4));
            ++t1;
        }
        LaunchClass lc;
        lc.first_tile = (int)t0;
        lc.n_tiles = (int)(t1 - t0);
        lc.nmax = (int)((nmax + 7) & ~7ll);
        lc.amax = amax;
        if (max_cells <= std::max<int64_t>(2048, 8 * (int64_t)lc.nmax) && max_cells <= 16384) {
            lc.use_grid = 1;
            lc.ccap = (int)((max_cells + 63) & ~63ll);
        } else {
            lc.use_grid = 0;
            int ht = 64;
            while (ht < lc.nmax + lc.nmax / 2) ht *= 2;
            lc.ccap = ht;
        }
        size_t lcap = std::max<size_t>({(size_t)lc.nmax, (size_t)(s.list_factor * lc.nmax), s.list_min});
        lcap = std::min<size_t>(lcap, kMaxListEntries);
        lcap = (lcap + 7) & ~(size_t)7;
        lc.lcap = (int)lcap;
        int threads = egg_step_threads(lc.nmax, 1);
        lc.lds = egg_step_lds_bytes(lc.nmax, lc.amax, lc.ccap, lc.use_grid, lc.lcap, single ? 1 : 0, 0, threads, 0, s.gens);
        bool want_global_state = h->opt_force_global_state != 0;
        if (!want_global_state && (lc.lds > h->lds_limit || lc.lds > 64 * 1024 || threads > 256)) {
            // dense or large tiles: particle state stays in LDS, the visit lists go to global memory
            lc.global_lists = 1;
            lcap = std::max<size_t>({lcap, (size_t)(32.0 * lc.nmax), s.list_min});
            lcap = std::min<size_t>((lcap + 7) & ~(size_t)7, kMaxGlobalListEntries);
            lc.lcap = (int)lcap;
            lc.lds = egg_step_lds_bytes(lc.nmax, lc.amax, lc.ccap, lc.use_grid, lc.lcap, single ? 1 : 0, 1, threads, 0, s.gens);
            if (lc.lds > h->lds_limit) want_global_state = true;
        }
        if (want_global_state) {
            // the particle state itself does not fit (a very large island): everything goes to the tile's
            // scratch slice, laid out like the LDS image followed by the lists
            lc.global_lists = lc.global_state = 1;
            lcap = std::max<size_t>({lcap, (size_t)(32.0 * lc.nmax), s.list_min});
            lc.lcap = (int)std::min<size_t>((lcap + 7) & ~(size_t)7, kMaxGlobalListEntries);
            const size_t state = egg_step_lds_bytes(lc.nmax, lc.amax, lc.ccap, lc.use_grid, lc.lcap, single ? 1 : 0, 1, threads, 0, s.gens);
            lc.scratch_stride = ((state + 255) & ~(size_t)255) + egg_step_scratch_bytes(lc.lcap, single ? 1 : 0, s.gens) + 256;
            lc.lds = 0;
        } else if (lc.global_lists) {
            lc.scratch_stride = (egg_step_scratch_bytes(lc.lcap, single ? 1 : 0, s.gens) + 255) & ~(size_t)255;
        }
        // A tile that has a CU (almost) to itself leaves most of the CU's issue slots idle: give it three
        // lanes per particle, which the kernel uses to build the visit lists column-wise.  With more
        // tiles than that, one lane per particle keeps the most tiles resident.
        if (!lc.global_lists && !lc.global_state && s.gens <= 2) {
            int spread = h->opt_spread;
            // (two such workgroups do not fit one CU's register file, so: at most one tile per CU)
            // -- and only while the white tiles do not fill the chip anyway (then the types share one launch)
            const bool chip_shared = which == 1 && h->stats.n_tiles[0] > h->prop.multiProcessorCount;
            if (spread <= 0) spread = (lc.n_tiles <= h->prop.multiProcessorCount && !chip_shared) ? 3 : 1;
            const int wide_threads = egg_step_threads(lc.nmax, spread);
            if (spread > 1 && wide_threads <= 512 && wide_threads >= 3 * lc.nmax) {
                // (a wide tile keeps velocities and sub-step start positions in LDS: the size depends on the thread count)
                const size_t wide_lds = egg_step_lds_bytes(lc.nmax, lc.amax, lc.ccap, lc.use_grid, lc.lcap, single ? 1 : 0, 0,
                                                           wide_threads, 0, s.gens);
                if (wide_lds <= h->lds_limit && wide_lds <= 64 * 1024) {
                    threads = wide_threads;
                    lc.wide = 1;
                    lc.lds = wide_lds;
                }
            }
            // room to spare (few tiles per CU): cache the position-independent terms of every pair
            const size_t with_cache = egg_step_lds_bytes(lc.nmax, lc.amax, lc.ccap, lc.use_grid, lc.lcap, single ? 1 : 0, 0,
                                                         threads, 1, s.gens);
            const int tiles_per_cu = (lc.n_tiles + h->prop.multiProcessorCount - 1) / h->prop.multiProcessorCount;
            if (with_cache <= h->lds_limit && with_cache <= 64 * 1024 && (size_t)tiles_per_cu * with_cache <= 96 * 1024) {
                lc.pair_cache = 1;
                lc.lds = with_cache;
            }
        }
        lc.threads = threads;
        if (lc.global_lists) {
            scratch_bytes = (scratch_bytes + 255) & ~(size_t)255;
            lc.scratch_offset = scratch_bytes;
            scratch_bytes += (size_t)lc.n_tiles * lc.scratch_stride;
        }
        s.classes.push_back(lc);
        t0 = t1;
    }

    lap(4);
    // ---- packed pipeline (eggsim_packed.hip): which classes take it, their packed ranges and groups
    s.pk.clear();
    s.pk_meta_host.clear();
    s.pk_n = s.pk_tiles = s.pk_groups = 0;
    s.pk_entries = 0;
    s.pk_sort_words = 0;
    s.pk_chunk_words = 0;
    {
        // automatic: scenes large enough that the chip is full of tiles whatever the kernel (the fused kernel's
        // latency per step is lower while every tile has a CU almost to itself); judged on the white tiles so
        // that both types of a scene take the same path
        // Measured crossover on MI355X (ms per step, fused vs packed): separate 157-particle blobs 1024: 0.58 / 0.68,
        // 1536: 0.78 / 0.73, 2048: 1.06 / 0.79, 3072: 1.38 / 0.86; dense 628-particle islands 192: 1.65 / 2.79, 256: 1.71 /
        // 2.82, 320: 2.99 / 2.85, 512: 3.20 / 2.92, 768: 4.92 / 3.09 -- the packed pipeline has a latency floor per
        // collision pass; the fused kernel's time grows with the tiles per CU, and a dense island fills a CU: from the
        // first CU that gets two, its step takes twice as long.
        if (which == 0) {
            const int64_t cus = std::max(1, h->prop.multiProcessorCount);
            const bool dense = !tiles.empty() && tiles.front().particles > 256;
            h->packed_auto = dense ? (int64_t)tiles.size() > cus : (int64_t)tiles.size() >= 6 * cus;
        }
        const bool want = h->opt_packed > 0 || (h->opt_packed < 0 && h->packed_auto);
        const bool allowed = want && !single && s.gens <= 2 && s.pk_allowed;
        for (size_t ci = 0; allowed && ci < s.classes.size(); ++ci) {
            LaunchClass &lc = s.classes[ci];
            if (lc.nmax > 8192) continue;  // (a class whose FUSED kernel would keep its state in global memory may still fit: the list kernel's LDS need is checked below)
            PackedClass pc;
            pc.cls = (int)ci;
            pc.n_tiles = lc.n_tiles;
            pc.lcap = lc.lcap;
            pc.scap = lc.lcap;
            pc.threads_lists = pc.threads_lists_stale = egg_step_threads(lc.nmax, 1);
            // the counting pass keeps up to stage_cap partners per particle in LDS as long as that does not cost a
            // resident tile per CU (residency: LDS and the 32-wave limit); sized by the stale pass, which holds two cell generations
            auto tiles_per_cu = [&](int stage, int gens, int threads) {
                const size_t lds = egg_pk_lists_lds_bytes(lc.nmax, lc.amax, lc.ccap, lc.use_grid, stage, gens);
                if (lds > h->lds_limit) return (size_t)0;  // (what a workgroup may have is a little less than the CU's 160 KiB)
                return std::min<size_t>(kLdsMax / std::max<size_t>(lds, 1), (size_t)2048 / (size_t)threads);
            };
            for (pc.stage_cap = 16; pc.stage_cap > 0; pc.stage_cap -= 2)
                if (tiles_per_cu(pc.stage_cap, 2, pc.threads_lists) == tiles_per_cu(0, 2, pc.threads_lists)) break;
            pc.lds_lists = egg_pk_lists_lds_bytes(lc.nmax, lc.amax, lc.ccap, lc.use_grid, pc.stage_cap, 1);
            pc.lds_lists_stale = egg_pk_lists_lds_bytes(lc.nmax, lc.amax, lc.ccap, lc.use_grid, pc.stage_cap, 2);
            if (pc.lds_lists_stale > h->lds_limit) continue;
            // Fewer threads than particles when that saves a whole round of workgroups: 1,024 dense tiles of 628 particles at
            // 640 threads are three to a CU (32 waves), i.e. two rounds on 256 CUs; at 512 threads four fit -- if the LDS
            // allows -- and one round of workgroups 1.5 x as long wins.  (Cost model: rounds x (1 + particles per thread) / 2.)
            for (int kind = 0; kind < 2; ++kind) {
                const int full = egg_step_threads(lc.nmax, 1);
                int best = full;
                double best_cost = 1e30;
                for (int threads = full; threads >= 256; threads -= 64) {
                    const size_t per_cu = tiles_per_cu(pc.stage_cap, kind + 1, threads);
                    if (!per_cu) break;
                    const double rounds = std::ceil((double)lc.n_tiles / ((double)per_cu * (double)std::max(1, h->prop.multiProcessorCount)));
                    const double cost = rounds * 0.5 * (1.0 + std::ceil((double)lc.nmax / threads));
                    if (cost < best_cost - 1e-9) {
                        best_cost = cost;
                        best = threads;
                    }
                }
                (kind ? pc.threads_lists_stale : pc.threads_lists) = best;
            }
            const size_t meta_mark = s.pk_meta_host.size();
            pc.p_begin = s.pk_n;
            pc.entry_base = s.pk_entries;
            pc.sort_base = s.pk_sort_words;
            pc.tile_base = s.pk_tiles;
            pc.group_base = s.pk_groups;
            // tile records: first packed particle, particles, first tile atom, atoms, cell origin, grid extent
            pc.meta_tile_geo = (s.pk_meta_host.size() + 3) & ~(size_t)3;  // 16-byte aligned records
            s.pk_meta_host.resize(pc.meta_tile_geo);
            int pn = s.pk_n;
            std::vector<int> tile_first_particle((size_t)lc.n_tiles + 1);
            for (int t = 0; t < lc.n_tiles; ++t) {
                const size_t gt = (size_t)lc.first_tile + t;
                const int a0 = s.tile_atom_begin[gt], a1 = s.tile_atom_begin[gt + 1];
                int32_t lx = std::numeric_limits<int32_t>::max(), ly = lx, hx = std::numeric_limits<int32_t>::min(), hy = hx;
                for (int k = a0; k < a1; ++k) {
                    const Box &c = claim[(size_t)s.tile_atoms[(size_t)k]];
                    lx = std::min(lx, c.lo_x);
                    ly = std::min(ly, c.lo_y);
                    hx = std::max(hx, c.hi_x);
                    hy = std::max(hy, c.hi_y);
                }
                tile_first_particle[(size_t)t] = pn;
                const int32_t rec[8] = {pn, (int32_t)tiles[gt].particles, a0, a1 - a0, lx - 2, ly - 2,
                                        (int32_t)std::min<int64_t>((int64_t)hx - lx + 4, 65535), (int32_t)std::min<int64_t>((int64_t)hy - ly + 4, 65535)};
                s.pk_meta_host.insert(s.pk_meta_host.end(), rec, rec + 8);
                pn += (int)tiles[gt].particles;
            }
            tile_first_particle[(size_t)lc.n_tiles] = pn;
            pc.p_end = pn;
            // groups: consecutive tiles while one wave's LDS holds their positions (tiles are sorted by size, largest first)
            pc.meta_grp_geo = s.pk_meta_host.size();
            // Particles per executor wave.  1280 fills the lanes best on a full chip (eight 157-particle blobs: ~80 pairs
            // per level).  While the class's groups are fewer than the chip's SIMDs the executor and the level walk are
            // latency bound -- a level costs the same ~0.3 us whether its chunk holds 10 pairs or 64, two chunks of one
            // level cost twice that -- so smaller groups (more waves, one chunk per level) are faster: measured ms per
            // step with 320 / 640 / 1280 particles per group, 2048 blobs: 0.77 / 0.82 / 1.01, 4096: 1.03 / 0.94 / 0.99,
            // 8192: 1.59 / 1.32 / 1.32.  Dense islands (four coincident blobs, 628 particles) keep 1280: 3.32 vs 3.54.
            int64_t gp_auto = 1280;
            if (tiles[(size_t)lc.first_tile].particles <= 256) {
                int64_t class_particles = 0;
                for (int t = 0; t < lc.n_tiles; ++t) class_particles += tiles[(size_t)lc.first_tile + t].particles;
                const int64_t simds = 4 * (int64_t)std::max(1, h->prop.multiProcessorCount);
                gp_auto = std::min<int64_t>(1280, std::max<int64_t>(320, (class_particles / simds + 159) / 160 * 160));
            }
            // (Four dense islands to an executor -- EGG_OPT_GROUP_PARTICLES = 2560, eight waves per fused group -- was 5 % faster
            // for deep, narrow dependency graphs while two groups filled a CU's LDS; since the level array is sized per step it
            // is 4 % slower there too -- 3.74 vs 3.59 ms per step after 600 steps of config 3 -- and is left to the option.)
            const int64_t gp_max = std::max<int64_t>(h->opt_group_particles > 0 ? h->opt_group_particles : gp_auto,
                                                     tiles[(size_t)lc.first_tile].particles);
            int max_tiles_in_group = 0;
            for (int t = 0; t < lc.n_tiles;) {
                int64_t in_group = 0;
                int t_end = t;
                while (t_end < lc.n_tiles && t_end - t < 64) {
                    const int64_t np = tiles[(size_t)lc.first_tile + t_end].particles;
                    if (t_end > t && (in_group + np > gp_max || in_group + np > 32767)) break;
                    in_group += np;
                    ++t_end;
                }
                const int32_t rec[4] = {t, t_end, tile_first_particle[(size_t)t], (int32_t)in_group};
                s.pk_meta_host.insert(s.pk_meta_host.end(), rec, rec + 4);
                pc.n_groups++;
                pc.max_group_particles = std::max<int>(pc.max_group_particles, (int)in_group);
                max_tiles_in_group = std::max(max_tiles_in_group, t_end - t);
                t = t_end;
            }
            // The level walk.  Dense islands (more than 256 particles: thousands of pairs per pass in hundreds of levels)
            // on a chip that is not full -- groups no more than SIMDs -- wait for the longest dependency chain of one
            // tile: the out-of-order walk levels whatever runs are ready (a 628-particle island's 6,500 pairs: ~165 turns
            // instead of ~540) and sorts in the same launch.  Its fixed costs (ranking pass, barriers, ~40 us) lose against
            // the in-order walk on sparse 157-particle tiles at every scene size (2,048 blobs: 0.41 vs 0.27 ms per step,
            // 16,384: 1.48 vs 0.54; profiles/r03_walk_sweep.txt), and on a full chip its extra instructions do.
            const int simds = 4 * std::max(1, h->prop.multiProcessorCount);
            const bool dense_tiles = tiles[(size_t)lc.first_tile].particles > 256;
            pc.levels_ooo = h->lds_lane_ordered && h->opt_level_walk != 1 && (h->opt_level_walk == 2 || (pc.n_groups <= simds && dense_tiles));
            if (pc.levels_ooo) {
                // the levels of a tile's stream live in LDS: room for 24 pairs per particle (a dense island's first
                // steps: ~20), more after a launch that needed more, never more than the stream itself can hold
                // (once steps have run: half as much again as the longest list any tile had in the last step -- a fused pass keeps
                // its LDS small enough for four islands per CU that way; a launch that needs more says so and is re-run)
                const size_t guess = s.pk_seen_list ? (size_t)(s.pk_seen_list * 3 / 2 + 512) : (size_t)24 * lc.nmax;
                pc.lev_lds_cap = (int)std::min<size_t>((size_t)pc.scap, std::max<size_t>(guess, s.pk_lev_lds_min));
                pc.lev_lds_cap = (pc.lev_lds_cap + 7) & ~7;
                // (more groups than CUs: two of them must fit a CU's LDS, or the launch takes two rounds.  The head room above
                // must not cost that -- a dense island's first steps have streams twice as long as later ones, and half as
                // much again on top of those is 98 KB per group)
                if (pc.n_groups > std::max(1, h->prop.multiProcessorCount)) {
                    const size_t budget = (kLdsMax - 4096) / 2;
                    const size_t fixed = egg_pk_levels_ooo_lds_bytes(s.pk_lev_cap, pc.max_group_particles, max_tiles_in_group, 0);
                    if (budget > fixed) {
                        const size_t fit = ((budget - fixed) / ((size_t)max_tiles_in_group * 2)) & ~(size_t)7;
                        const size_t need = std::max<size_t>((size_t)(s.pk_seen_list + s.pk_seen_list / 16 + 64), s.pk_lev_lds_min);
                        if (fit >= need && (size_t)pc.lev_lds_cap > fit) pc.lev_lds_cap = (int)fit;
                    }
                }
                // a wave per tile, at least four per group (eight for two tiles were 10 % slower); two waves per tile for up to four
                pc.levels_threads = 64 * std::min(16, std::max(4, max_tiles_in_group <= 4 ? 2 * max_tiles_in_group : max_tiles_in_group));
                pc.lds_levels = egg_pk_levels_ooo_lds_bytes(s.pk_lev_cap, pc.max_group_particles, max_tiles_in_group, pc.lev_lds_cap);
                if (pc.lds_levels > h->lds_limit) pc.levels_ooo = 0;  // (a stream too long for LDS: the in-order walk)
            }
            // (levels + sort + executor in one launch: whenever the out-of-order walk runs with four waves in the latency regime)
            pc.fused_pass = pc.levels_ooo && pc.levels_threads <= 512 && max_tiles_in_group <= 4 && pc.n_groups <= simds && !(h->opt_tune & 64);
            if (!pc.levels_ooo) {
                pc.levels_threads = std::min(256, (max_tiles_in_group * 16 + 63) / 64 * 64);
                pc.lds_levels = egg_pk_levels_mr_lds_bytes(s.pk_lev_cap, pc.max_group_particles, pc.levels_threads);
            }
            const size_t sort_words = (size_t)max_tiles_in_group * (size_t)pc.scap + 64;
            pc.chunk_cap = (int)std::min<size_t>(sort_words / 64 + (size_t)s.pk_lev_cap + 8, (size_t)1 << 28);
            pc.lds_exec = (size_t)pc.max_group_particles * 16;  // (the chain variant adds a spare slot per lane at launch: 1,280 particles + 64 would cost the eighth wave of a CU)
            pc.lds_sort = egg_align16((size_t)(s.pk_lev_cap + 2) * 4) + sort_words * 4;
            if (pc.lds_sort > 64 * 1024) pc.lds_sort = 0;
            if (pc.lds_exec + 64 * 16 > h->lds_limit || pc.lds_levels > h->lds_limit || sort_words >= ((size_t)1 << 26)) {
                s.pk_meta_host.resize(meta_mark);
                continue;
            }
            if (pc.fused_pass) {
                pc.lds_pass = std::max(pc.lds_levels, pc.lds_exec + 64 * 16 + (size_t)EGG_PK_RING_BYTES);
                if (pc.lds_pass > h->lds_limit) pc.fused_pass = 0;
            }
            pc.sort_cap = (int)sort_words;
            pc.max_tiles_in_group = max_tiles_in_group;
            pc.lev_lds_now = pc.lev_lds_cap;
            pc.lds_levels_now = pc.lds_levels;
            pc.lds_pass_now = pc.lds_pass;
            if (getenv("EGGSIM_DEBUG"))
                fprintf(stderr, "eggsim: type %d packed class: %d tiles of <= %d particles in %d groups, grid cells %d (%s), lists: %d threads, %d staged partners, %zu B LDS; levels: %s, %d threads, %zu B LDS (level array %d entries); pass fused %d, %zu B LDS\n",
                        which, lc.n_tiles, lc.nmax, pc.n_groups, lc.ccap, lc.use_grid ? "dense grid" : "hash", pc.threads_lists, pc.stage_cap, pc.lds_lists,
                        pc.levels_ooo ? "out of order" : "in order", pc.levels_threads, pc.lds_levels, pc.lev_lds_cap, pc.fused_pass, pc.lds_pass);
            pc.chunk_base = s.pk_chunk_words;
            s.pk_chunk_words += (size_t)pc.n_groups * (size_t)pc.chunk_cap;
            lc.packed = (int)s.pk.size();
            s.pk_n = pn;
            s.pk_tiles += lc.n_tiles;
            s.pk_groups += pc.n_groups;
            s.pk_entries += (size_t)lc.n_tiles * (size_t)pc.scap;
            s.pk_sort_words += (size_t)pc.n_groups * (size_t)pc.sort_cap;
            s.pk.push_back(pc);
        }
        if (!s.pk.empty()) {
            // the claims of every tile atom, in tile order (one load per atom slot in the kernels)
            s.pk_meta_claims = (s.pk_meta_host.size() + 3) & ~(size_t)3;
            s.pk_meta_host.resize(s.pk_meta_claims);
            for (int32_t a : s.tile_atoms) {
                const Box &c = claim[(size_t)a];
                const int32_t rec[4] = {c.lo_x, c.lo_y, c.hi_x, c.hi_y};
                s.pk_meta_host.insert(s.pk_meta_host.end(), rec, rec + 4);
            }
            lap(5);
            const size_t np = (size_t)s.pk_n, nt = (size_t)s.pk_tiles, ng = (size_t)s.pk_groups;
            HIP_TRY(h, s.pk_meta.reserve(s.pk_meta_host.size() + 4, false, s.stream));
            HIP_TRY(h, s.pk_src.reserve(np, false, s.stream));
            HIP_TRY(h, s.pk_atom.reserve(np, false, s.stream));
            HIP_TRY(h, s.pk_aslot.reserve(np + 8, false, s.stream));
            HIP_TRY(h, s.pk_pos.reserve(2 * np, false, s.stream));
            HIP_TRY(h, s.pk_prev.reserve(2 * np, false, s.stream));
            HIP_TRY(h, s.pk_wr.reserve(2 * np, false, s.stream));
            HIP_TRY(h, s.pk_ckey.reserve(2 * np, false, s.stream));
            HIP_TRY(h, s.pk_lists.reserve(s.pk_entries + 64, false, s.stream));
            HIP_TRY(h, s.pk_lvl.reserve(s.pk_entries + 64, false, s.stream));
            bool any_ooo = false;
            for (const PackedClass &pc : s.pk) any_ooo |= pc.levels_ooo != 0;
            if (any_ooo) HIP_TRY(h, s.pk_rank.reserve(s.pk_entries + 64, false, s.stream));
            HIP_TRY(h, s.pk_sorted.reserve(s.pk_sort_words + 64, false, s.stream));
            HIP_TRY(h, s.pk_chunks.reserve(s.pk_chunk_words + 64, false, s.stream));
            HIP_TRY(h, s.pk_nchunks.reserve(2 * ng + 8, false, s.stream));
            HIP_TRY(h, s.pk_levstart.reserve(ng * ((size_t)s.pk_lev_cap + 2) + 64, false, s.stream));
            HIP_TRY(h, s.pk_tile.reserve(nt * (3 + 2 * EGG_PK_MAX_PASSES) + 4, false, s.stream));
        }
        s.pk_plan_dirty = true;
    }

    HIP_TRY(h, s.d_scratch.reserve(scratch_bytes + 16, false, s.stream));
    lap(6);
    s.h_claim = claim;
    s.meta_dirty = true;
    s.tiling_dirty = false;
    h->stats.retiles++;
    h->stats.n_tiles[which] = (int64_t)tiles.size();
    h->stats.max_tile_particles[which] = tiles.empty() ? 0 : tiles.front().particles;
    lap(7);
    return EGG_OK;
}

}  // namespace egghost
