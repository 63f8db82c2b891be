// Headless renderer: the reference's draw path (SimulationHandler:draw, simulation_handler.lua:158-161) as HIP kernels for
// gfx950, for image-level regression without LOVE / OpenGL.  "L:" = /root/reference/simulation_handler.lua.
//
//   pass 1, per type (_update_canvases, L:1995-2113; simulation_handler_instanced_draw.glsl): every particle is a quad
//     textured with the gaussian density texture (simulation_handler_particle_texture.glsl), scaled by radius *
//     texture_scale, stretched along its velocity by the motion blur, blended into the type's canvas with the "screen",
//     "premultiplied" blend: dst = src + dst * (1 - src) per channel.
//   pass 2, per type (_draw_canvases, L:2117-2175): outline (simulation_handler_outline.glsl) and thresholding +
//     lighting (simulation_handler_lighting.glsl), alpha-blended onto the screen.
//
// GPU mapping.  GL rasterises the quads in instance order and blends in that order; blending is order dependent at the
// rounding level.  Here the canvas is cut into 16 x 16 px tiles, every particle is appended to the list of each tile its
// quad's bounding box touches (two passes: count, fill), and one workgroup per tile sorts its list by particle index in
// LDS and blends the particles in that order, one thread per pixel, the accumulator in registers: the image is the one an
// in-order rasteriser produces, independent of scheduling, and no pixel is ever written twice.  The 38 x 38 texture sits
// in LDS with a zero border ("clampzero" wrap without a branch).  Everything is IEEE float32 without contraction; the
// CPU model in oracle/render_model.py computes the same expressions in the same order.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "eggsim_device.h"

namespace {

__device__ __forceinline__ float4 f4(float x, float y, float z, float w) { return make_float4(x, y, z, w); }

// instance record of one particle in canvas space (instanced_draw.glsl:14-46)
struct Instance {
    float cx, cy, c, s;      // quad centre; cos / sin of the velocity angle
    float isx, isy, ex, ey;  // 1 / half extent along / across the velocity; half extents of the bounding box
};

__device__ __forceinline__ bool make_instance(const EggRenderArgs &A, int i, Instance &I) {
    const float one = 1.0f, t = A.t;
    const float x = (float)A.x[i], y = (float)A.y[i], lx = (float)A.last_x[i], ly = (float)A.last_y[i];
    const float vx = (float)A.vx[i], vy = (float)A.vy[i], rad = (float)A.radius[i];
    const float ox = lx * (one - t) + x * t;  // mix(previous, current, interpolation_alpha)
    const float oy = ly * (one - t) + y * t;
    const float speed = sqrtf(vx * vx + vy * vy);
    const float base = rad * A.texture_scale;
    const float sx = base * (one + speed * A.motion_blur), sy = base;
    I.c = speed > 0.0f ? vx / speed : one;  // cos(atan(vy, vx)), sin(atan(vy, vx))
    I.s = speed > 0.0f ? vy / speed : 0.0f;
    I.cx = ox + A.tx;
    I.cy = oy + A.ty;
    I.isx = one / sx;
    I.isy = one / sy;
    I.ex = fabsf(I.c) * sx + fabsf(I.s) * sy;
    I.ey = fabsf(I.s) * sx + fabsf(I.c) * sy;
    // (a NaN anywhere fails these comparisons: the particle is not drawn)
    return fabsf(I.cx) < 1.0e7f && fabsf(I.cy) < 1.0e7f && I.ex < 1.0e6f && I.ey < 1.0e6f && sx > 0.0f && sy > 0.0f;
}

// tiles touched by the bounding box, with a pixel of slack on every side; false if none
__device__ __forceinline__ bool tile_range(const EggRenderArgs &A, const Instance &I, int &tx0, int &ty0, int &tx1, int &ty1) {
    const float x0 = floorf(I.cx - I.ex - 1.5f), x1 = ceilf(I.cx + I.ex + 0.5f);
    const float y0 = floorf(I.cy - I.ey - 1.5f), y1 = ceilf(I.cy + I.ey + 0.5f);
    if (x1 < 0.0f || y1 < 0.0f || x0 >= (float)A.cw || y0 >= (float)A.ch) return false;
    tx0 = (int)fmaxf(x0, 0.0f) / EGG_RENDER_TILE;
    ty0 = (int)fmaxf(y0, 0.0f) / EGG_RENDER_TILE;
    tx1 = (int)fminf(x1, (float)(A.cw - 1)) / EGG_RENDER_TILE;
    ty1 = (int)fminf(y1, (float)(A.ch - 1)) / EGG_RENDER_TILE;
    return true;
}

template <bool Fill>
__device__ __forceinline__ void bin_particles(const EggRenderArgs &A) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= A.n) return;
    Instance I;
    if (!make_instance(A, i, I)) return;
    int tx0, ty0, tx1, ty1;
    if (!tile_range(A, I, tx0, ty0, tx1, ty1)) return;
    for (int ty = ty0; ty <= ty1; ++ty)
        for (int tx = tx0; tx <= tx1; ++tx) {
            const int tile = ty * A.tiles_x + tx;
            if (Fill)
                A.entries[atomicAdd(&A.tile_cursor[tile], 1u)] = (uint32_t)i;
            else
                atomicAdd(&A.tile_count[tile], 1u);
        }
}

__device__ __forceinline__ float smoothstep_f(float e0, float e1, float x) {
    float t = (x - e0) / (e1 - e0);
    t = fminf(fmaxf(t, 0.0f), 1.0f);
    return t * t * (3.0f - 2.0f * t);
}

// GL_LINEAR, clamp to edge, of a (h, w) float4 canvas at texture coordinates (u, v)
__device__ __forceinline__ float4 sample_clamp(const float4 *canvas, int w, int h, float u, float v) {
    const float tu = u * (float)w - 0.5f, tv = v * (float)h - 0.5f;
    const float i0f = floorf(tu), j0f = floorf(tv);
    const float fu = tu - i0f, fv = tv - j0f;
    // (the float clamp first: a far-off coordinate must not overflow the int conversion)
    const int i0 = (int)fminf(fmaxf(i0f, -1.0f), (float)w), j0 = (int)fminf(fmaxf(j0f, -1.0f), (float)h);
    const int ia = min(max(i0, 0), w - 1), ib = min(max(i0 + 1, 0), w - 1);
    const int ja = min(max(j0, 0), h - 1), jb = min(max(j0 + 1, 0), h - 1);
    const float4 a = canvas[(size_t)ja * w + ia], b = canvas[(size_t)ja * w + ib];
    const float4 c = canvas[(size_t)jb * w + ia], d = canvas[(size_t)jb * w + ib];
    const float gu = 1.0f - fu, gv = 1.0f - fv;
    return f4((a.x * gu + b.x * fu) * gv + (c.x * gu + d.x * fu) * fv, (a.y * gu + b.y * fu) * gv + (c.y * gu + d.y * fu) * fv,
              (a.z * gu + b.z * fu) * gv + (c.z * gu + d.z * fu) * fv, (a.w * gu + b.w * fu) * gv + (c.w * gu + d.w * fu) * fv);
}

// love "alpha", "alphamultiply"
__device__ __forceinline__ float4 blend_alpha(float4 dst, float4 src) {
    const float k = 1.0f - src.w;
    return f4(src.x * src.w + dst.x * k, src.y * src.w + dst.y * k, src.z * src.w + dst.z * k, src.w + dst.w * k);
}

__device__ __forceinline__ void normalize3(float &x, float &y, float &z) {
    const float len = sqrtf((x * x + y * y) + z * z);
    x = x / len;
    y = y / len;
    z = z / len;
}

}  // namespace

extern "C" __global__ void __launch_bounds__(256) egg_render_count_kernel(EggRenderArgs A) { bin_particles<false>(A); }
extern "C" __global__ void __launch_bounds__(256) egg_render_fill_kernel(EggRenderArgs A) { bin_particles<true>(A); }

// exclusive scan of the tile counts (one workgroup); also the longest list
extern "C" __global__ void __launch_bounds__(1024) egg_render_scan_kernel(EggRenderArgs A) {
    __shared__ uint32_t part[1024];
    __shared__ uint32_t longest[1024];
    const int n = A.tiles_x * A.tiles_y, tid = threadIdx.x;
    const int per = (n + 1023) / 1024, lo = tid * per, hi = min(lo + per, n);
    uint32_t sum = 0, mx = 0;
    for (int k = lo; k < hi; ++k) {
        sum += A.tile_count[k];
        mx = max(mx, A.tile_count[k]);
    }
    part[tid] = sum;
    longest[tid] = mx;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {
        const uint32_t v = tid >= d ? part[tid - d] : 0u, m = tid >= d ? longest[tid - d] : 0u;
        __syncthreads();
        part[tid] += v;
        longest[tid] = max(longest[tid], m);
        __syncthreads();
    }
    uint32_t run = tid ? part[tid - 1] : 0u;
    for (int k = lo; k < hi; ++k) {
        A.tile_start[k] = run;
        A.tile_cursor[k] = run;
        run += A.tile_count[k];
    }
    if (tid == 1023) {
        A.tile_start[n] = part[1023];
        A.totals[0] = part[1023];
        A.totals[1] = longest[1023];
    }
}

// one workgroup per canvas tile, one thread per pixel
extern "C" __global__ void __launch_bounds__(256) egg_render_splat_kernel(EggRenderArgs A) {
    extern __shared__ __align__(16) unsigned char lds_raw[];
    const int T = A.tsize, TP = T + 2;  // texture with a zero border
    float *tex = (float *)lds_raw;
    Instance *stage = (Instance *)(lds_raw + (((size_t)TP * TP * 4 + 15) & ~(size_t)15));
    float4 *stage_col = (float4 *)(stage + EGG_RENDER_STAGE);
    uint32_t *list = (uint32_t *)(stage_col + EGG_RENDER_STAGE);
    const int tid = threadIdx.x;
    const int tile = blockIdx.x, tile_x = tile % A.tiles_x, tile_y = tile / A.tiles_x;
    const uint32_t begin = A.tile_start[tile], count = A.tile_start[tile + 1] - begin;
    const int px_i = tile_x * EGG_RENDER_TILE + (tid & (EGG_RENDER_TILE - 1)), px_j = tile_y * EGG_RENDER_TILE + tid / EGG_RENDER_TILE;
    const bool live = px_i < A.cw && px_j < A.ch;
    float4 dst = f4(0.0f, 0.0f, 0.0f, 0.0f);  // love.graphics.clear(0, 0, 0, 0) (L:2087, L:2102)
    if (count) {
        for (int k = tid; k < TP * TP; k += 256) {
            const int i = k % TP - 1, j = k / TP - 1;
            tex[k] = (i >= 0 && i < T && j >= 0 && j < T) ? A.texture[j * T + i] : 0.0f;
        }
        // the list in ascending particle index: GL draws (and blends) the instances in that order
        uint32_t padded = 1;
        while (padded < count) padded <<= 1;
        for (uint32_t k = tid; k < padded; k += 256) list[k] = k < count ? A.entries[begin + k] : 0xFFFFFFFFu;
        __syncthreads();
        for (uint32_t size = 2; size <= padded; size <<= 1)
            for (uint32_t stride = size >> 1; stride > 0; stride >>= 1) {
                for (uint32_t k = tid; k < padded / 2; k += 256) {
                    const uint32_t lo = 2 * k - (k & (stride - 1)), hi = lo + stride;
                    const uint32_t a = list[lo], b = list[hi];
                    const bool up = (lo & size) == 0;
                    if ((a > b) == up) {
                        list[lo] = b;
                        list[hi] = a;
                    }
                }
                __syncthreads();
            }
        const float px = (float)px_i + 0.5f, py = (float)px_j + 0.5f, half = 0.5f, one = 1.0f, tsize = (float)T;
        for (uint32_t base = 0; base < count; base += EGG_RENDER_STAGE) {
            const uint32_t m = min((uint32_t)EGG_RENDER_STAGE, count - base);
            __syncthreads();
            if ((uint32_t)tid < m) {
                const int i = (int)list[base + tid];
                Instance I;
                make_instance(A, i, I);
                stage[tid] = I;
                // the particle's batch colour: atom of particle i by bisection over the atoms' first particles
                int lo = 0, hi = A.n_atoms - 1;
                while (lo < hi) {
                    const int mid = (lo + hi + 1) >> 1;
                    if (A.atom_offset[mid] <= i)
                        lo = mid;
                    else
                        hi = mid - 1;
                }
                float4 col = A.atom_color[lo];
                if (A.premultiply) col = f4(col.x * col.w, col.y * col.w, col.z * col.w, col.w);  // L:2035-2041
                stage_col[tid] = col;
            }
            __syncthreads();
            if (!live) continue;
            for (uint32_t k = 0; k < m; ++k) {
                const Instance I = stage[k];
                const float dx = px - I.cx, dy = py - I.cy;
                const float u = (dx * I.c + dy * I.s) * I.isx;  // quad coordinates in [-1, 1]
                const float v = (dy * I.c - dx * I.s) * I.isy;
                if (!(fabsf(u) <= one && fabsf(v) <= one)) continue;
                const float tu = (u * half + half) * tsize - half, tv = (v * half + half) * tsize - half;
                const float i0f = floorf(tu), j0f = floorf(tv);
                const float fu = tu - i0f, fv = tv - j0f;
                const float *row = tex + ((int)j0f + 1) * TP + ((int)i0f + 1);
                const float top = row[0] * (one - fu) + row[1] * fu;
                const float bot = row[TP] * (one - fu) + row[TP + 1] * fu;
                const float g = top * (one - fv) + bot * fv;
                const float4 col = stage_col[k];
                const float4 src = f4(g * col.x, g * col.y, g * col.z, g * col.w);  // texture * color * color_override
                dst = f4(src.x + dst.x * (one - src.x), src.y + dst.y * (one - src.y), src.z + dst.z * (one - src.z),
                         src.w + dst.w * (one - src.w));  // "screen", "premultiplied"
            }
        }
    }
    if (live) A.canvas[(size_t)px_j * A.cw + px_i] = dst;
}

// _draw_canvases: one thread per screen pixel, the layers (white, yolk) in order
extern "C" __global__ void __launch_bounds__(256) egg_render_composite_kernel(EggCompositeArgs A) {
    const int sx_i = blockIdx.x * 16 + (threadIdx.x & 15), sy_i = blockIdx.y * 16 + (threadIdx.x >> 4);
    if (sx_i >= A.screen_w || sy_i >= A.screen_h) return;
    const float sx = (float)sx_i + 0.5f, sy = (float)sy_i + 0.5f;
    const float psx = 1.0f / (float)A.screen_w, psy = 1.0f / (float)A.screen_h;  // 1 / love_ScreenSize
    float4 out = A.screen[(size_t)sy_i * A.screen_w + sx_i];
    float4 color = f4(1.0f, 1.0f, 1.0f, 1.0f);  // love's current colour: set only inside the outline branch (L:2137-2142)
    for (int layer = 0; layer < A.n_layers; ++layer) {
        const EggCompositeLayer &L = A.layer[layer];
        const float dx = sx - L.x0, dy = sy - L.y0;
        const bool cover = dx >= 0.0f && dx < (float)L.w && dy >= 0.0f && dy < (float)L.h;
        if (L.outline_thickness > 0.0f) color = L.color;  // (uniform over the image: taken even where the quad does not cover)
        if (!cover) continue;
        const float u = dx / (float)L.w, v = dy / (float)L.h;
        const float4 data = sample_clamp(L.canvas, L.w, L.h, u, v);
        if (L.outline_thickness > 0.0f && data.w != 0.0f) {  // outline.glsl: `discard` where the canvas is empty
            const int steps = (int)ceilf(L.outline_thickness) + 1;
            const float step_size = L.outline_thickness / (float)steps;
            const float diag = 0.70710678118654752440f;
            const float dirs[8][2] = {{1.0f, 0.0f}, {-1.0f, 0.0f}, {0.0f, 1.0f}, {0.0f, -1.0f},
                                      {diag, diag}, {-diag, diag}, {diag, -diag}, {-diag, -diag}};
            float max_alpha = 0.0f;
            for (int d = 0; d < 8; ++d)
                for (int step = 1; step <= steps; ++step) {
                    const float reach = (float)step * step_size;
                    const float4 s = sample_clamp(L.canvas, L.w, L.h, u + (dirs[d][0] * reach) * psx, v + (dirs[d][1] * reach) * psy);
                    max_alpha = fmaxf(max_alpha, s.w);
                }
            max_alpha = fminf(max_alpha, 1.0f);
            const float e0 = 0.5f * A.threshold;
            const float oa = smoothstep_f(e0, e0 + 0.035f, max_alpha);
            out = blend_alpha(out, f4(L.outline_color.x * oa, L.outline_color.y * oa, L.outline_color.z * oa, L.outline_color.w * oa));
        }
        // lighting.glsl
        const float value = smoothstep_f(A.threshold - A.smoothness, A.threshold + A.smoothness, data.w);
        float4 center;
        if (A.use_particle_color)
            center = f4(data.x * color.x, data.y * color.y, data.z * color.z, value * color.w);
        else
            center = f4(value * color.x, value * color.y, value * color.z, value * color.w);
        auto a_at = [&](float ox, float oy) { return sample_clamp(L.canvas, L.w, L.h, u + ox * psx, v + oy * psy).w; };
        const float tl = a_at(-1.0f, -1.0f), tm = a_at(0.0f, -1.0f), tr = a_at(1.0f, -1.0f);
        const float ml = a_at(-1.0f, 0.0f), mr = a_at(1.0f, 0.0f);
        const float bl = a_at(-1.0f, 1.0f), bm = a_at(0.0f, 1.0f), br = a_at(1.0f, 1.0f);
        const float gx = -tl + tr - 2.0f * ml + 2.0f * mr - bl + br;
        const float gy = -tl - 2.0f * tm - tr + bl + 2.0f * bm + br;
        float nx = -gx, ny = -gy, nz = 1.0f;
        normalize3(nx, ny, nz);
        float specular = 0.0f, shadow = 0.0f;
        if (L.highlight_strength > 0.0f && A.use_lighting) {
            float lx = 1.0f, ly = -1.0f, lz = 1.0f;
            normalize3(lx, ly, lz);
            float hx = lx + 0.0f, hy = ly + 0.0f, hz = lz + 1.0f;
            normalize3(hx, hy, hz);
            const float d = fmaxf((nx * hx + ny * hy) + nz * hz, 0.0f);
            const float d2 = d * d, d4 = d2 * d2, d8 = d4 * d4, d16 = d8 * d8, d32 = d16 * d16;
            specular = specular + L.highlight_strength * (d32 * d16);  // pow(., 48)
        }
        if (L.shadow_strength > 0.0f && A.use_lighting) {
            float lx = -0.5f, ly = 0.75f, lz = 0.0f;
            normalize3(lx, ly, lz);
            const float sh = (nx * lx + ny * ly) + nz * lz;
            shadow = smoothstep_f(0.0f, 1.0f, fminf(fmaxf(sh * L.shadow_strength, 0.0f), 1.0f));
        }
        out = blend_alpha(out, f4(center.x - shadow + specular, center.y - shadow + specular, center.z - shadow + specular, center.w));
    }
    A.screen[(size_t)sy_i * A.screen_w + sx_i] = out;
}

extern "C" __global__ void __launch_bounds__(256) egg_render_clear_kernel(float4 *dst, size_t n, float4 value) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) dst[i] = value;
}
