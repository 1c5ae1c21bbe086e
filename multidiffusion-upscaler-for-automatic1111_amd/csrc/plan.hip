// Grid planning: host-side integer logic + the small device lookup tables the gather-formulated kernels use.
// Upstream behaviour followed: tile_utils/utils.py:160-177 (split_bboxes), tile_methods/abstractdiffusion.py:173-186
// (init_grid_bbox), scripts/tilevae.py:390-462 (split_tiles / get_best_tile_size).
#include <cmath>
#include <vector>

#include "common.h"

namespace mdt {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace mdt

extern "C" int mdtile_version(void) { return MDTILE_VERSION; }

// ---- arithmetic of the matrix-core kernels: split-bf16 ("bf16x3") by default, exact fp32 MFMA on request ----
namespace mdt {
static int g_precision = [] {
    const char* c = getenv("MDTILE_CONV_MODE");
    const char* a = getenv("MDTILE_ATTN_MODE");
    return ((c && strcmp(c, "f32") == 0) ? 1 : 0) | ((a && strcmp(a, "f32") == 0) ? 2 : 0);
}();
bool conv_strict_f32() { return (g_precision & 1) != 0; }
bool attn_strict_f32() { return (g_precision & 2) != 0; }
}  // namespace mdt
extern "C" int mdtile_set_precision(int mode) {
    if (mode != MDTILE_PRECISION_BF16X3 && mode != MDTILE_PRECISION_F32) {
        mdt::set_error("mdtile_set_precision: unknown mode %d", mode);
        return MDTILE_E_ARG;
    }
    mdt::g_precision = mode == MDTILE_PRECISION_F32 ? 3 : 0;
    return MDTILE_OK;
}
extern "C" int mdtile_get_precision(void) { return mdt::g_precision == 3 ? MDTILE_PRECISION_F32 : MDTILE_PRECISION_BF16X3; }
extern "C" const char* mdtile_last_error(void) { return mdt::g_err; }

// 1-D origins: count = ceil((extent-ov)/(tile-ov)); step is a double; origin = min(trunc(i*step), extent-tile).
static void origins_1d(int extent, int tile, int ov, std::vector<int>& out) {
    int n = (int)std::ceil((double)(extent - ov) / (double)(tile - ov));
    if (n < 1) n = 1;
    double step = n > 1 ? (double)(extent - tile) / (double)(n - 1) : 0.0;
    out.resize(n);
    for (int i = 0; i < n; ++i) {
        int o = (int)((double)i * step);
        out[i] = o < extent - tile ? o : extent - tile;
    }
}

// per canvas coordinate: first tile index covering it and how many consecutive tiles do (origins are non-decreasing)
static void cover_ranges(int extent, int tile, const std::vector<int>& org, std::vector<int>& packed) {
    packed.assign(extent, 0);
    int n = (int)org.size();
    for (int p = 0; p < extent; ++p) {
        int first = -1, cnt = 0;
        for (int i = 0; i < n; ++i)
            if (org[i] <= p && p < org[i] + tile) {
                if (first < 0) first = i;
                ++cnt;
            }
        packed[p] = (first < 0 ? 0 : first) | (cnt << 16);
    }
}

extern "C" mdtile_plan* mdtile_plan_create(int w, int h, int tile_w, int tile_h, int overlap, int tile_bs, int clamp) {
    if (w <= 0 || h <= 0 || tile_w <= 0 || tile_h <= 0 || tile_bs <= 0 || w > 65535 || h > 65535) {
        mdt::set_error("mdtile_plan_create: bad arguments w=%d h=%d tile=%dx%d bs=%d", w, h, tile_w, tile_h, tile_bs);
        return nullptr;
    }
    int tw = tile_w, th = tile_h, ov = overlap;
    if (clamp) {
        tw = tile_w < w ? tile_w : w;
        th = tile_h < h ? tile_h : h;
        int mn = tile_w < tile_h ? tile_w : tile_h;  // NOTE: requested, not clamped, sizes (abstractdiffusion.py:178)
        ov = overlap < mn - 4 ? overlap : mn - 4;
        if (ov < 0) ov = 0;
    } else if (tw > w || th > h || ov < 0) {
        mdt::set_error("mdtile_plan_create: raw grid needs tile <= canvas and overlap >= 0");
        return nullptr;
    }
    if (tw - ov == 0 || th - ov == 0) {  // upstream raises ZeroDivisionError here (utils.py:161-162)
        mdt::set_error("mdtile_plan_create: overlap %d equals the canvas-clamped tile %dx%d (division by zero upstream)", ov, tw, th);
        return nullptr;
    }
    std::vector<int> xs, ys, cr, rr;
    origins_1d(w, tw, ov, xs);
    origins_1d(h, th, ov, ys);
    if (xs.size() > 32767 || ys.size() > 32767) {
        mdt::set_error("mdtile_plan_create: too many tiles");
        return nullptr;
    }
    cover_ranges(w, tw, xs, cr);
    cover_ranges(h, th, ys, rr);

    mdtile_plan* p = new mdtile_plan();
    p->w = w; p->h = h; p->tw = tw; p->th = th; p->ov = ov;
    p->cols = (int)xs.size(); p->rows = (int)ys.size(); p->T = p->cols * p->rows;
    p->num_batches = (p->T + tile_bs - 1) / tile_bs;
    p->tile_bs = (p->T + p->num_batches - 1) / p->num_batches;
    // one host block [xs | ys | colrange | rowrange]; mirrored to the device lazily (mdt::plan_upload) so that the
    // integer planning works on a machine without a GPU.
    const int W4 = (w + 3) / 4;
    const size_t head = xs.size() + ys.size() + cr.size() + rr.size();
    p->quad_off = (head + 3) & ~(size_t)3;
    p->table_len = p->quad_off + 4 * (size_t)W4 + 4 * (size_t)h;
    p->h_table = new int[p->table_len]();
    int* q = p->h_table;
    p->h_xs = q; memcpy(q, xs.data(), xs.size() * sizeof(int)); q += xs.size();
    p->h_ys = q; memcpy(q, ys.data(), ys.size() * sizeof(int)); q += ys.size();
    memcpy(q, cr.data(), cr.size() * sizeof(int)); q += cr.size();
    memcpy(q, rr.data(), rr.size() * sizeof(int));
    // 16-byte records (see common.h).  A quad's candidate columns are the union over its (in-canvas) pixels; origins are
    // non-decreasing and every pixel is covered, so the union is a contiguous index range.
    auto origin3 = [](const std::vector<int>& org, int first, int k) { return first + k < (int)org.size() ? org[first + k] : 0; };
    int* cq = p->h_table + p->quad_off;
    p->nc_max = p->nr_max = 0;
    for (int xq = 0; xq < W4; ++xq) {
        int first = 1 << 30, last = -1;
        for (int j = 0; j < 4 && 4 * xq + j < w; ++j) {
            const int f = cr[4 * xq + j] & 0xffff, n = cr[4 * xq + j] >> 16;
            if (n == 0) continue;
            if (f < first) first = f;
            if (f + n - 1 > last) last = f + n - 1;
        }
        if (last < 0) { first = 0; last = -1; }
        cq[4 * xq + 0] = first | ((last - first + 1) << 16);
        if (last - first + 1 > p->nc_max) p->nc_max = last - first + 1;
        for (int k = 0; k < 3; ++k) cq[4 * xq + 1 + k] = origin3(xs, first, k);
    }
    int* ri = cq + 4 * (size_t)W4;
    for (int y = 0; y < h; ++y) {
        const int f = rr[y] & 0xffff;
        ri[4 * y + 0] = rr[y];
        if ((rr[y] >> 16) > p->nr_max) p->nr_max = rr[y] >> 16;
        for (int k = 0; k < 3; ++k) ri[4 * y + 1 + k] = origin3(ys, f, k);
    }
    p->d_xs = p->d_ys = p->d_colrange = p->d_rowrange = nullptr;
    p->d_colquad = p->d_rowinfo = nullptr;
    return p;
}

namespace mdt {
int plan_upload(const mdtile_plan* cp) {
    mdtile_plan* p = const_cast<mdtile_plan*>(cp);
    if (p->d_xs) return MDTILE_OK;
    int* d = nullptr;
    MDT_HIP(hipMalloc(&d, p->table_len * sizeof(int)));
    hipError_t e = hipMemcpy(d, p->h_table, p->table_len * sizeof(int), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        (void)hipFree(d);
        set_error("plan_upload: hipMemcpy failed: %s", hipGetErrorString(e));
        return MDTILE_E_HIP;
    }
    p->d_xs = d; p->d_ys = d + p->cols; p->d_colrange = p->d_ys + p->rows; p->d_rowrange = p->d_colrange + p->w;
    p->d_colquad = reinterpret_cast<int4*>(d + p->quad_off);          // hipMalloc is 256-B aligned, quad_off % 4 == 0
    p->d_rowinfo = p->d_colquad + (p->w + 3) / 4;
    return MDTILE_OK;
}
}  // namespace mdt

extern "C" void mdtile_plan_destroy(mdtile_plan* p) {
    if (!p) return;
    if (p->d_xs) (void)hipFree(p->d_xs);
    delete[] p->h_table;
    delete p;
}

extern "C" int mdtile_plan_info(const mdtile_plan* p, int* info8) {
    MDT_CHECK_ARG(p && info8, "mdtile_plan_info: null argument");
    info8[0] = p->cols; info8[1] = p->rows; info8[2] = p->T; info8[3] = p->num_batches;
    info8[4] = p->tile_bs; info8[5] = p->tw; info8[6] = p->th; info8[7] = p->ov;
    return MDTILE_OK;
}

extern "C" int mdtile_plan_bboxes(const mdtile_plan* p, int* xywh) {
    MDT_CHECK_ARG(p && xywh, "mdtile_plan_bboxes: null argument");
    for (int r = 0; r < p->rows; ++r)
        for (int c = 0; c < p->cols; ++c) {
            int* o = xywh + 4 * (r * p->cols + c);
            o[0] = p->h_xs[c]; o[1] = p->h_ys[r]; o[2] = p->tw; o[3] = p->th;
        }
    return MDTILE_OK;
}

// ---- Tiled VAE split (scripts/tilevae.py:390-462) -----------------------------------------------------------
static int best_tile_size(int lower, int upper) {
    for (int div = 32; div >= 2; div /= 2) {
        int rem = lower % div;
        if (rem == 0) return lower;
        int cand = lower - rem + div;
        if (cand <= upper) return cand;
    }
    return lower;
}

// get_best_tile_size (scripts/tilevae.py:390-403): smallest multiple of 32 / 16 / 8 / 4 / 2 >= lowerbound that still fits under upperbound
extern "C" int mdtile_vae_best_tile_size(int lowerbound, int upperbound) { return best_tile_size(lowerbound, upperbound); }

extern "C" int mdtile_vae_split_tiles(int h, int w, int tile_size, int is_decoder, int* in_bboxes, int* out_bboxes, int cap) {
    MDT_CHECK_ARG(h > 0 && w > 0 && tile_size > 0, "mdtile_vae_split_tiles: bad arguments");
    const int pad = is_decoder ? 11 : 32;
    int nh = (int)std::ceil((double)(h - 2 * pad) / (double)tile_size);
    int nw = (int)std::ceil((double)(w - 2 * pad) / (double)tile_size);
    if (nh < 1) nh = 1;
    if (nw < 1) nw = 1;
    int rh = best_tile_size((int)std::ceil((double)(h - 2 * pad) / (double)nh), tile_size);
    int rw = best_tile_size((int)std::ceil((double)(w - 2 * pad) / (double)nw), tile_size);
    int n = nh * nw;
    if (cap <= 0 || !in_bboxes || !out_bboxes) return n;
    MDT_CHECK_ARG(cap >= n, "mdtile_vae_split_tiles: capacity %d < %d tiles", cap, n);
    auto mn = [](int a, int b) { return a < b ? a : b; };
    auto mx = [](int a, int b) { return a > b ? a : b; };
    for (int i = 0; i < nh; ++i)
        for (int j = 0; j < nw; ++j) {
            int b[4] = {pad + j * rw, mn(pad + (j + 1) * rw, w), pad + i * rh, mn(pad + (i + 1) * rh, h)};
            int o[4] = {b[0] > pad ? b[0] : 0, b[1] < w - pad ? b[1] : w, b[2] > pad ? b[2] : 0, b[3] < h - pad ? b[3] : h};
            int* ob = out_bboxes + 4 * (i * nw + j);
            int* ib = in_bboxes + 4 * (i * nw + j);
            for (int k = 0; k < 4; ++k) ob[k] = is_decoder ? o[k] * 8 : o[k] / 8;  // Python // on non-negative ints
            ib[0] = mx(0, b[0] - pad); ib[1] = mn(w, b[1] + pad); ib[2] = mx(0, b[2] - pad); ib[3] = mn(h, b[3] + pad);
        }
    return n;
}
