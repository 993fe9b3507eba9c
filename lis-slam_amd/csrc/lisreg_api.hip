// lisreg_api.hip — C-ABI host layer of liblisreg.so (declared in include/lisreg.h).
//
// Creates the seam the reference lacks: each entry point replaces a piece of
// scan2SubMapOptimization() — /root/reference/src/node/odomEstimationNode.cpp:596-626 and the copies at
// src/node/subMapOptmizationNode.cpp:1509-1541, 4485-4540.  The GN loop is enqueued in full (optional index rebuild,
// optional source sort, `bound` x {correspondence kernel, solve kernel}, finalize) with no host synchronisation inside
// lisreg_batch_run (the synchronous entry points may stop launching early once every item has converged); a
// context is single-threaded and owns its stream, so several contexts run concurrently (callers #2/#3).
// There is deliberately NO CPU fallback: without a HIP device every compute entry point fails with
// LISREG_ERR_HIP.
#include "lisreg_ctx.hpp"

#include <dlfcn.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

using namespace lisreg;

namespace lisreg {
thread_local std::string g_static_err;
int ctx_fail(lisreg_ctx* c, int code, const std::string& msg)
{
    if (c) c->err = msg; else g_static_err = msg;
    return code;
}
}  // namespace lisreg

namespace {


int fail(lisreg_ctx* c, int code, const std::string& msg) { return lisreg::ctx_fail(c, code, msg); }

float env_float(const char* name, float dflt)
{
    const char* s = getenv(name);
    return s ? (float)atof(s) : dflt;
}

void pose_to_matrix_host(const float T[6], float M[12])
{
    // pcl::getTransformation via trans2Affine3f (src/core/common.cpp:54-57)
    float A = cosf(T[2]), B = sinf(T[2]), C = cosf(T[1]), D = sinf(T[1]), E = cosf(T[0]), F = sinf(T[0]);
    float DE = D * E, DF = D * F;
    M[0] = A * C;  M[1] = A * DF - B * E;  M[2]  = B * F + A * DE;  M[3]  = T[3];
    M[4] = B * C;  M[5] = A * E + B * DF;  M[6]  = B * DE - A * F;  M[7]  = T[4];
    M[8] = -D;     M[9] = C * F;           M[10] = C * E;           M[11] = T[5];
}

// pack PCL structs (stride/format of common.h:9,25-35) into 16-B device records
}  // namespace
namespace lisreg {
void pack_cloud(const void* cloud, int n, int stride, int fmt, lisreg_dpoint* out)
{
    const unsigned char* b = static_cast<const unsigned char*>(cloud);
    for (int i = 0; i < n; ++i) {
        const unsigned char* r = b + (size_t)i * (size_t)stride;
        memcpy(&out[i], r, 12);
        uint16_t lab = 0;
        if (fmt == LISREG_FMT_XYZIL) memcpy(&lab, r + 20, 2);
        out[i].payload = lab;
    }
}
}  // namespace lisreg
namespace {

DevParams make_dev_params(const lisreg_params& p)
{
    DevParams d;
    memset(&d, 0, sizeof d);
    d.tau = p.knn_sq_thresh; d.line_ratio = p.line_ratio; d.plane_tol = p.plane_tol; d.accept_s = p.accept_s;
    d.conv_deg = p.conv_deg; d.conv_cm = p.conv_cm; d.eig_thresh = p.eig_thresh;
    d.min_corr = p.min_corr; d.use_label = p.use_label_weight; d.emulate_shadow = p.emulate_matp_shadow;
    d.skip_empty = p.skip_empty_target; d.fixed_iters = p.fixed_iters;
    d.bound = p.fixed_iters > 0 ? p.fixed_iters : p.max_iters;
    d.edge_min = p.edge_min; d.surf_min = p.surf_min; d.use_imu = p.use_imu_blend;
    d.imu_w = p.imu_rpy_weight; d.rot_tol = p.rotation_tol; d.z_tol = p.z_tol;
    for (int i = 0; i < 32; ++i) d.wtab[i] = (float)(2.0 - (double)p.label_score[i]);   // subMapOptmizationNode.cpp:1671
    return d;
}

}  // namespace
namespace lisreg {
SortBuffers sort_buffers(lisreg_ctx* c)
{
    SortBuffers sb;
    sb.hist = c->hist.as<int>(); sb.bucket_start = c->bucket_start.as<int>(); sb.scan_tmp = c->scan_tmp.as<int>();
    sb.elem_bucket = c->elem_bucket.as<uint32_t>(); sb.elem_sub = c->elem_sub.as<uint32_t>();
    sb.tmp_bucket = c->tmp_bucket.as<uint32_t>(); sb.tmp_sub = c->tmp_sub.as<uint32_t>();
    sb.tmp_idx = c->tmp_idx.as<int>();
    sb.tmp_pts = c->tmp_pts.as<float4>();
    return sb;
}
}  // namespace lisreg
namespace {

}  // namespace
namespace lisreg {
int ensure_sort_scratch(lisreg_ctx* c, size_t n_elems, size_t n_buckets)
{
    HIPCHK(c, c->hist.ensure(sizeof(int) * (n_buckets + 1)));
    HIPCHK(c, c->bucket_start.ensure(sizeof(int) * (n_buckets + 2)));
    HIPCHK(c, c->scan_tmp.ensure(sizeof(int) * (n_buckets / 2048 + 4)));
    HIPCHK(c, c->elem_bucket.ensure(sizeof(uint32_t) * (n_elems + 1)));
    HIPCHK(c, c->elem_sub.ensure(sizeof(uint32_t) * (n_elems + 1)));
    HIPCHK(c, c->tmp_bucket.ensure(sizeof(uint32_t) * (n_elems + 1)));
    HIPCHK(c, c->tmp_sub.ensure(sizeof(uint32_t) * (n_elems + 1)));
    HIPCHK(c, c->tmp_idx.ensure(sizeof(int) * (n_elems + 1)));
    HIPCHK(c, c->tmp_pts.ensure(sizeof(float4) * (n_elems + 1)));
    return LISREG_OK;
}
}  // namespace lisreg
namespace {

// grid geometry from a bounding box; cell edge grows if the box would need too many cells
}  // namespace
namespace lisreg {
void make_grid(const float bb_in[6], int n, GridIndex* g, int* n_cells, int margin_cells)
{
    float bb[6] = { bb_in[0], bb_in[1], bb_in[2], bb_in[3], bb_in[4], bb_in[5] };
    memset(g, 0, sizeof *g);
    g->n = n;
    // Cell edge: 0.5 m is the measured optimum for the 200 k-point submap of BASELINE configs[1]; the optimum
    // scales with the point spacing, so denser maps get smaller cells (footprint density as the proxy: lidar maps are
    // surfaces over a ground plane).  1 M points over the same 80 x 80 m: 0.25 m, +14 % registrations/s.  LISREG_CELL overrides.
    float cell = 0.5f;
    if (n > 0) {
        const double area = std::max(1.0, (double)(bb[3] - bb[0]) * (double)(bb[4] - bb[1]));
        cell = (float)std::min(0.5, std::max(0.25, 2.8 / std::sqrt((double)n / area)));
    }
    cell = env_float("LISREG_CELL", cell);
    if (n <= 0) { g->cell = cell; g->inv_cell = 1.f / cell; g->nx = g->ny = g->nz = 0; *n_cells = 1; return; }
    const double max_cells = 1 << 24;
    // registration targets: the grid reaches `margin_cells` cells past the cloud on every side, so that a query a pose error away from
    // a wall that bounds the cloud still has a cell of its own (cell rows, search_mode 5); empty cells cost four bytes of table each
    const float bb0[6] = { bb[0], bb[1], bb[2], bb[3], bb[4], bb[5] };
    for (;;) {
        for (int d = 0; d < 3; ++d) { bb[d] = bb0[d] - (float)margin_cells * cell; bb[3 + d] = bb0[3 + d] + (float)margin_cells * cell; }
        double nx = floor((bb[3] - bb[0]) / cell) + 1, ny = floor((bb[4] - bb[1]) / cell) + 1,
               nz = floor((bb[5] - bb[2]) / cell) + 1;
        if (nx * ny * nz <= max_cells) { g->nx = (int)nx; g->ny = (int)ny; g->nz = (int)nz; break; }
        cell *= 1.26f;
    }
    g->ox = bb[0]; g->oy = bb[1]; g->oz = bb[2];
    g->cell = cell; g->inv_cell = 1.f / cell;
    *n_cells = g->nx * g->ny * g->nz;
}
}  // namespace lisreg
namespace {

constexpr int kCrowGridMargin = 2;

int build_target_kind(lisreg_ctx* c, Target& t, int k)
{
    const int n = t.n[k];
    HIPCHK(c, t.cell_start[k].ensure(sizeof(int) * ((size_t)t.n_cells[k] + 2)));
    HIPCHK(c, t.sorted[k].ensure(sizeof(float4) * (size_t)std::max(n, 1)));
    int rc = ensure_sort_scratch(c, (size_t)std::max(n, 1), (size_t)t.n_cells[k]);
    if (rc) return rc;
    t.g[k].pts = t.sorted[k].as<float4>();
    t.g[k].cell_start = t.cell_start[k].as<int>();
    launch_build_target(t.raw_ptr[k], n, t.g[k], t.sorted[k].as<float4>(), t.cell_start[k].as<int>(), t.n_cells[k],
                        sort_buffers(c), c->stream);
    HIPCHK(c, hipGetLastError());
    return LISREG_OK;
}

// search_mode 3: neighbour lists of target kind k (the index itself must already be enqueued on the stream)
int ensure_graph(lisreg_ctx* c, Target& t, int k, bool launch)
{
    const size_t n = (size_t)std::max(t.n[k], 1);
    HIPCHK(c, t.nbr[k].ensure(sizeof(float4) * kGraphK * n));
    HIPCHK(c, t.nbr_meta[k].ensure(sizeof(float2) * n));
    t.g[k].nbr = t.nbr[k].as<float4>();
    t.g[k].nbr_meta = t.nbr_meta[k].as<float2>();
    if (launch && !t.graph_valid[k]) {
        launch_build_graph_one(t.g[k], c->stream);
        HIPCHK(c, hipGetLastError());
        t.graph_valid[k] = true;
    }
    return LISREG_OK;
}

// search_mode 5: cell rows of target kind k (the index itself must already be enqueued on the stream).  The row count depends on the
// data, so the first build of a target classifies its cells, reads the count back (one host round trip, at set / prepare time) and sizes the
// buffers; rebuilds inside a run (rebuild_targets_each_run) reuse that capacity — a cell that does not fit gets no row and its queries walk.
lisreg::CrowBuffers crow_buffers(Target& t, int k, bool with_reach = false)
{
    lisreg::CrowBuffers cb;
    cb.need = t.crow_need[k].as<int>(); cb.omask = t.crow_omask[k].as<int>(); cb.scan = t.crow_scan[k].as<int>(); cb.scan_tmp = t.crow_scan_tmp[k].as<int>();
    cb.cap_rows = t.crow_cap[k];
    cb.reach = with_reach && t.g[k].qmark ? t.crow_reach[k].as<unsigned>() : nullptr;      // (never for the classification that SIZES the row buffers)
    return cb;
}

// option "row_reach": mark and reach words of target kind k (one bit per grid cell, lisreg_internal.hpp) — made when a batch that rebuilds its
// targets takes the cell rows; the pointer travels in the GridIndex (the device table is refreshed when it changes)
int ensure_reach(lisreg_ctx* c, Target& t, int k, bool on)
{
    unsigned* want = nullptr;
    int w = 0;
    if (on && t.n[k] > 0 && t.g[k].crow_tab) {
        w = (t.g[k].nz + 31) / 32;
        const size_t words = (size_t)t.g[k].nx * (size_t)t.g[k].ny * (size_t)w;
        HIPCHK(c, t.crow_qmark[k].ensure(sizeof(unsigned) * (words + 8)));
        HIPCHK(c, t.crow_reach[k].ensure(sizeof(unsigned) * (words + 8)));
        want = t.crow_qmark[k].as<unsigned>();
    }
    if (t.g[k].qmark != want || t.g[k].qmark_w != w) { t.g[k].qmark = want; t.g[k].qmark_w = w; c->grids_dirty = true; }
    return LISREG_OK;
}

// may_decline (front-end chosen by auto): a target whose rows would not fit "cell_rows_max_mb" is left without them (crow_too_big) and the
// batch takes another front-end; with search_mode 5 set by the caller the buffers are capped there instead and the cells past them walk.
// rows_left (auto): rows the batch may still take under "cell_rows_max_mb" — the bound is on the batch's targets together, not on each
int ensure_crows(lisreg_ctx* c, Target& t, int k, bool may_decline = false, long long* rows_left = nullptr)
{
    if (t.crow_valid[k] && t.g[k].crow_tab) { if (rows_left) *rows_left -= t.crow_cap[k]; return LISREG_OK; }
    if (may_decline && t.crow_too_big[k]) return LISREG_OK;       // found too big before (the note is cleared when the target is set again)
    t.crow_too_big[k] = false;
    if (t.n[k] > 0 && t.grid_margin[k] < kCrowGridMargin) {
        // a wall that bounds the cloud has half of a not-yet-registered scan's points OUTSIDE the cloud's bounding box: the grid is
        // re-made two cells wider on every side (empty cells: four bytes of table each) and the index rebuilt on it, once per target
        t.grid_margin[k] = kCrowGridMargin;
        make_grid(t.bbox[k], t.n[k], &t.g[k], &t.n_cells[k], t.grid_margin[k]);
        int rc = build_target_kind(c, t, k);
        if (rc) return rc;
        t.graph_valid[k] = false;
        t.g[k].nbr = nullptr; t.g[k].nbr_meta = nullptr;
        // the geometry (origin, dims) and possibly the cell table's address have changed: a batch prepared against the old grid holds stale
        // copies of both (the device GridIndex table, the rebuild table h_tsegs) — it has to be prepared again (lisreg_batch_prepare calls
        // this before it reads any geometry; lisreg_get_target_cell_rows on a slot a prepared batch uses lands here too)
        c->prepared = false;
        c->grids_dirty = true;
    }
    const size_t nc = (size_t)std::max(t.n_cells[k], 1);
    HIPCHK(c, t.crow_need[k].ensure(sizeof(int) * (nc + 8)));
    { const size_t before = t.crow_omask[k].cap; HIPCHK(c, t.crow_omask[k].ensure(sizeof(int) * (nc + 8))); if (t.crow_omask[k].cap != before) t.omask_zero[k] = 0; }     // (a new allocation: nothing known)
    HIPCHK(c, t.crow_scan[k].ensure(sizeof(int) * (nc + 8)));
    HIPCHK(c, t.crow_scan_tmp[k].ensure(sizeof(int) * (nc / 2048 + 8)));
    HIPCHK(c, t.crow_tab[k].ensure(sizeof(int) * (nc + 8)));
    t.g[k].crow_tab = t.crow_tab[k].as<int>();
    int rows = 0;
    if (t.n[k] > 0) {
        launch_crow_classify(t.g[k], t.n_cells[k], crow_buffers(t, k), c->stream, &t.omask_zero[k]);
        HIPCHK(c, hipMemcpyAsync(&rows, t.crow_scan[k].as<int>() + t.n_cells[k], sizeof(int), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
    }
    t.crow_cap[k] = std::max(rows, 1);
    {
        // (a cloud scattered through space instead of lying on surfaces — vegetation, rain — asks for up to 125 centre rows per point)
        long long budget = (long long)std::max(c->cell_rows_max_mb, 1) * 1048576LL / (long long)(sizeof(float4) * kGraphK + sizeof(float2));
        if (rows_left) budget = std::min(budget, std::max(*rows_left, 0LL));
        if ((long long)rows > budget) {
            if (may_decline) { t.crow_too_big[k] = true; t.g[k].crow_tab = nullptr; return LISREG_OK; }
            t.crow_cap[k] = (int)std::max(1LL, budget);
        }
    }
    if (const char* e = getenv("LISREG_CROW_CAP_PERCENT"))      // tests: under-size the row buffers (cells past the capacity get no row: their queries walk)
        t.crow_cap[k] = std::max(1, (int)((long long)t.crow_cap[k] * std::max(0, atoi(e)) / 100));
    if (getenv("LISREG_CROW_DEBUG")) fprintf(stderr, "[lisreg] cell rows of kind %d: %d rows for %d points in %d cells (%d x %d x %d of %.3f m)\n", k, rows, t.n[k], t.n_cells[k], t.g[k].nx, t.g[k].ny, t.g[k].nz, t.g[k].cell);
    {
        // (auto: a refused allocation — many own-target items, each under the bound — is a reason to take another front-end, not an error)
        const hipError_t e1 = t.crow[k].ensure(sizeof(float4) * kGraphK * (size_t)t.crow_cap[k]);
        const hipError_t e2 = e1 == hipSuccess ? t.crow_meta[k].ensure(sizeof(float2) * (size_t)t.crow_cap[k]) : e1;
        if (e2 != hipSuccess) {
            (void)hipGetLastError();
            t.crow[k].release(); t.crow_meta[k].release();
            if (may_decline) { t.crow_too_big[k] = true; t.g[k].crow_tab = nullptr; return LISREG_OK; }
            return lisreg::ctx_fail(c, LISREG_ERR_HIP, std::string("cell rows: ") + hipGetErrorString(e2));
        }
    }
    if (rows_left) *rows_left -= t.crow_cap[k];
    t.crow_chosen[k] = true;
    t.g[k].crow = t.crow[k].as<float4>();
    t.g[k].crow_meta = t.crow_meta[k].as<float2>();
    if (t.n[k] > 0) launch_crow_build(t.g[k], t.n_cells[k], crow_buffers(t, k), c->stream, &t.omask_zero[k]);
    else HIPCHK(c, hipMemsetAsync(t.crow_tab[k].p, 0xff, sizeof(int) * nc, c->stream));
    HIPCHK(c, hipGetLastError());
    t.crow_valid[k] = true;
    return LISREG_OK;
}

int upload_grids(lisreg_ctx* c)
{
    std::vector<GridIndex> h(c->targets.size() * 2);
    for (size_t s = 0; s < c->targets.size(); ++s)
        for (int k = 0; k < 2; ++k) {
            if (c->targets[s].valid) h[s * 2 + k] = c->targets[s].g[k];
            else memset(&h[s * 2 + k], 0, sizeof(GridIndex));
        }
    HIPCHK(c, c->grids_dev.ensure(sizeof(GridIndex) * std::max<size_t>(h.size(), 1)));
    const size_t bytes = sizeof(GridIndex) * h.size();
    // through a pinned buffer of the context, guarded by an event: the copy is asynchronous and the host goes on building the batch's
    // tables while the stream still works on the target index it has just been given (a frame loop sets a target per frame: waiting
    // here meant an idle device for the rest of lisreg_batch_prepare)
    if (bytes > c->grids_host_cap) {
        if (c->grids_done) (void)hipEventSynchronize(c->grids_done);
        if (c->grids_host) (void)hipHostFree(c->grids_host);
        c->grids_host = nullptr; c->grids_host_cap = 0;
        if (hipHostMalloc((void**)&c->grids_host, 2 * bytes + 1024, hipHostMallocDefault) == hipSuccess) c->grids_host_cap = 2 * bytes + 1024;
        else { (void)hipGetLastError(); c->grids_host = nullptr; }
    }
    if (!c->grids_done && hipEventCreateWithFlags(&c->grids_done, hipEventDisableTiming) != hipSuccess) c->grids_done = nullptr;
    if (bytes && c->grids_host && c->grids_done) {
        HIPCHK(c, hipEventSynchronize(c->grids_done));          // (recorded by the previous upload: long past)
        memcpy(c->grids_host, h.data(), bytes);
        HIPCHK(c, hipMemcpyAsync(c->grids_dev.p, c->grids_host, bytes, hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipEventRecord(c->grids_done, c->stream));
    } else if (bytes) {
        HIPCHK(c, hipMemcpyAsync(c->grids_dev.p, h.data(), bytes, hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));     // h is a local
    }
    c->grids_dirty = false;
    return LISREG_OK;
}

// (sidx 1: the mark goes on the side stream — the second half of an interleaved run, run_impl; an interval runs from a mark to the next mark
//  on the SAME stream)
void prof_mark(lisreg_ctx* c, int kind_of_next_interval, int sidx = 0)
{
    if (!c->profiling) return;
    hipEvent_t e;
    if (hipEventCreate(&e) != hipSuccess) return;
    (void)hipEventRecord(e, sidx ? c->side_stream : c->stream);
    c->ev.push_back(e);
    c->ev_kind.push_back(kind_of_next_interval);
    c->ev_sidx.push_back(sidx);
}

void prof_collect(lisreg_ctx* c)
{
    for (int i = 0; i < 5; ++i) c->timing[i] = 0;
    const bool dump = getenv("LISREG_PROF_DUMP") != nullptr;
    for (size_t i = 0; i + 1 < c->ev.size(); ++i) {
        float ms = 0;
        size_t j = i + 1;
        while (j < c->ev.size() && c->ev_sidx[j] != c->ev_sidx[i]) ++j;          // the next mark on the same stream
        if (j >= c->ev.size()) continue;
        if (c->ev_kind[i] >= 0 && hipEventElapsedTime(&ms, c->ev[i], c->ev[j]) == hipSuccess) {
            if (dump) fprintf(stderr, "[lisreg prof] interval %zu kind %d: %.4f ms\n", i, c->ev_kind[i], ms);
            if (c->ev_kind[i] == 0) { c->timing[0] += ms; c->timing[1] += 1; }
            else if (c->ev_kind[i] == 1) { c->timing[2] += ms; c->timing[3] += 1; }
            else if (c->ev_kind[i] == 2) c->timing[4] += ms;
        }
    }
    for (auto e : c->ev) (void)hipEventDestroy(e);
    c->ev.clear(); c->ev_kind.clear(); c->ev_sidx.clear();
}

}  // namespace
namespace lisreg {
// HIP-event marks for the other translation units (lisreg_set_profiling / lisreg_get_timing)
void ctx_prof_mark(lisreg_ctx* c, int kind_of_next_interval) { prof_mark(c, kind_of_next_interval); }
void ctx_prof_collect(lisreg_ctx* c) { if (!c->ev.empty()) prof_collect(c); }
}  // namespace lisreg

// =============================================================================================================
extern "C" {

int lisreg_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int lisreg_create(int device, lisreg_ctx** out)
{
    if (!out) return fail(nullptr, LISREG_ERR_ARG, "lisreg_create: out is NULL");
    *out = nullptr;
    int n = lisreg_device_count();
    if (n <= 0) return fail(nullptr, LISREG_ERR_HIP, "lisreg_create: no HIP device visible (liblisreg has no CPU fallback)");
    if (device < 0 || device >= n) return fail(nullptr, LISREG_ERR_ARG, "lisreg_create: bad device index");
    lisreg_ctx* c = new (std::nothrow) lisreg_ctx();
    if (!c) return fail(nullptr, LISREG_ERR_NOMEM, "lisreg_create: out of host memory");
    c->device = device;
    if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking) != hipSuccess) {
        delete c;
        return fail(nullptr, LISREG_ERR_HIP, "lisreg_create: hipSetDevice/hipStreamCreate failed");
    }
    c->stream = c->own_stream;
    c->targets.resize(1);
    if (hipHostMalloc((void**)&c->done_host, sizeof(int), hipHostMallocDefault) != hipSuccess) c->done_host = nullptr;
    if (c->done_dev.ensure(sizeof(int)) != hipSuccess) { lisreg_destroy(c); return fail(nullptr, LISREG_ERR_HIP, "lisreg_create: hipMalloc failed"); }
    lisreg_default_params(LISREG_VARIANT_ODOM, &c->params);
    if (const char* m = getenv("LISREG_SEARCH_MODE")) {            // the same values lisreg_set_option("search_mode") takes
        const int v = atoi(m);
        if (v == 0 || v == 1 || v == 3 || v == 4 || v == 5) c->search_mode = v;
        else fprintf(stderr, "[lisreg] LISREG_SEARCH_MODE=%s ignored (0, 1, 3, 4 or 5)\n", m);
    }
    if (const char* m = getenv("LISREG_SORT_SOURCES")) c->sort_sources = atoi(m);
    if (const char* m = getenv("LISREG_EXACT")) c->exact = atoi(m) != 0;
    if (const char* m = getenv("LISREG_CANONICAL_TIES")) c->canonical_ties = atoi(m) != 0;
    if (const char* m = getenv("LISREG_FIRST_PASS_MM")) c->first_pass_r = 1e-3f * (float)atoi(m);
    if (const char* m = getenv("LISREG_WIDE_UNTIL")) c->wide_until = atoi(m);
    if (const char* m = getenv("LISREG_GRAPH_WIDE_UNTIL")) c->graph_wide_until = atoi(m);
    if (const char* m = getenv("LISREG_CROW_WIDE_UNTIL")) c->crow_wide_until = atoi(m);
    if (const char* m = getenv("LISREG_GRAPH_HOPS")) c->graph_hops = atoi(m);
    if (const char* m = getenv("LISREG_CELL_ANCHOR_UNTIL")) c->cell_anchor_until = std::max(atoi(m), 0);
    if (const char* m = getenv("LISREG_XCD_ORDER")) c->xcd_order = atoi(m);
    if (const char* m = getenv("LISREG_ROW_REACH")) c->row_reach = atoi(m);
    if (const char* m = getenv("LISREG_GRAPH_MIN_RATIO")) c->graph_min_ratio = atoi(m);
    if (const char* m = getenv("LISREG_CELL_MIN_RATIO")) c->cell_min_ratio = atoi(m);
    if (const char* m = getenv("LISREG_WIDE_FROM")) c->wide_from = c->wide_from_small = atoi(m);
    *out = c;
    return LISREG_OK;
}

void lisreg_destroy(lisreg_ctx* c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    lisreg_comm_destroy(c);
    feeder_destroy(c);
    for (auto& t : c->targets) for (int k = 0; k < 2; ++k) { t.raw[k].release(); t.sorted[k].release(); t.cell_start[k].release(); t.nbr[k].release(); t.nbr_meta[k].release();
        t.crow[k].release(); t.crow_meta[k].release(); t.crow_tab[k].release(); t.crow_need[k].release(); t.crow_omask[k].release(); t.crow_scan[k].release(); t.crow_scan_tmp[k].release(); t.crow_qmark[k].release(); t.crow_reach[k].release(); }
    DevBuf* bufs[] = { &c->grids_dev, &c->hist, &c->bucket_start, &c->scan_tmp, &c->elem_bucket, &c->elem_sub,
                       &c->tmp_bucket, &c->tmp_sub, &c->tmp_idx, &c->tmp_pts, &c->bbox_dev, &c->bbox_scratch, &c->blocks, &c->segs,
                       &c->items, &c->sorted_all, &c->order_all, &c->partials, &c->results, &c->trace, &c->src_upload, &c->raw_upload, &c->dbg_nn, &c->blocks_q, &c->coef, &c->coef_ok, &c->nn, &c->counters, &c->tseg_dev, &c->tblk_dev, &c->tchunk_dev, &c->strip_tab, &c->done_dev, &c->xcd_tab, &c->vox_in, &c->vox_lab, &c->vox_order, &c->vox_sidx,
                       &c->vox_head, &c->vox_slot, &c->vox_start, &c->vox_out, &c->vox_outlab, &c->vox_M,
                       &c->ft_owner, &c->ft_flag, &c->ft_pos, &c->ft_scan, &c->ft_col, &c->ft_range, &c->ft_src, &c->ft_curv,
                       &c->ft_picked, &c->ft_label, &c->ft_rlists, &c->ft_rcounts, &c->ft_lists, &c->ft_counts, &c->ft_rings,
                       &c->ft_gather, &c->ft_cat, &c->ft_bounds, &c->ft_dsk_tab, &c->ft_dsk_pts, &c->ft_dsk_misc, &c->ft_dsk_time };
    for (auto b : bufs) b->release();
    for (auto& kv : c->maps) { auto& m = kv.second; m.raw.release(); m.sorted.release(); m.cell_start.release(); m.g_dev.release(); }
    for (auto& m : c->localmaps) { for (auto& b : m.cls) b.release(); m.tgt[0].release(); m.tgt[1].release(); }
    for (auto& r : c->keyrings) { for (auto& f : r.frames) { f.cloud[0].release(); f.cloud[1].release(); } r.cat[0].release(); r.cat[1].release(); r.tgt[0].release(); r.tgt[1].release(); }
    DevBuf* mbufs[] = { &c->lm_in, &c->lm_tmp, &c->lm_bbox, &c->exact_trig, &c->mp_pts, &c->mp_flag, &c->mp_pos, &c->mp_idx, &c->mp_cnt, &c->mp_d2, &c->mp_out, &c->icp_state, &c->icp_partials, &c->icp_cur, &c->icp_items, &c->map_tab, &c->map_tsegs, &c->map_tblocks, &c->map_stage };
    for (auto b : mbufs) b->release();
    for (auto e : c->ev) (void)hipEventDestroy(e);
    if (c->done_host) (void)hipHostFree(c->done_host);
    if (c->stage_host) (void)hipHostFree(c->stage_host);
    if (c->fetch_host) (void)hipHostFree(c->fetch_host);
    if (c->stage_done) (void)hipEventDestroy(c->stage_done);
    if (c->side_stream) { (void)hipStreamSynchronize(c->side_stream); (void)hipStreamDestroy(c->side_stream); }
    if (c->grids_done) { (void)hipEventSynchronize(c->grids_done); (void)hipEventDestroy(c->grids_done); }
    if (c->grids_host) (void)hipHostFree(c->grids_host);
    if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
    if (c->ev_join) (void)hipEventDestroy(c->ev_join);
    if (c->ev_ab) (void)hipEventDestroy(c->ev_ab);
    if (c->ev_ba) (void)hipEventDestroy(c->ev_ba);
    if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
    delete c;
}

const char* lisreg_last_error(const lisreg_ctx* c) { return c ? c->err.c_str() : g_static_err.c_str(); }

int lisreg_set_stream(lisreg_ctx* c, void* s)
{
    if (!c) return LISREG_ERR_ARG;
    (void)hipStreamSynchronize(c->stream);
    c->stream = s ? (hipStream_t)s : c->own_stream;
    return LISREG_OK;
}

void* lisreg_get_stream(const lisreg_ctx* c) { return c ? (void*)c->stream : nullptr; }

int lisreg_default_params(int variant, lisreg_params* p)
{
    if (!p || variant < 1 || variant > 3) return LISREG_ERR_ARG;
    memset(p, 0, sizeof *p);
    static const float score[20] = { 1.0f, 1.0f, 0.6f, 0.5f, 0.8f, 0.5f, 0.5f, 0.5f, 0.5f, 1.2f,     // config/label.yaml:214-234
                                     1.2f, 1.2f, 0.5f, 1.0f, 0.8f, 0.5f, 1.3f, 0.5f, 1.5f, 1.5f };
    p->max_iters = variant == 1 ? 15 : (variant == 2 ? 20 : 30);
    p->fixed_iters = 0;
    p->knn_sq_thresh = variant == 1 ? 1.0f : 2.0f;
    p->conv_deg = variant == 1 ? 0.005f : (variant == 2 ? 0.003f : 0.002f);
    p->conv_cm = variant == 1 ? 0.05f : (variant == 2 ? 0.03f : 0.02f);
    p->min_corr = 50; p->eig_thresh = 100.f; p->edge_min = -1; p->surf_min = 100;
    p->line_ratio = 3.f; p->plane_tol = 0.2f; p->accept_s = 0.1f;
    p->use_label_weight = variant == 1 ? 0 : 1;
    for (int i = 0; i < 32; ++i) p->label_score[i] = i < 20 ? score[i] : 0.0f;    // a label the std::map does not hold reads as 0 (:1671)
    p->emulate_matp_shadow = 1;
    p->skip_empty_target = variant == 3 ? 1 : 0;
    p->use_imu_blend = variant == 3 ? 0 : 1;
    p->imu_rpy_weight = 0.1f; p->rotation_tol = 1000.f; p->z_tol = 1000.f;
    return LISREG_OK;
}

void lisreg_pose_to_matrix(const float T[6], float M[12]) { pose_to_matrix_host(T, M); }

// transformUpdate, host form (odomEstimationNode.cpp:976-1006): tf's double-precision quaternion slerp of the
// single-axis roll / pitch rotations, then constraintTransformation (common.cpp:285-291).
void lisreg_transform_update(const lisreg_params* p, const lisreg_imu* imu, float T[6])
{
    auto blend = [](double a, double b, double w, bool is_pitch) {
        // slerp between two rotations about ONE axis by angles a and b: quaternions (sin(a/2), cos(a/2))
        double qa_s = sin(a * 0.5), qa_c = cos(a * 0.5), qb_s = sin(b * 0.5), qb_c = cos(b * 0.5);
        double d = qa_s * qb_s + qa_c * qb_c;
        double theta = d < 0 ? acos(-d) : acos(d);
        double rs = qa_s, rc = qa_c;
        if (theta != 0.0) {
            double dd = 1.0 / sin(theta), s0 = sin((1.0 - w) * theta), s1 = sin(w * theta), sg = d < 0 ? -1.0 : 1.0;
            rs = (qa_s * s0 + sg * qb_s * s1) * dd;
            rc = (qa_c * s0 + sg * qb_c * s1) * dd;
        }
        double n2 = rs * rs + rc * rc, s = 2.0 / n2;
        if (!is_pitch) {                       // rotation about X: m21 = 2wx, m22 = 1 - 2xx ; getRPY roll = atan2(m21, m22)
            return atan2(rc * rs * s, 1.0 - rs * rs * s);
        }
        double m20 = -(rc * rs * s);           // rotation about Y: m20 = xz - wy = -2wy
        if (fabs(m20) >= 1) return m20 < 0 ? M_PI / 2.0 : -M_PI / 2.0;
        return -asin(m20);
    };
    if (p->use_imu_blend && imu && imu->imu_available && fabsf(imu->imu_pitch_init) < 1.4f) {
        T[0] = (float)blend((double)T[0], (double)imu->imu_roll_init, (double)p->imu_rpy_weight, false);
        T[1] = (float)blend((double)T[1], (double)imu->imu_pitch_init, (double)p->imu_rpy_weight, true);
    }
    auto clampf = [](float v, float lim) { if (v < -lim) v = -lim; if (v > lim) v = lim; return v; };
    T[0] = clampf(T[0], p->rotation_tol);
    T[1] = clampf(T[1], p->rotation_tol);
    T[5] = clampf(T[5], p->z_tol);
}

// ---- target -------------------------------------------------------------------------------------------------
static int set_target_impl(lisreg_ctx* c, int slot, const void* clouds[2], const int counts[2], int stride, int fmt)
{
    if (!c) return LISREG_ERR_ARG;
    if (slot < 0 || slot > 65535) return fail(c, LISREG_ERR_ARG, "set_target: bad slot");
    for (int k = 0; k < 2; ++k)
        if (counts[k] < 0 || (counts[k] > 0 && !clouds[k])) return fail(c, LISREG_ERR_ARG, "set_target: NULL cloud with n > 0");
    for (int k = 0; k < 2; ++k)          // the kernels address a target's 16-byte records by 32-bit byte offsets from a scalar base
        if (counts[k] >= (1 << 28)) return fail(c, LISREG_ERR_ARG, "set_target: a cloud of 2^28 points or more");
    if (fmt != LISREG_FMT_DEVICE && stride < 12) return fail(c, LISREG_ERR_ARG, "set_target: stride < 12");
    if (fmt == LISREG_FMT_XYZIL && stride < 22) return fail(c, LISREG_ERR_ARG, "set_target: XYZIL needs stride >= 22");
    HIPCHK(c, hipSetDevice(c->device));
    if ((size_t)slot >= c->targets.size()) c->targets.resize((size_t)slot + 1);
    Target& t = c->targets[(size_t)slot];
    t.gen = ++c->target_gen;
    float bb_dev[2][8] = { { 0 }, { 0 } };
    if (fmt == LISREG_FMT_DEVICE && (counts[0] > 0 || counts[1] > 0)) {          // both bounding boxes, one read-back
        HIPCHK(c, c->bbox_dev.ensure(sizeof(float) * 16));
        HIPCHK(c, c->bbox_scratch.ensure(sizeof(float) * 2 * 6 * 256));
        for (int k = 0; k < 2; ++k)
            if (counts[k] > 0)
                launch_bbox(static_cast<const float4*>(clouds[k]), counts[k], c->bbox_dev.as<float>() + 8 * k, c->bbox_scratch.as<float>() + 6 * 256 * k, c->stream);
        HIPCHK(c, hipMemcpyAsync(bb_dev, c->bbox_dev.p, sizeof bb_dev, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
    }
    for (int k = 0; k < 2; ++k) {
        const int n = counts[k];
        t.n[k] = n;
        float bb[6] = { 0, 0, 0, 0, 0, 0 };
        if (fmt == LISREG_FMT_DEVICE) {
            t.raw_external[k] = true;
            t.raw_ptr[k] = static_cast<const float4*>(clouds[k]);
            if (n > 0) memcpy(bb, bb_dev[k], sizeof bb);
        } else {
            std::vector<lisreg_dpoint> h((size_t)std::max(n, 1));
            if (n > 0) pack_cloud(clouds[k], n, stride, fmt, h.data());
            for (int d = 0; d < 3; ++d) { bb[d] = 3.0e38f; bb[3 + d] = -3.0e38f; }
            for (int i = 0; i < n; ++i) {
                const float v[3] = { h[(size_t)i].x, h[(size_t)i].y, h[(size_t)i].z };
                for (int d = 0; d < 3; ++d) { bb[d] = std::min(bb[d], v[d]); bb[3 + d] = std::max(bb[3 + d], v[d]); }
            }
            HIPCHK(c, t.raw[k].ensure(sizeof(float4) * (size_t)std::max(n, 1)));
            if (n > 0) HIPCHK(c, hipMemcpy(t.raw[k].p, h.data(), sizeof(float4) * (size_t)n, hipMemcpyHostToDevice));
            t.raw_external[k] = false;
            t.raw_ptr[k] = t.raw[k].as<float4>();
        }
        for (int d = 0; d < 6; ++d)
            if (n > 0 && !std::isfinite(bb[d])) return fail(c, LISREG_ERR_ARG, "set_target: the cloud has infinite coordinates (NaN points are ignored, Inf is not indexable)");
        if (n > 0 && !(bb[0] <= bb[3] && bb[1] <= bb[4] && bb[2] <= bb[5]))
            return fail(c, LISREG_ERR_ARG, "set_target: the cloud has no finite point (every coordinate is NaN)");
        // the cell rows (search front-end 5) want a grid that reaches two cells past the cloud; when the front-end is left to the batch
        // (auto), a target gets that margin only once a batch has chosen the cell rows for it (ensure_crow_margin)
        // — and keeps it from then on: a slot whose targets went through the cell rows last time (a frame stream re-sets its slot
        // every frame) is indexed with the margin straight away, instead of being built twice per set
        memcpy(t.bbox[k], bb, sizeof bb);
        t.grid_margin[k] = (c->search_mode == 5 || (c->search_mode == 4 && t.crow_chosen[k])) ? kCrowGridMargin : 0;
        make_grid(bb, n, &t.g[k], &t.n_cells[k], t.grid_margin[k]);
        prof_mark(c, 2);
        int rc = build_target_kind(c, t, k);
        t.graph_valid[k] = false;
        t.crow_valid[k] = false; t.crow_too_big[k] = false;
        t.g[k].crow = nullptr; t.g[k].crow_meta = nullptr; t.g[k].crow_tab = nullptr;
        if (!rc && c->search_mode == 3) rc = ensure_graph(c, t, k, true);
        if (!rc && c->search_mode == 5) rc = ensure_crows(c, t, k);
        prof_mark(c, -1);
        if (rc) return rc;
    }
    t.valid = true;
    c->grids_dirty = true;
    c->prepared = false;
    return LISREG_OK;
}

int lisreg_set_target_slot(lisreg_ctx* c, int slot, const void* corner, int n_corner, const void* surf, int n_surf,
                           int stride, int fmt)
{
    const void* clouds[2] = { corner, surf };
    const int counts[2] = { n_corner, n_surf };
    return set_target_impl(c, slot, clouds, counts, stride, fmt);
}

int lisreg_set_target(lisreg_ctx* c, const void* corner, int n_corner, const void* surf, int n_surf, int stride, int fmt)
{
    return lisreg_set_target_slot(c, 0, corner, n_corner, surf, n_surf, stride, fmt);
}

int lisreg_target_from_classes(lisreg_ctx* c, int slot, const void* pole, int n_pole, const void* ground, int n_ground,
                               const void* building, int n_building, const void* dynamic, int n_dynamic,
                               int stride, int fmt)
{
    if (!c) return LISREG_ERR_ARG;
    if (fmt == LISREG_FMT_DEVICE) return fail(c, LISREG_ERR_ARG, "target_from_classes: host clouds only");
    if (stride < 12) return fail(c, LISREG_ERR_ARG, "target_from_classes: stride < 12");
    // surf = ground + building + dynamic, the concat order of extractSlidingCloud (subMapOptmizationNode.cpp:1408-1419)
    const void* parts[3] = { ground, building, dynamic };
    const int cnt[3] = { std::max(n_ground, 0), std::max(n_building, 0), std::max(n_dynamic, 0) };
    std::vector<unsigned char> surf((size_t)(cnt[0] + cnt[1] + cnt[2]) * (size_t)stride + 1);
    size_t off = 0;
    for (int i = 0; i < 3; ++i) if (cnt[i] > 0 && parts[i]) {
        memcpy(surf.data() + off, parts[i], (size_t)cnt[i] * (size_t)stride);
        off += (size_t)cnt[i] * (size_t)stride;
    }
    return lisreg_set_target_slot(c, slot, pole, pole ? std::max(n_pole, 0) : 0, surf.data(), (int)(off / (size_t)stride), stride, fmt);
}

// ---- batch --------------------------------------------------------------------------------------------------
static bool xcd_order_wanted(const lisreg_ctx* c);

int lisreg_batch_prepare(lisreg_ctx* c, int n_items, const lisreg_item* items, const lisreg_params* params,
                         const float* T_init)
{
    if (!c) return LISREG_ERR_ARG;
    if (n_items < 0 || (n_items > 0 && (!items || !T_init)) || !params) return fail(c, LISREG_ERR_ARG, "batch_prepare: bad arguments");
    HIPCHK(c, hipSetDevice(c->device));
    c->prepared = false;
    c->params = *params;
    c->prm = make_dev_params(*params);
    c->prm.exact = c->exact ? 1 : 0;
    c->prm.ties = (c->canonical_ties || c->exact) ? 1 : 0;
#ifdef LISREG_XP_HOOKS      /* timing experiments that change results are compiled in only on request (csrc/Makefile: XP=1), never into the shipped library */
    c->prm.freeze_pose = getenv("LISREG_XP_FREEZE_POSE") ? 1 : 0;          // tests/ab.sh: see DevParams
    if (c->prm.freeze_pose) fprintf(stderr, "[lisreg] LISREG_XP_FREEZE_POSE is set: poses are NOT updated (timing experiment)\n");
#else
    c->prm.freeze_pose = 0;
#endif
    c->n_items = n_items;
    c->h_blocks.clear(); c->h_segs.clear(); c->h_items.assign((size_t)n_items, ItemState());
    c->batch_slots.clear();
    // front-end of this batch.  The k-NN graph costs ~1 ns per target point and saves ~0.015 ns per query-iteration
    // (MI355X): it pays for shared / long-lived targets (a batch of scans against one submap), not for
    // one-shot targets (a loop-closure candidate pair, a single odometry frame).
    {
        long long total_src = 0;
        double t_pts = 0;
        std::vector<int> seen;
        for (int i = 0; i < n_items; ++i) {
            total_src += (long long)std::max(items[i].n_corner, 0) + (long long)std::max(items[i].n_surf, 0);
            const int sl = items[i].target;
            if (sl >= 0 && (size_t)sl < c->targets.size() && std::find(seen.begin(), seen.end(), sl) == seen.end()) {
                seen.push_back(sl);
                t_pts += (double)c->targets[(size_t)sl].n[0] + (double)c->targets[(size_t)sl].n[1];
            }
        }
        if (total_src > 2000000000LL) return fail(c, LISREG_ERR_ARG, "batch_prepare: more than 2e9 source points in one batch");
        c->mode_now = c->search_mode;
        // the LDS-staged box (front-end 0, the first version of the search, kept for cross-checks) keeps the first point met on equal
        // distances and has no canonical re-selection: it cannot honour "canonical_ties" / "exact_arithmetic", so it refuses them
        if (c->search_mode == 0 && (c->canonical_ties || c->exact))
            return fail(c, LISREG_ERR_ARG, "batch_prepare: search_mode 0 does not implement canonical_ties / exact_arithmetic (use 1, 3 or 4)");
        if (c->search_mode == 4) {
            const double qi = (double)total_src * (double)c->prm.bound;
            c->mode_now = (t_pts > 0 && qi >= (double)c->graph_min_ratio * t_pts) ? 3 : 1;
            if (t_pts > 0 && qi >= (double)c->cell_min_ratio * t_pts && t_pts * 5120.0 <= (double)c->cell_rows_max_mb * 1048576.0) c->mode_now = 5;
        }
        // a batch this small cannot fill the chip with one lane per query: eight lanes share a query (k_assoc_walk<.., 8>)
        c->lanes_q = (c->mode_now == 1 && c->lanes_per_query_auto && total_src > 0 && total_src <= 131072) ? 8 : 1;
        // the cell rows re-make a target's grid with a margin the first time they are chosen for it: before anything below reads the geometry
        if (c->mode_now == 5) {
            bool declined = false;
            long long rows_left = (long long)std::max(c->cell_rows_max_mb, 1) * 1048576LL / (long long)(sizeof(float4) * kGraphK + sizeof(float2));
            for (int sl : seen)
                for (int k = 0; k < 2 && !declined; ++k) {
                    Target& t = c->targets[(size_t)sl];
                    if (!t.valid) continue;
                    const bool had = t.crow_valid[k] && t.g[k].crow_tab;
                    int rc = ensure_crows(c, t, k, c->search_mode == 4, c->search_mode == 4 ? &rows_left : nullptr);
                    if (rc) return rc;
                    if (!had) c->grids_dirty = true;
                    declined = c->search_mode == 4 && t.crow_too_big[k];
                }
            if (declined) {
                // (auto only) the rows of the batch's targets would not fit "cell_rows_max_mb" together: the graph scan — and the rows already
                // made for the batch's other targets go back (they would only hold memory)
                c->mode_now = 3;
                for (int sl : seen)
                    for (int k = 0; k < 2; ++k) {
                        Target& t = c->targets[(size_t)sl];
                        if (t.crow_valid[k]) { t.crow[k].release(); t.crow_meta[k].release(); t.crow_valid[k] = false; t.g[k].crow = nullptr; t.g[k].crow_meta = nullptr; t.g[k].crow_tab = nullptr; c->grids_dirty = true; }
                        t.crow_chosen[k] = false;         // (the next set_target of this slot builds its grid without the rows' two-cell margin again)
                    }
            }
        }
    }
    // (after the front-end choice: the cell rows re-make a target's grid with a margin, and the tile count follows the grids)
    // 2-D sort columns of the (optional) source sort: 0.25 m tiles over every item's target footprint.  The bucket count is kept
    // in 64 bits and the tile grows until the whole batch fits 2^26 buckets (km-scale submaps x large batches would otherwise
    // overflow the int32 bucket numbering and ask for gigabytes of histogram).
    float tile = env_float("LISREG_TILE", 0.25f);
    for (;;) {
        long long total = 0;
        for (int i = 0; i < n_items; ++i) {
            if (items[i].target < 0 || (size_t)items[i].target >= c->targets.size()) continue;
            for (int k = 0; k < 2; ++k) {
                const GridIndex& g = c->targets[(size_t)items[i].target].g[k];
                total += (long long)std::max(1.0, std::ceil((double)g.nx * g.cell / tile)) * (long long)std::max(1.0, std::ceil((double)g.ny * g.cell / tile));
            }
        }
        if (total <= (1LL << 26)) break;
        tile *= 1.5f;
    }
    const int qpb = kBlockQ / c->lanes_q;                  // queries per workgroup of the kQ-lane search (h_blocks_q)
    c->h_blocks_q.clear();
    int flat = 0, bucket = 0;
    for (int i = 0; i < n_items; ++i) {
        const lisreg_item& in = items[i];
        if (in.fmt != LISREG_FMT_DEVICE) return fail(c, LISREG_ERR_ARG, "batch_prepare: items must be LISREG_FMT_DEVICE (use lisreg_align_batch for host clouds)");
        if (in.n_corner < 0 || in.n_surf < 0 || (in.n_corner > 0 && !in.src_corner) || (in.n_surf > 0 && !in.src_surf))
            return fail(c, LISREG_ERR_ARG, "batch_prepare: NULL source cloud with n > 0");
        if (in.target < 0 || (size_t)in.target >= c->targets.size() || !c->targets[(size_t)in.target].valid)
            return fail(c, LISREG_ERR_NO_TARGET, "batch_prepare: item refers to a target slot that was never set");
        if (std::find(c->batch_slots.begin(), c->batch_slots.end(), in.target) == c->batch_slots.end()) c->batch_slots.push_back(in.target);
        const Target& t = c->targets[(size_t)in.target];
        ItemState& st = c->h_items[(size_t)i];
        memset(&st, 0, sizeof st);
        memcpy(st.T_init, T_init + 6 * (size_t)i, 24);
        memcpy(st.T, st.T_init, 24);
        st.degenerate_in = in.degenerate_in;
        st.imu = in.imu;
        st.n_sc = in.n_corner; st.n_ss = in.n_surf;
        if (!(st.n_sc > c->prm.edge_min && st.n_ss > c->prm.surf_min)) ++c->prm.n_guard_failed;      // (:598; the done counter's value after a reset)
        st.blk_begin = (int)c->h_blocks.size();
        for (int k = 0; k < 2; ++k) {
            Segment sg;
            memset(&sg, 0, sizeof sg);
            sg.src = static_cast<const float4*>(k == 0 ? in.src_corner : in.src_surf);
            sg.n = k == 0 ? in.n_corner : in.n_surf;
            sg.item = i; sg.kind = k; sg.target = in.target * 2 + k;
            sg.flat_base = flat; sg.bucket_base = bucket;
            const GridIndex& g = t.g[k];
            sg.tox = g.ox; sg.toy = g.oy; sg.toz = g.oz;
            sg.inv_tile = 1.f / tile;
            sg.tnx = std::max(1, (int)ceilf(g.nx * g.cell / tile));
            sg.tny = std::max(1, (int)ceilf(g.ny * g.cell / tile));
            sg.tnz = 1;
            const int seg_id = (int)c->h_segs.size();
            c->h_segs.push_back(sg);
            for (int s = 0; s < sg.n; s += kBlockQ)
                c->h_blocks.push_back(BlockDesc{ seg_id, s, std::min(kBlockQ, sg.n - s), i });
            if (c->lanes_q > 1)
                for (int s = 0; s < sg.n; s += qpb)
                    c->h_blocks_q.push_back(BlockDesc{ seg_id, s, std::min(qpb, sg.n - s), i });
            flat += sg.n;
            bucket += sg.tnx * sg.tny;
        }
        st.blk_count = (int)c->h_blocks.size() - st.blk_begin;
    }
    c->n_blocks = (int)c->h_blocks.size();
    c->n_segs = (int)c->h_segs.size();
    c->n_elems = flat;
    c->n_buckets = std::max(bucket, 1);
    int rc = LISREG_OK;
    if (c->pack_pending) {                        // sources staged by lisreg_stage_host_items: the device waits for the uploads, the host does not
        HIPCHK(c, hipStreamWaitEvent(c->stream, c->pack_pending, 0));
        c->pack_pending = nullptr;
        c->pack_in_use = c->pack_last;
    }
    HIPCHK(c, c->blocks.ensure(sizeof(BlockDesc) * (size_t)std::max(c->n_blocks, 1)));
    HIPCHK(c, c->segs.ensure(sizeof(Segment) * (size_t)std::max(c->n_segs, 1)));
    HIPCHK(c, c->items.ensure(sizeof(ItemState) * (size_t)std::max(n_items, 1)));
    HIPCHK(c, c->sorted_all.ensure(sizeof(float4) * (size_t)std::max(flat, 1)));
    HIPCHK(c, c->order_all.ensure(sizeof(int) * (size_t)std::max(flat, 1)));
    HIPCHK(c, c->nn.ensure(sizeof(int) * 5 * (size_t)std::max(flat, 1)));
    if (c->lanes_q > 1) {
        HIPCHK(c, c->blocks_q.ensure(sizeof(BlockDesc) * std::max<size_t>(c->h_blocks_q.size(), 1)));
        HIPCHK(c, c->coef.ensure(sizeof(float4) * (size_t)std::max(flat, 1)));
        HIPCHK(c, c->coef_ok.ensure(sizeof(int) * (size_t)std::max(flat, 1)));
    }
    if (c->dump_neighbors) {
        HIPCHK(c, c->dbg_nn.ensure(sizeof(int) * 6 * (size_t)std::max(flat, 1)));
        HIPCHK(c, hipMemsetAsync(c->dbg_nn.p, 0xff, sizeof(int) * 6 * (size_t)std::max(flat, 1), c->stream));
    }
    HIPCHK(c, c->partials.ensure(sizeof(double) * kNumAcc * (size_t)std::max(c->n_blocks, 1)));
    HIPCHK(c, c->results.ensure(sizeof(float) * kResultSize * ((size_t)std::max(n_items, 1) + 1)));      // (+ one record: the run's "row_reach" miss count)
    if (c->trace_cap > 0) HIPCHK(c, c->trace.ensure(sizeof(float) * kTraceStride * (size_t)c->trace_cap * (size_t)std::max(n_items, 1)));
    if (c->mode_now == 3)                        // the mode may have been chosen after the targets were set
        for (int slot : c->batch_slots)
            for (int k = 0; k < 2; ++k) {
                Target& t = c->targets[(size_t)slot];
                if (!t.graph_valid[k] || !t.g[k].nbr) { rc = ensure_graph(c, t, k, true); if (rc) return rc; c->grids_dirty = true; }
            }
    if (c->mode_now == 5)
        for (int slot : c->batch_slots)
            for (int k = 0; k < 2; ++k) {
                Target& t = c->targets[(size_t)slot];
                if (!t.crow_valid[k] || !t.g[k].crow_tab) { rc = ensure_crows(c, t, k); if (rc) return rc; c->grids_dirty = true; }
            }
    for (int slot : c->batch_slots)            // query marks for the row build of every run (option "row_reach"): buffers + the pointer in the grid
        for (int k = 0; k < 2; ++k) {
            const unsigned* before = c->targets[(size_t)slot].g[k].qmark;
            rc = ensure_reach(c, c->targets[(size_t)slot], k, c->mode_now == 5 && c->rebuild_targets_each_run && c->row_reach != 0);
            if (rc) return rc;
            Target& t = c->targets[(size_t)slot];
            if (t.g[k].qmark && t.g[k].qmark != before)       // new words: zero once, every run hands them back clean
                HIPCHK(c, hipMemsetAsync(t.g[k].qmark, 0, sizeof(unsigned) * (size_t)t.g[k].nx * (size_t)t.g[k].ny * (size_t)t.g[k].qmark_w, c->stream));
        }
    if (c->grids_dirty) { rc = upload_grids(c); if (rc) return rc; }
    // table for rebuilding every target index of this batch in ONE launch sequence (rebuild_targets_each_run)
    c->h_tsegs.clear(); c->h_tblocks.clear(); c->h_tchunks.clear();
    int tflat = 0, tbucket = 0, tstrip = 0;
    c->t_max_units = 0; c->t_max_ucells = 0;
    bool strips_fit = true;
    for (int slot : c->batch_slots)
        for (int k = 0; k < 2; ++k) {
            Target& t = c->targets[(size_t)slot];
            TargetSeg ts;
            memset(&ts, 0, sizeof ts);
            ts.raw = t.raw_ptr[k]; ts.sorted_out = t.sorted[k].as<float4>(); ts.cell_start_out = t.cell_start[k].as<int>();
            ts.n = t.n[k]; ts.n_cells = t.n[k] > 0 ? t.n_cells[k] : 0;
            ts.flat_base = tflat; ts.bucket_base = tbucket;
            ts.ox = t.g[k].ox; ts.oy = t.g[k].oy; ts.oz = t.g[k].oz; ts.inv_cell = t.g[k].inv_cell;
            ts.nx = t.g[k].nx; ts.ny = t.g[k].ny; ts.nz = t.g[k].nz;
            ts.grid_id = slot * 2 + k;
            ts.strip_base = tstrip;
            // cells per strip: a batch of one or two targets (a shared submap: configs[1]) wants more, smaller strips than the 2048 cells that suit
            // a batch of hundreds — its strip build is a handful of workgroups (k_strip_build<true> 22 -> 15.6 us with 1024; configs[3] loses
            // 2.7 % with it): "index_strip_cells" = 0 (default) picks by the number of targets in the batch
            const int strip_cells = c->strip_cells > 0 ? c->strip_cells : (c->batch_slots.size() <= 2 ? 1024 : 2048);
            ts.ystrip = std::max(1, std::min(ts.ny, (strip_cells + ts.nz / 2) / std::max(ts.nz, 1)));
            // wide grids: fewer, longer strips so that nx * nstrips stays inside the partition histogram (kMaxStrips bins)
            if (ts.nx > 0 && ts.nx <= kMaxStrips) {
                const int max_nstrips = std::max(1, kMaxStrips / ts.nx);
                ts.ystrip = std::max(ts.ystrip, (ts.ny + max_nstrips - 1) / max_nstrips);
            }
            ts.nstrips = (ts.ny + ts.ystrip - 1) / ts.ystrip;
            const int id = (int)c->h_tsegs.size();
            c->h_tsegs.push_back(ts);
            for (int s = 0; s < ts.n; s += kBlockQ) c->h_tblocks.push_back(BlockDesc{ id, s, std::min(kBlockQ, ts.n - s), 0 });
            for (int s = 0; s < ts.n; s += kPartChunkHost) c->h_tchunks.push_back(BlockDesc{ id, s, std::min(kPartChunkHost, ts.n - s), 0 });
            tflat += ts.n; tbucket += ts.n_cells;
            if (ts.n > 0) {
                const int units = ts.nx * ts.nstrips;
                tstrip += units; c->t_max_units = std::max(c->t_max_units, units); c->t_max_ucells = std::max(c->t_max_ucells, ts.ystrip * ts.nz);
                strips_fit = strips_fit && units <= kMaxStrips;
            }
        }
    c->t_elems = tflat; c->t_buckets = std::max(tbucket, 1); c->t_strips = tstrip;
    strips_fit = strips_fit && ((size_t)c->t_max_ucells + 1) * 4 + (size_t)c->strip_cap * 6 + 8192 <= kStripLdsLarge;
    // auto = the strip form whenever the grids fit it: measured faster from 2 own-target items up (index 0.063 vs 0.078 ms) to 256
    // (2.2 vs 5.3 ms), and equal for the single shared submap of configs[1]
    c->strip_now = strips_fit && c->index_build != 0;
    if (c->index_build == 1 && !c->strip_now)
        return fail(c, LISREG_ERR_ARG, "index_build 1: a target grid of this batch does not fit the strip form (strips per target or cells per strip)");
    HIPCHK(c, c->tchunk_dev.ensure(sizeof(BlockDesc) * std::max<size_t>(c->h_tchunks.size(), 1)));
    { const void* before = c->strip_tab.p; HIPCHK(c, c->strip_tab.ensure(sizeof(int) * (3 * ((size_t)tstrip + 4) + (size_t)tstrip / 2048 + 8))); if (c->strip_tab.p != before) c->strip_zero_ints = 0; }
    HIPCHK(c, c->tseg_dev.ensure(sizeof(TargetSeg) * std::max<size_t>(c->h_tsegs.size(), 1)));
    HIPCHK(c, c->tblk_dev.ensure(sizeof(BlockDesc) * std::max<size_t>(c->h_tblocks.size(), 1)));
    // every table crosses PCIe from ONE pinned staging buffer: five asynchronous copies, no host synchronisation
    // (pageable sources would make each hipMemcpyAsync a blocking staged copy — most of a single registration's latency)
    {
        struct Part { const void* src; size_t bytes; void* dst; };
        const Part parts[7] = {
            { c->h_tchunks.data(), sizeof(BlockDesc) * c->h_tchunks.size(), c->tchunk_dev.p },
            { c->h_blocks_q.data(), sizeof(BlockDesc) * c->h_blocks_q.size(), c->blocks_q.p },
            { c->h_blocks.data(), sizeof(BlockDesc) * (size_t)c->n_blocks, c->blocks.p },
            { c->h_segs.data(), sizeof(Segment) * (size_t)c->n_segs, c->segs.p },
            { c->h_items.data(), sizeof(ItemState) * (size_t)n_items, c->items.p },
            { c->h_tsegs.data(), sizeof(TargetSeg) * c->h_tsegs.size(), c->tseg_dev.p },
            { c->h_tblocks.data(), sizeof(BlockDesc) * c->h_tblocks.size(), c->tblk_dev.p } };
        size_t total = 0;
        for (const Part& pt : parts) total += (pt.bytes + 63) & ~(size_t)63;
        if (c->stage_done) HIPCHK(c, hipEventSynchronize(c->stage_done));      // the previous batch's copies have left the buffer
        else HIPCHK(c, hipEventCreateWithFlags(&c->stage_done, hipEventDisableTiming));
        if (total > c->stage_cap) {
            if (c->stage_host) (void)hipHostFree(c->stage_host);
            c->stage_host = nullptr; c->stage_cap = 0;
            HIPCHK(c, hipHostMalloc((void**)&c->stage_host, total + total / 2 + 4096, hipHostMallocDefault));
            c->stage_cap = total + total / 2 + 4096;
        }
        size_t off = 0;
        for (const Part& pt : parts) {
            if (pt.bytes) {
                memcpy(c->stage_host + off, pt.src, pt.bytes);
                HIPCHK(c, hipMemcpyAsync(pt.dst, c->stage_host + off, pt.bytes, hipMemcpyHostToDevice, c->stream));
            }
            off += (pt.bytes + 63) & ~(size_t)63;
        }
        HIPCHK(c, hipEventRecord(c->stage_done, c->stream));
    }
    // sort_sources: scan order and voxel-grid order are spatially coherent and beat a re-sort; an arbitrary order costs
    // the cell walk its L1 locality (2x slower), so in auto mode a cheap probe decides once per prepared batch.  Small batches
    // (a single odometry frame) skip the probe and its host round trip: a sort could not pay for itself there.
    c->sort_now = c->sort_sources == 1;
    // (the probe costs a pass over the sources and a host round trip — 0.15 ms in front of a pipelined batch: batches of the same shape as
    // the one last probed reuse its verdict, re-probed every 32nd; the verdict decides speed only, never results)
    if (c->sort_sources == 2 && c->n_elems >= 65536 && c->probe_items == n_items && c->probe_elems == c->n_elems && c->probe_age < 32) {
        ++c->probe_age;
        c->sort_now = c->probe_verdict;
    } else
    if (c->sort_sources == 2 && c->n_elems >= 65536) {
        launch_count_jumps(c->blocks.as<BlockDesc>(), c->n_blocks, c->segs.as<Segment>(), 1.5f, c->done_dev.as<int>(), c->stream);
        int jumps_stack = 0;
        int* jumps = c->done_host ? c->done_host : &jumps_stack;       // lisreg_create tolerates a failed pinned allocation
        HIPCHK(c, hipMemcpyAsync(jumps, c->done_dev.p, sizeof(int), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        c->sort_now = (double)*jumps > 0.25 * (double)c->n_elems;
        c->probe_verdict = c->sort_now; c->probe_items = n_items; c->probe_elems = c->n_elems; c->probe_age = 0;
    }
    // scratch of the bucket sorts: the target rebuild (if the batch does that) and the source sort (only if it will run)
    {
        const size_t se = (size_t)std::max(c->rebuild_targets_each_run ? tflat : 1, c->sort_now ? flat : 1);
        const size_t sbk = (size_t)std::max(c->rebuild_targets_each_run ? c->t_buckets : 1, c->sort_now ? c->n_buckets : 1);
        rc = ensure_sort_scratch(c, se, sbk);
        if (rc) return rc;
    }
    // The registrations start reset (pose caches of the initial poses, done counter = the guard failures): every run leaves them reset again
    // for the next one (launch_finalize), so a run has no reset launch of its own.  The "row_reach" miss count is cumulative from here.
    launch_reset_items(c->items.as<ItemState>(), c->n_items, c->prm, c->done_dev.as<int>(), c->stream,
                       reinterpret_cast<int*>(c->results.as<float>() + (size_t)std::max(c->n_items, 1) * kResultSize));
    c->items_reset = true; c->reach_miss_seen = 0; c->runs_since_fetch = 0;
    // Query marks for the row builds of this batch's runs (option "row_reach"; cell rows, targets rebuilt inside every run).  A scan sees a
    // part of its map — the lidar's elevation span leaves the upper walls of the benchmark room unseen: a fifth of the cells that would
    // get rows — so every run builds rows only for the cells a query comes within a metre of under its INITIAL pose: one pass over the
    // batch's source points (k_query_marks: the cell each falls into), those marks grown by ceil(1 m / cell) cells (`reach`, read by the
    // classification of every run), the mark words handed back clean.  They depend on the sources and the initial poses — what this call
    // is given — not on the target's points, which a run may find changed.
    c->reach_ready = false;
    if (c->reach_backoff > 0) --c->reach_backoff;
    else if (c->mode_now == 5 && c->rebuild_targets_each_run && c->row_reach != 0 && c->lanes_q == 1 && c->n_blocks > 0) {
        bool any = false;
        for (int slot : c->batch_slots) for (int k = 0; k < 2; ++k) any = any || (c->targets[(size_t)slot].g[k].qmark != nullptr && c->targets[(size_t)slot].n[k] > 0);
        if (any) {
            launch_query_marks(c->blocks.as<BlockDesc>(), c->n_blocks, c->segs.as<Segment>(), c->grids_dev.as<GridIndex>(), c->items.as<ItemState>(), c->stream);
            for (int slot : c->batch_slots)
                for (int k = 0; k < 2; ++k) {
                    Target& t = c->targets[(size_t)slot];
                    if (!t.g[k].qmark || t.n[k] <= 0) continue;
                    launch_reach_dilate(t.g[k], t.crow_reach[k].as<unsigned>(), c->stream);
                    HIPCHK(c, hipMemsetAsync(t.g[k].qmark, 0, sizeof(unsigned) * (size_t)t.g[k].nx * (size_t)t.g[k].ny * (size_t)t.g[k].qmark_w, c->stream));
                }
            HIPCHK(c, hipGetLastError());
            c->reach_ready = true;
        }
    }
    // The dispatch order of the runs' correspondence launches (blocks ranked by the sector their queries fall into, one eighth per XCD) is a
    // function of the same inputs — block descriptors, sources, initial poses, grid geometry: made here once instead of by every run
    // (k_xcd_keys + a single-workgroup counting sort: 70 us per run on the side stream, which the shorter row build of "row_reach" had turned
    // into the tail of a run's index phase).  Runs that sort their sources or split the batch in halves keep making their own.
    c->xcd_cached = false;
    if (xcd_order_wanted(c) && !c->sort_now && !c->exact && c->n_blocks > 0) {
        HIPCHK(c, c->xcd_tab.ensure(sizeof(int) * 2 * (size_t)c->n_blocks));
        launch_xcd_order(c->blocks.as<BlockDesc>(), c->n_blocks, c->segs.as<Segment>(), c->grids_dev.as<GridIndex>(), c->items.as<ItemState>(),
                         nullptr, c->batch_slots.size() >= 8 && c->xcd_order == 1, c->xcd_tab.as<int>(), c->xcd_tab.as<int>() + c->n_blocks, c->stream);
        HIPCHK(c, hipGetLastError());
        c->xcd_cached = true;
    }
    c->prepared = true;
    return LISREG_OK;
}

// exact_arithmetic: the sine and cosine of the three pose angles are taken by the HOST's libm — the library the reference's
// pcl::getTransformation and LMOptimization call (cosf / sinf on x86-64) — because no device routine returns its last bit in every case
// (1 configuration in 100 showed a 1-ulp matrix entry, which swapped two candidates 8e-7 apart in squared distance).  One 24-byte-per-item
// round trip per Gauss-Newton iteration; only this build pays it.
static int exact_pose_caches(lisreg_ctx* c)
{
    const size_t n = (size_t)c->n_items;
    if (n == 0) return LISREG_OK;
    HIPCHK(c, c->exact_trig.ensure(sizeof(float) * 6 * n));
    std::vector<float> T(6 * n), trig(6 * n);
    launch_pose_gather(c->items.as<ItemState>(), c->n_items, c->exact_trig.as<float>(), c->stream);
    HIPCHK(c, hipMemcpyAsync(T.data(), c->exact_trig.p, sizeof(float) * 6 * n, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    for (size_t i = 0; i < n; ++i) {
        const float* t = &T[6 * i];
        float* g = &trig[6 * i];
        g[0] = cosf(t[2]); g[1] = sinf(t[2]); g[2] = cosf(t[1]); g[3] = sinf(t[1]); g[4] = cosf(t[0]); g[5] = sinf(t[0]);
    }
    HIPCHK(c, hipMemcpyAsync(c->exact_trig.p, trig.data(), sizeof(float) * 6 * n, hipMemcpyHostToDevice, c->stream));
    launch_pose_cache_from_trig(c->items.as<ItemState>(), c->n_items, c->exact_trig.as<float>(), c->stream);
    HIPCHK(c, hipStreamSynchronize(c->stream));          // `trig` is a local
    return LISREG_OK;
}

// XCD-aware dispatch order for the prepared batch? (see run_impl)
static bool xcd_order_wanted(const lisreg_ctx* c)
{
    const bool many_targets = c->batch_slots.size() >= 8 && c->xcd_order == 1;
    return c->lanes_q != 8 && c->mode_now != 0 &&
           (((c->mode_now == 3 || c->mode_now == 5) && (c->xcd_order == 1 || (c->xcd_order == 2 && c->n_blocks >= 2048 && c->n_items >= 32))) ||
            (many_targets && c->xcd_order != 0 && c->n_blocks >= 2048));
}

// Enqueue one full pass.  early_stop (synchronous entry points only): after every few iterations the host reads a
// 4-byte "registrations finished" counter and stops launching once every item has converged — the reference's
// `break` at :617 — instead of launching no-op kernels up to max_iters.  Results are identical either way.
static int run_impl(lisreg_ctx* c, bool early_stop)
{
    if (!c->prepared) return fail(c, LISREG_ERR_ARG, "batch_run: no prepared batch");
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t st = c->stream;
    // XCD-aware dispatch order (lisreg_assoc.hip, launch_xcd_order): two small launches per run, at the initial poses.  Auto: the graph
    // front-end with >= 32 registrations (measured: +2.5 % at 64 scans, +4.9 % at 256; with 8 big scans the sectors are unevenly loaded
    // and it costs 2 %; the walk front-end gains nothing)
    // Round 5: a batch whose registrations bring targets of their own (configs[3]: 256 candidate pairs, 256 x 6 MB of index) takes the order
    // by TARGET whatever the front-end — whole targets per XCD, so that an L2 holds the one or two targets its CUs are working on instead of
    // a slice of all eight-plus in flight.
    // (measured on configs[3], 256 own-target registrations through the cell walk: 13 539 reg/s against 13 716 in plain order — no gain, the walk
    //  is not bound by L2 misses either; the order by target is taken only with xcd_order = 1, profiles/r05_kernel_experiments.md)
    const bool many_targets = c->batch_slots.size() >= 8 && c->xcd_order == 1;
    c->xcd_now = xcd_order_wanted(c);
    if (c->xcd_now) HIPCHK(c, c->xcd_tab.ensure(sizeof(int) * 2 * (size_t)c->n_blocks));
    // per-run reset of the registrations and the dispatch order: both depend on the batch only (items, initial poses, grid geometry), not on
    // the rebuilt index — with the cell rows they ride on the side stream behind the corner target's rows, underneath the surf target's
    // (25-30 us of two small launches and a single-workgroup counting sort off the critical path of a configs[1] step)
    // Two halves of the batch on two streams (round 5).  The 6x6 solves are one workgroup per registration and ~13 us of dependent latency
    // between two correspondence launches: the chip idles through ten of them per step.  Registrations are independent, so the batch is cut
    // in two at an item boundary near the middle of the workgroups: the first half iterates on the context's stream, the second on the side
    // stream, and each half's solve (and launch gaps, and the thinning tail of its correspondence launch) runs underneath the other half's
    // correspondence launch.  Same kernels on the same data in the same order per registration: same results to the bit.  Only for runs
    // that do not stop early from the host (fixed iteration counts, lisreg_batch_run) and are big enough to fill the chip twice.
    const bool can_stop = early_stop && c->prm.fixed_iters <= 0 && c->done_host && c->early_stop_chunk != 0;
    int split_item = 0, split_blk = 0;
    if (c->interleave != 0 && !c->exact && !can_stop && c->lanes_q == 1 && c->mode_now != 0 && c->n_items >= 2 && c->n_blocks >= c->interleave_min_blocks) {
        if (!c->side_stream) {
            if (hipStreamCreateWithFlags(&c->side_stream, hipStreamNonBlocking) != hipSuccess) c->side_stream = nullptr;
            if (c->side_stream && (hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming) != hipSuccess ||
                                   hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming) != hipSuccess)) {
                (void)hipStreamDestroy(c->side_stream); c->side_stream = nullptr;
            }
        }
        if (c->side_stream && !c->ev_ab && (hipEventCreateWithFlags(&c->ev_ab, hipEventDisableTiming) != hipSuccess ||
                                            hipEventCreateWithFlags(&c->ev_ba, hipEventDisableTiming) != hipSuccess)) { c->ev_ab = nullptr; c->ev_ba = nullptr; }
        if (c->side_stream) {
            for (int i = 1; i < c->n_items; ++i)
                if (c->h_items[(size_t)i].blk_begin * 2 >= c->n_blocks) { split_item = i; split_blk = c->h_items[(size_t)i].blk_begin; break; }
            if (split_blk * 4 < c->n_blocks || (c->n_blocks - split_blk) * 4 < c->n_blocks) { split_item = 0; split_blk = 0; }     // (a lop-sided cut hides nothing)
        }
    }
    // Round 6, option "row_reach": the rows of this run are built only for the cells a query of the batch comes within a metre of under
    // its initial pose (`reach` words made by lisreg_batch_prepare: they depend on the sources and the initial poses alone, not on the target's
    // points).  A query that ends up in a cell without rows all the same takes the cell walk: results cannot depend on the marks.
    c->reach_now = c->rebuild_targets_each_run && c->mode_now == 5 && c->reach_ready && !c->exact;
    int* const miss_dev = reinterpret_cast<int*>(c->results.as<float>() + (size_t)std::max(c->n_items, 1) * kResultSize);
    c->prm.reach_miss = c->reach_now ? miss_dev : nullptr;
    bool reset_done = false, order_done = false;
    auto dispatch_order = [&](hipStream_t s_) {
        order_done = true;
        if (c->xcd_now && !(c->xcd_cached && split_blk == 0)) {
            // (an interleaved run dispatches its halves separately: one table per half, positions and ids relative to the half)
            const int nb0 = split_blk ? split_blk : c->n_blocks;
            launch_xcd_order(c->blocks.as<BlockDesc>(), nb0, c->segs.as<Segment>(), c->grids_dev.as<GridIndex>(), c->items.as<ItemState>(),
                             c->sort_now ? c->sorted_all.as<float4>() : nullptr, many_targets, c->xcd_tab.as<int>(), c->xcd_tab.as<int>() + c->n_blocks, s_);
            if (split_blk)
                launch_xcd_order(c->blocks.as<BlockDesc>() + split_blk, c->n_blocks - split_blk, c->segs.as<Segment>(), c->grids_dev.as<GridIndex>(),
                                 c->items.as<ItemState>(), c->sort_now ? c->sorted_all.as<float4>() : nullptr, many_targets,
                                 c->xcd_tab.as<int>() + split_blk, c->xcd_tab.as<int>() + c->n_blocks + split_blk, s_);
        }
    };
    // (the registrations are reset already — by lisreg_batch_prepare or by the run before, launch_finalize — unless a run ended in an error)
    if (c->items_reset) reset_done = true;
    c->items_reset = false;
    auto reset_and_order = [&](hipStream_t s_) {
        if (!reset_done) launch_reset_items(c->items.as<ItemState>(), c->n_items, c->prm, c->done_dev.as<int>(), s_);
        dispatch_order(s_);
        reset_done = true;
    };
    if (c->rebuild_targets_each_run) {                 // the reference rebuilds both kd-trees per registration (:602-603)
        prof_mark(c, 2);
        if (c->strip_now) {
            StripBuffers sl;
            sl.cnt = c->strip_tab.as<int>(); sl.fill = sl.cnt + (c->t_strips + 1); sl.start = sl.fill + (c->t_strips + 1);
            sl.scan_tmp = sl.start + (c->t_strips + 2);
            sl.tmp_pts = c->tmp_pts.as<float4>(); sl.slot_idx = c->elem_bucket.as<uint32_t>(); sl.slot_pos = c->elem_sub.as<uint32_t>();
            if (!c->side_stream) {                    // created once; failure just means the two variants run back to back
                if (hipStreamCreateWithFlags(&c->side_stream, hipStreamNonBlocking) != hipSuccess) c->side_stream = nullptr;
                if (c->side_stream && (hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming) != hipSuccess ||
                                       hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming) != hipSuccess)) {
                    (void)hipStreamDestroy(c->side_stream); c->side_stream = nullptr;
                }
            }
            sl.side = c->side_stream; sl.ev_fork = c->ev_fork; sl.ev_join = c->ev_join;
            if (launch_build_targets_strips(c->tchunk_dev.as<BlockDesc>(), (int)c->h_tchunks.size(), c->tseg_dev.as<TargetSeg>(),
                                            (int)c->h_tsegs.size(), c->t_strips, c->t_max_units, c->t_max_ucells, c->strip_cap, sl, st, &c->strip_zero_ints))
                return fail(c, LISREG_ERR_HIP, "strip index build: LDS configuration refused");
        } else
            launch_build_targets_batched(c->tblk_dev.as<BlockDesc>(), (int)c->h_tblocks.size(), c->tseg_dev.as<TargetSeg>(),
                                         (int)c->h_tsegs.size(), c->t_elems, c->t_buckets, sort_buffers(c), st);
        if (c->mode_now == 3)
            launch_build_graph(c->tblk_dev.as<BlockDesc>(), (int)c->h_tblocks.size(), c->tseg_dev.as<TargetSeg>(),
                               c->grids_dev.as<GridIndex>(), st);
        static const bool crow_pair = !(getenv("LISREG_CROW_PAIR") && atoi(getenv("LISREG_CROW_PAIR")) == 0);      // A/B: 0 = rounds 4-5's side stream
        if (c->mode_now == 5 && crow_pair) {
            // corner and surf target of a slot in ONE launch sequence on this stream (launch_crow_rows_pair): no side stream, no event hops
            for (int slot : c->batch_slots) {
                Target& t = c->targets[(size_t)slot];
                const GridIndex g2[2] = { t.g[0], t.g[1] };
                const int nc2[2] = { t.n_cells[0], t.n_cells[1] };
                const lisreg::CrowBuffers cb2[2] = { crow_buffers(t, 0, c->reach_now), crow_buffers(t, 1, c->reach_now) };
                int* oz2[2] = { &t.omask_zero[0], &t.omask_zero[1] };
                launch_crow_rows_pair(g2, nc2, cb2, st, oz2);
            }
        } else
        if (c->mode_now == 5) {
            // (LISREG_CROW_PAIR=0) the rows of the corner targets (a few ten thousand rows: launches that leave most of the chip idle) are built on the side stream,
            // underneath the surf targets' — separate buffers per target kind, joined before the first correspondence launch
            if (!c->side_stream) {
                if (hipStreamCreateWithFlags(&c->side_stream, hipStreamNonBlocking) != hipSuccess) c->side_stream = nullptr;
                if (c->side_stream && (hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming) != hipSuccess ||
                                       hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming) != hipSuccess)) {
                    (void)hipStreamDestroy(c->side_stream); c->side_stream = nullptr;
                }
            }
            const bool fork = c->side_stream && hipEventRecord(c->ev_fork, st) == hipSuccess &&
                              hipStreamWaitEvent(c->side_stream, c->ev_fork, 0) == hipSuccess;
            for (int k = 0; k < 2; ++k) {
                hipStream_t sk = (k == 0 && fork) ? c->side_stream : st;
                for (int slot : c->batch_slots) {
                    Target& t = c->targets[(size_t)slot];
                    if (t.n[k] <= 0) continue;
                    launch_crow_classify(t.g[k], t.n_cells[k], crow_buffers(t, k, c->reach_now), sk, &t.omask_zero[k]);
                    launch_crow_build(t.g[k], t.n_cells[k], crow_buffers(t, k, c->reach_now), sk, &t.omask_zero[k]);
                }
            }
            if (fork && !c->sort_now && !c->exact) reset_and_order(c->side_stream);
            if (fork) { (void)hipEventRecord(c->ev_join, c->side_stream); (void)hipStreamWaitEvent(st, c->ev_join, 0); }
        }
        prof_mark(c, -1);
    }
    if (!reset_done) launch_reset_items(c->items.as<ItemState>(), c->n_items, c->prm, c->done_dev.as<int>(), st);
    if (c->exact) { int rc = exact_pose_caches(c); if (rc) return rc; }
    prof_mark(c, 2);
    launch_sort_sources(c->blocks.as<BlockDesc>(), c->n_blocks, c->segs.as<Segment>(), c->n_segs, c->items.as<ItemState>(),
                        c->n_elems, c->sort_now ? c->n_buckets : 0, sort_buffers(c), c->sorted_all.as<float4>(), c->order_all.as<int>(), st);
    prof_mark(c, -1);
    if (!order_done) dispatch_order(st);                   // (after the source sort: the keys read the sorted records)
    // how often the host looks at the "registrations finished" counter: a skipped launch of a big batch still dispatches tens of
    // thousands of workgroups (check every 3 iterations), a skipped launch of a single frame costs ~2 us (check every 6: one
    // round trip for the typical 3-6 iteration registration)
    const int chunk = c->early_stop_chunk > 0 ? c->early_stop_chunk : (c->n_blocks <= 1024 ? 6 : 3);
    // sequential use: the last batch fetched from this context needed `last_launches` iterations — frames of a drive converge alike, so the
    // first look is taken right there (a replay frame converges in 3: three no-op iterations, nine launches, not enqueued), later looks every 3
    int next_check = chunk;
    if (c->early_stop_chunk <= 0 && c->last_launches > 0) next_check = std::max(2, std::min(c->last_launches, chunk));
    c->prm.cell_anchor_until = c->cell_anchor_until;
    // one Gauss-Newton iteration of the items [i0, i0 + ni) = the workgroups [b0, b0 + nb) on stream s_ (sidx: which stream the profiling marks go on)
    // (after_assoc: an event recorded right behind the correspondence launch, for the alternation of an interleaved run)
    auto iteration = [&](int it, int i0, int ni, int b0, int nb, hipStream_t s_, int sidx, hipEvent_t after_assoc) {
        if (!after_assoc) prof_mark(c, 0, sidx);          // (an alternating run marks behind its wait for the other half)
        (c->exact ? launch_assoc_exact : launch_assoc)(
                     c->blocks.as<BlockDesc>() + b0, nb, c->segs.as<Segment>(), c->grids_dev.as<GridIndex>(),
                     c->items.as<ItemState>(), c->prm, c->sort_now ? c->sorted_all.as<float4>() : nullptr, c->partials.as<double>() + (size_t)b0 * kNumAcc,
                     c->mode_now, c->nn.as<int>(), c->n_elems, c->first_pass_r * c->first_pass_r,
                     it >= (c->lanes_q == 8 ? c->wide_from_small : c->wide_from) && it <= (c->mode_now == 5 ? c->crow_wide_until : (c->mode_now == 3 ? c->graph_wide_until : c->wide_until)), c->graph_hops,
                     c->count_searches ? c->counters.as<unsigned long long>() : nullptr,
                     c->dump_neighbors ? c->dbg_nn.as<int>() : nullptr, c->lanes_q,
                     c->blocks_q.as<BlockDesc>(), (int)c->h_blocks_q.size(), c->coef.as<float4>(), c->coef_ok.as<int>(),
                     c->xcd_now ? c->xcd_tab.as<int>() + c->n_blocks + b0 : nullptr, s_);
        if (after_assoc) (void)hipEventRecord(after_assoc, s_);
        prof_mark(c, 1, sidx);
        launch_solve(c->items.as<ItemState>() + i0, ni, c->prm, c->partials.as<double>(),
                     c->trace_cap > 0 ? c->trace.as<float>() + (size_t)i0 * (size_t)c->trace_cap * kTraceStride : nullptr, c->trace_cap, c->done_dev.as<int>(), s_);
        prof_mark(c, -1, sidx);
    };
    c->interleaved_now = false;
    if (split_blk > 0 && hipEventRecord(c->ev_fork, st) == hipSuccess && hipStreamWaitEvent(c->side_stream, c->ev_fork, 0) == hipSuccess) {
        c->interleaved_now = true;
        // interleave 1: the two correspondence launches ALTERNATE (each waits for the other half's previous one through an event), so that
        // every launch has the chip to itself apart from the other half's solve — per-launch durations stay what they are for a launch
        // alone, which is what the roofline figures of bench.py and the profiler tables are made of; 2: free-running (the launches of the
        // two streams share the chip whenever both are ready: same throughput, launch durations no longer comparable)
        const bool alternate = c->interleave == 1 && c->ev_ab && c->ev_ba;
        // a failing event call must not leave work of the second half un-joined on the side stream (the caller's next fetch on `st` would race
        // with it): on any failure the side stream is drained by the HOST before the error is returned, and the run counts as not interleaved
        hipError_t ie = hipSuccess;
        for (int it = 0; it < c->prm.bound && ie == hipSuccess; ++it) {
            if (alternate && it > 0) ie = hipStreamWaitEvent(st, c->ev_ba, 0);                   // B's launch of the iteration before is through
            if (ie != hipSuccess) break;
            if (alternate) prof_mark(c, 0, 0);
            iteration(it, 0, split_item, 0, split_blk, st, 0, alternate ? c->ev_ab : nullptr);
            if (alternate) ie = hipStreamWaitEvent(c->side_stream, c->ev_ab, 0);                 // A's launch of this iteration is through
            if (ie != hipSuccess) break;
            if (alternate) prof_mark(c, 0, 1);
            iteration(it, split_item, c->n_items - split_item, split_blk, c->n_blocks - split_blk, c->side_stream, 1, alternate ? c->ev_ba : nullptr);
        }
        if (ie == hipSuccess) ie = hipEventRecord(c->ev_join, c->side_stream);
        if (ie == hipSuccess) ie = hipStreamWaitEvent(st, c->ev_join, 0);
        if (ie != hipSuccess) {
            (void)hipStreamSynchronize(c->side_stream);
            c->interleaved_now = false;
            return lisreg::ctx_fail(c, LISREG_ERR_HIP, std::string("interleaved run: ") + hipGetErrorString(ie));
        }
    } else
    for (int it = 0; it < c->prm.bound; ++it) {
        iteration(it, 0, c->n_items, 0, c->n_blocks, st, 0, nullptr);
        if (c->exact && it + 1 < c->prm.bound) { int rc = exact_pose_caches(c); if (rc) return rc; }
        if (can_stop && it + 1 == next_check && it + 1 < c->prm.bound) {
            HIPCHK(c, hipMemcpyAsync(c->done_host, c->done_dev.p, sizeof(int), hipMemcpyDeviceToHost, st));
            HIPCHK(c, hipStreamSynchronize(st));
            if (*c->done_host >= c->n_items) break;
            next_check += c->early_stop_chunk > 0 ? chunk : std::min(chunk, 3);
        }
    }
    launch_finalize(c->items.as<ItemState>(), c->n_items, c->prm, c->results.as<float>(), st, c->done_dev.as<int>());      // (+ reset for the next run)
    c->items_reset = true; ++c->runs_since_fetch;
    HIPCHK(c, hipGetLastError());
    if (c->pack_in_use >= 0 && c->pack_free[c->pack_in_use])      // the staged source buffer may be overwritten once this run is through
        HIPCHK(c, hipEventRecord(c->pack_free[c->pack_in_use], st));
    return LISREG_OK;
}

int lisreg_batch_run(lisreg_ctx* c)
{
    if (!c) return LISREG_ERR_ARG;
    return run_impl(c, false);
}

int lisreg_batch_fetch(lisreg_ctx* c, float* T, lisreg_stats* stats)
{
    if (!c) return LISREG_ERR_ARG;
    if (!c->prepared) return fail(c, LISREG_ERR_ARG, "batch_fetch: no prepared batch");
    HIPCHK(c, hipSetDevice(c->device));
    // results (and, for lisreg_align, the trace) land in pinned memory: asynchronous copies, ONE synchronisation
    const size_t res_floats = ((size_t)std::max(c->n_items, 1) + 1) * kResultSize;
    const size_t trace_floats = c->fetch_trace_records > 0 ? (size_t)kTraceStride * (size_t)c->fetch_trace_records : 0;
    if ((res_floats + trace_floats) * sizeof(float) > c->fetch_cap) {
        if (c->fetch_host) (void)hipHostFree(c->fetch_host);
        c->fetch_host = nullptr; c->fetch_cap = 0;
        const size_t want = (res_floats + trace_floats) * sizeof(float) * 2 + 4096;
        HIPCHK(c, hipHostMalloc((void**)&c->fetch_host, want, hipHostMallocDefault));
        c->fetch_cap = want;
    }
    const bool with_miss = c->n_items > 0 && c->reach_now;
    if (c->n_items) HIPCHK(c, hipMemcpyAsync(c->fetch_host, c->results.p, sizeof(float) * kResultSize * ((size_t)c->n_items + (with_miss ? 1 : 0)), hipMemcpyDeviceToHost, c->stream));
    if (trace_floats && c->trace.p) HIPCHK(c, hipMemcpyAsync(c->fetch_host + res_floats, c->trace.p, sizeof(float) * trace_floats, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->h_results.assign(c->fetch_host, c->fetch_host + res_floats);
    if (with_miss) {
        // "row_reach" watches itself: the marks are the queries' cells under their INITIAL poses grown by a metre; a batch whose first steps move
        // its points farther than that sends queries into cells without rows, where they walk (exact, but slow: configs[4] with a 0.5 m
        // dilation ran 30 % longer).  More than one query-iteration in a thousand there: the prepared batch's later runs, and the next 32
        // batches prepared on this context, build all rows.
        int cum; memcpy(&cum, c->fetch_host + (size_t)c->n_items * kResultSize, sizeof cum);      // (cumulative since the batch was prepared)
        const int miss = cum - c->reach_miss_seen;
        c->reach_miss_seen = cum;
        c->reach_miss_last = miss;
        if ((long long)miss * 1000LL > (long long)c->n_elems * (long long)std::max(c->prm.bound, 1) * (long long)std::max(c->runs_since_fetch, 1)) { c->reach_ready = false; c->reach_backoff = 32; }
    }
    if (!c->ev.empty()) prof_collect(c);          // events of every profiled run since the last fetch (profiling may be off again by now)
    c->last_launches = 0;
    for (int i = 0; i < c->n_items; ++i) {
        const float* r = &c->h_results[(size_t)i * kResultSize];
        c->last_launches = std::max(c->last_launches, (int)r[6] + 1);
        if (T) memcpy(T + 6 * (size_t)i, r, 24);
        if (stats) {
            stats[i].iters = (int)r[6]; stats[i].deltaR = r[7]; stats[i].deltaT = r[8];
            stats[i].degenerate = (int)r[9]; stats[i].n_corr_last = (int)r[10]; stats[i].status = (int)r[11];
        }
    }
    c->runs_since_fetch = 0;
    return LISREG_OK;
}

void* lisreg_batch_result_device(const lisreg_ctx* c) { return c ? c->results.p : nullptr; }

// Options that are not part of the reference's parameter surface (kept out of lisreg_params on purpose).
int lisreg_set_option(lisreg_ctx* c, const char* name, int value)
{
    if (!c || !name) return LISREG_ERR_ARG;
    if (!strcmp(name, "rebuild_targets_each_run")) {
        c->rebuild_targets_each_run = value != 0;
        if (c->prepared && c->rebuild_targets_each_run) {      // the rebuild needs its sort scratch
            int rc = ensure_sort_scratch(c, (size_t)std::max(c->t_elems, 1), (size_t)std::max(c->t_buckets, 1));
            if (rc) return rc;
        }
        return LISREG_OK;
    }
    if (!strcmp(name, "sort_sources")) { c->sort_sources = value; c->probe_items = -1; return LISREG_OK; }
    if (!strcmp(name, "cell_anchor_until")) { c->cell_anchor_until = std::max(value, 0); return LISREG_OK; }
    if (!strcmp(name, "search_mode")) {
        if (value < 0 || value > 5 || value == 2) return fail(c, LISREG_ERR_ARG, "search_mode: 0 LDS-staged box, 1 cell walk, 3 k-NN graph scan, 4 auto, 5 cell rows");
        c->search_mode = value; c->prepared = false; return LISREG_OK;
    }
    if (!strcmp(name, "graph_min_ratio")) { c->graph_min_ratio = value; c->prepared = false; return LISREG_OK; }
    if (!strcmp(name, "cell_min_ratio")) { c->cell_min_ratio = value; c->prepared = false; return LISREG_OK; }
    if (!strcmp(name, "row_reach")) { c->row_reach = value; c->reach_backoff = 0; c->prepared = false; return LISREG_OK; }
    if (!strcmp(name, "cell_rows_max_mb")) {
        c->cell_rows_max_mb = value; c->prepared = false;
        for (auto& t : c->targets) for (int k = 0; k < 2; ++k) {        // a new bound: what was too big may fit now, what fitted may have to be capped
            t.crow_too_big[k] = false;
            if (t.crow_valid[k]) { t.crow_valid[k] = false; c->grids_dirty = true; }
        }
        return LISREG_OK;
    }
    if (!strcmp(name, "early_stop_chunk")) { c->early_stop_chunk = value; return LISREG_OK; }
    if (!strcmp(name, "xcd_order")) { if (value < 0 || value > 2) return fail(c, LISREG_ERR_ARG, "xcd_order: 0 off, 1 on, 2 auto"); c->xcd_order = value; c->xcd_cached = false; return LISREG_OK; }
    if (!strcmp(name, "index_build")) {
        if (value < 0 || value > 2) return fail(c, LISREG_ERR_ARG, "index_build: 0 bucket sort, 1 strip form, 2 auto");
        c->index_build = value; c->prepared = false;
        return LISREG_OK;
    }
    if (!strcmp(name, "index_strip_cells")) { c->strip_cells = std::max(value, 0); c->prepared = false; return LISREG_OK; }
    if (!strcmp(name, "index_strip_cap")) { c->strip_cap = std::min(std::max(value, 64), 16384); c->prepared = false; return LISREG_OK; }
    if (!strcmp(name, "count_searches")) {
        c->count_searches = value != 0;
        if (c->count_searches) { HIPCHK(c, c->counters.ensure(128 * 8)); HIPCHK(c, hipMemset(c->counters.p, 0, 128 * 8)); }
        return LISREG_OK;
    }
    if (!strcmp(name, "lanes_per_query")) { c->lanes_per_query_auto = value != 1; c->prepared = false; return LISREG_OK; }   // 1 forces one lane per query, anything else = auto
    if (!strcmp(name, "dump_neighbors")) { c->dump_neighbors = value != 0; c->prepared = false; return LISREG_OK; }
    if (!strcmp(name, "exact_arithmetic")) { c->exact = value != 0; c->prepared = false; return LISREG_OK; }
    if (!strcmp(name, "feeder_copy_engine")) { c->feeder_engine = value; return LISREG_OK; }
    if (!strcmp(name, "canonical_ties")) { c->canonical_ties = value != 0; c->prepared = false; return LISREG_OK; }
    if (!strcmp(name, "first_pass_mm")) { c->first_pass_r = 1e-3f * (float)value; return LISREG_OK; }
    if (!strcmp(name, "trace_cap")) { c->trace_cap = std::max(0, value); c->prepared = false; return LISREG_OK; }
    if (!strcmp(name, "feeder_threads")) { c->feeder_threads = std::min(std::max(value, 0), 64); return LISREG_OK; }
    if (!strcmp(name, "feeder_numa")) { c->feeder_numa = value != 0; return LISREG_OK; }       // takes effect when the thread pool is created
    if (!strcmp(name, "interleave_min_blocks")) { c->interleave_min_blocks = std::max(value, 2); return LISREG_OK; }
    if (!strcmp(name, "interleave")) { if (value < 0 || value > 2) return fail(c, LISREG_ERR_ARG, "interleave: 0 off, 1 alternating halves, 2 free-running halves"); c->interleave = value; return LISREG_OK; }
    return fail(c, LISREG_ERR_ARG, std::string("set_option: unknown option ") + name);
}

int lisreg_get_option(const lisreg_ctx* c, const char* name, int* value)
{
    if (!c || !name || !value) return LISREG_ERR_ARG;
    if (!strcmp(name, "search_mode")) { *value = c->search_mode; return LISREG_OK; }
    if (!strcmp(name, "exact_arithmetic")) { *value = c->exact ? 1 : 0; return LISREG_OK; }
    if (!strcmp(name, "feeder_chunks_by_copy_engine")) { *value = c->pack_stolen; return LISREG_OK; }
    if (!strcmp(name, "feeder_chunks")) { *value = c->pack_chunks_n; return LISREG_OK; }
    if (!strcmp(name, "canonical_ties")) { *value = (c->canonical_ties || c->exact) ? 1 : 0; return LISREG_OK; }
    if (!strcmp(name, "index_build")) { *value = c->index_build; return LISREG_OK; }
    if (!strcmp(name, "index_build_now")) { *value = c->strip_now ? 1 : 0; return LISREG_OK; }
    if (!strcmp(name, "front_end")) { *value = c->mode_now; return LISREG_OK; }
    if (!strcmp(name, "lanes_per_query")) { *value = c->lanes_q; return LISREG_OK; }          // what the prepared batch runs (auto resolved)
    if (!strcmp(name, "sort_sources")) { *value = c->sort_sources; return LISREG_OK; }
    if (!strcmp(name, "cell_anchor_until")) { *value = c->cell_anchor_until; return LISREG_OK; }
    if (!strcmp(name, "feeder_threads")) { *value = c->feeder_threads; return LISREG_OK; }
    if (!strcmp(name, "feeder_numa_node")) { *value = c->feeder_node; return LISREG_OK; }
    if (!strcmp(name, "feeder_numa_cpus")) { *value = c->feeder_cpus; return LISREG_OK; }
    if (!strcmp(name, "comm_nranks")) { *value = c->comm ? c->comm_nranks : 0; return LISREG_OK; }
    if (!strcmp(name, "sorted_now")) { *value = c->sort_now ? 1 : 0; return LISREG_OK; }
    if (!strcmp(name, "rebuild_targets_each_run")) { *value = c->rebuild_targets_each_run ? 1 : 0; return LISREG_OK; }
    if (!strcmp(name, "graph_min_ratio")) { *value = c->graph_min_ratio; return LISREG_OK; }
    if (!strcmp(name, "cell_min_ratio")) { *value = c->cell_min_ratio; return LISREG_OK; }
    if (!strcmp(name, "row_reach")) { *value = c->row_reach; return LISREG_OK; }
    if (!strcmp(name, "row_reach_now")) { *value = c->reach_now ? 1 : 0; return LISREG_OK; }
    if (!strcmp(name, "row_reach_misses")) { *value = c->reach_miss_last; return LISREG_OK; }
    if (!strcmp(name, "cell_rows_max_mb")) { *value = c->cell_rows_max_mb; return LISREG_OK; }
    if (!strcmp(name, "xcd_order")) { *value = c->xcd_order; return LISREG_OK; }
    if (!strcmp(name, "xcd_order_now")) { *value = c->xcd_now ? 1 : 0; return LISREG_OK; }
    if (!strcmp(name, "interleave")) { *value = c->interleave; return LISREG_OK; }
    if (!strcmp(name, "interleaved_now")) { *value = c->interleaved_now ? 1 : 0; return LISREG_OK; }
    // size of the search index of the prepared batch's targets, in KiB (what the front-end in use reads), and their points:
    //   index_kib_grid: sorted records + cell table; index_kib_front_end: k-NN graph rows (front-end 3) or cell rows + their table (front-end 5)
    if (!strcmp(name, "index_kib_front_end_built")) {
        // the cell rows the LAST run actually built (option "row_reach" leaves out the cells no query comes near): row counts read back from the
        // device (a synchronous diagnostic); other front-ends: what "index_kib_front_end" says
        if (c->mode_now != 5) return lisreg_get_option(c, "index_kib_front_end", value);
        unsigned long long fe = 0;
        for (int slot : c->batch_slots) {
            if (slot < 0 || (size_t)slot >= c->targets.size()) continue;
            const lisreg::Target& t = c->targets[(size_t)slot];
            for (int k = 0; k < 2; ++k) {
                if (t.n[k] <= 0 || !t.crow_scan[k].p) continue;
                int rows = 0;
                if (hipSetDevice(c->device) != hipSuccess || hipStreamSynchronize(c->stream) != hipSuccess ||
                    hipMemcpy(&rows, t.crow_scan[k].as<int>() + t.n_cells[k], sizeof(int), hipMemcpyDeviceToHost) != hipSuccess) return LISREG_ERR_HIP;
                rows = std::min(std::max(rows, 0), t.crow_cap[k]);
                fe += (unsigned long long)rows * (sizeof(float4) * lisreg::kGraphK + sizeof(float2)) + (unsigned long long)t.n_cells[k] * sizeof(int);
            }
        }
        *value = (int)std::min<unsigned long long>((fe + 1023) / 1024, 0x7fffffffULL);
        return LISREG_OK;
    }
    if (!strcmp(name, "index_kib_grid") || !strcmp(name, "index_kib_front_end") || !strcmp(name, "index_target_points")) {
        unsigned long long grid = 0, fe = 0, pts = 0;
        for (int slot : c->batch_slots) {
            if (slot < 0 || (size_t)slot >= c->targets.size()) continue;
            const lisreg::Target& t = c->targets[(size_t)slot];
            for (int k = 0; k < 2; ++k) {
                if (t.n[k] <= 0) continue;
                pts += (unsigned long long)t.n[k];
                grid += (unsigned long long)t.n[k] * sizeof(float4) + ((unsigned long long)t.n_cells[k] + 1) * sizeof(int);
                if (c->mode_now == 3) fe += (unsigned long long)t.n[k] * (sizeof(float4) * lisreg::kGraphK + sizeof(float2));
                if (c->mode_now == 5) fe += (unsigned long long)t.crow_cap[k] * (sizeof(float4) * lisreg::kGraphK + sizeof(float2)) +
                                            (unsigned long long)t.n_cells[k] * sizeof(int);
            }
        }
        const unsigned long long v = !strcmp(name, "index_target_points") ? pts : ((!strcmp(name, "index_kib_grid") ? grid : fe) + 1023) / 1024;
        *value = (int)std::min<unsigned long long>(v, 0x7fffffffULL);
        return LISREG_OK;
    }
    return LISREG_ERR_ARG;
}

int lisreg_align_batch(lisreg_ctx* c, int n_items, const lisreg_item* items, const lisreg_params* params, float* T,
                       lisreg_stats* stats)
{
    if (!c) return LISREG_ERR_ARG;
    if (n_items < 0 || (n_items > 0 && (!items || !T)) || !params) return fail(c, LISREG_ERR_ARG, "align_batch: bad arguments");
    const auto t_in = std::chrono::steady_clock::now();
    HIPCHK(c, hipSetDevice(c->device));
    // stage host clouds into one device buffer of 16-B records; device items pass through
    size_t total = 0;
    for (int i = 0; i < n_items; ++i) {
        const lisreg_item& in = items[i];
        if (in.n_corner < 0 || in.n_surf < 0) return fail(c, LISREG_ERR_ARG, "align_batch: negative count");
        if (in.fmt != LISREG_FMT_DEVICE) {
            if (in.stride_bytes < 12 || (in.fmt == LISREG_FMT_XYZIL && in.stride_bytes < 22)) return fail(c, LISREG_ERR_ARG, "align_batch: bad stride");
            if ((in.n_corner > 0 && !in.src_corner) || (in.n_surf > 0 && !in.src_surf)) return fail(c, LISREG_ERR_ARG, "align_batch: NULL cloud");
            total += (size_t)in.n_corner + (size_t)in.n_surf;
        }
    }
    std::vector<lisreg_item> dev_items((size_t)n_items);
    // Big host batches go through the feeder (lisreg_api_feed.hip): a few host threads pack the structs to 16-byte records in pinned
    // staging and the chunks are uploaded as they complete — half the bytes on the link, a few big copies instead of one per cloud.
    if (total >= 262144 && c->feeder_threads > 0) {
        int rc = lisreg_stage_host_items(c, n_items, items, dev_items.data());
        if (rc) return rc;
        rc = lisreg_batch_prepare(c, n_items, dev_items.data(), params, T);
        if (rc) { (void)hipStreamSynchronize(c->stream); return rc; }
        rc = run_impl(c, true);
        if (rc) { (void)hipStreamSynchronize(c->stream); return rc; }
        return lisreg_batch_fetch(c, T, stats);
    }
    // Small ones (a single odometry frame) cross PCIe as they are — the caller's structs, asynchronously on the context's stream —
    // and are packed to 16-byte records on the device: no thread is woken, nothing is repacked on the CPU.
    HIPCHK(c, c->src_upload.ensure(sizeof(lisreg_dpoint) * std::max<size_t>(total, 1)));
    size_t raw_bytes = 0;
    for (int i = 0; i < n_items; ++i)
        if (items[i].fmt != LISREG_FMT_DEVICE) raw_bytes += ((size_t)items[i].n_corner + (size_t)items[i].n_surf) * (size_t)items[i].stride_bytes + 32;
    HIPCHK(c, c->raw_upload.ensure(std::max<size_t>(raw_bytes, 16)));
    size_t off = 0, roff = 0;
    for (int i = 0; i < n_items; ++i) {
        dev_items[(size_t)i] = items[i];
        const lisreg_item& in = items[i];
        if (in.fmt == LISREG_FMT_DEVICE) continue;
        lisreg_item& d = dev_items[(size_t)i];
        const void* srcs[2] = { in.src_corner, in.src_surf };
        const int cnts[2] = { in.n_corner, in.n_surf };
        const void** dsts[2] = { &d.src_corner, &d.src_surf };
        for (int k = 0; k < 2; ++k) {
            lisreg_dpoint* dst = c->src_upload.as<lisreg_dpoint>() + off;
            *dsts[k] = dst;
            if (cnts[k] > 0) {
                unsigned char* rdst = static_cast<unsigned char*>(c->raw_upload.p) + roff;
                const size_t bytes = (size_t)cnts[k] * (size_t)in.stride_bytes;
                HIPCHK(c, hipMemcpyAsync(rdst, srcs[k], bytes, hipMemcpyHostToDevice, c->stream));
                launch_pack_cloud(rdst, (size_t)cnts[k], in.stride_bytes, in.fmt == LISREG_FMT_XYZIL, reinterpret_cast<float4*>(dst), c->stream);
                roff += (bytes + 15) & ~(size_t)15;
            }
            off += (size_t)cnts[k];
        }
        d.fmt = LISREG_FMT_DEVICE;
    }
    HIPCHK(c, hipGetLastError());
    static const bool host_prof = getenv("LISREG_HOST_PROF") != nullptr;      // where a synchronous call spends its host time
    auto now = [] { return std::chrono::steady_clock::now(); };
    const auto t0 = now();
    // a failure after uploads have been enqueued must not return while the DMA still reads the caller's (possibly pinned) buffers
    int rc = lisreg_batch_prepare(c, n_items, dev_items.data(), params, T);
    if (rc) { (void)hipStreamSynchronize(c->stream); return rc; }
    const auto t1 = now();
    rc = run_impl(c, true);
    if (rc) { (void)hipStreamSynchronize(c->stream); return rc; }
    const auto t2 = now();
    rc = lisreg_batch_fetch(c, T, stats);
    if (host_prof) {
        auto us = [](auto a, auto b) { return std::chrono::duration<double, std::micro>(b - a).count(); };
        fprintf(stderr, "[lisreg host] upload %.1f us, prepare %.1f us, enqueue %.1f us, fetch (incl. GPU wait) %.1f us\n",
                us(t_in, t0), us(t0, t1), us(t1, t2), us(t2, now()));
    }
    return rc;
}

int lisreg_align(lisreg_ctx* c, const void* src_corner, int n_corner, const void* src_surf, int n_surf, int stride,
                 int fmt, const lisreg_params* params, const lisreg_imu* imu, float T[6], lisreg_stats* stats)
{
    if (!c) return LISREG_ERR_ARG;
    if (!params || !T) return fail(c, LISREG_ERR_ARG, "align: params/T NULL");
    if (c->targets.empty() || !c->targets[0].valid) return fail(c, LISREG_ERR_NO_TARGET, "align: call lisreg_set_target first");
    lisreg_item item;
    memset(&item, 0, sizeof item);
    item.src_corner = src_corner; item.n_corner = n_corner;
    item.src_surf = src_surf; item.n_surf = n_surf;
    item.stride_bytes = stride; item.fmt = fmt; item.target = 0;
    item.degenerate_in = c->degenerate;
    if (imu) item.imu = *imu;
    const int bound = params->fixed_iters > 0 ? params->fixed_iters : params->max_iters;
    const int saved_cap = c->trace_cap;
    c->trace_cap = std::max(bound, 1);
    lisreg_stats st;
    memset(&st, 0, sizeof st);
    c->fetch_trace_records = c->trace_cap;             // the trace rides along with the result copy (one synchronisation)
    int rc = lisreg_align_batch(c, 1, &item, params, T, &st);
    c->fetch_trace_records = 0;
    if (rc == LISREG_OK) {
        c->degenerate = st.degenerate;
        c->last_trace_n = std::min(bound, st.iters + 1);
        if (st.status == LISREG_NOT_ENOUGH_FEATURES) c->last_trace_n = 0;
        c->last_trace.assign((size_t)kTraceStride * (size_t)std::max(c->last_trace_n, 1), 0.f);
        if (c->last_trace_n > 0 && c->fetch_host)
            memcpy(c->last_trace.data(), c->fetch_host + 2 * (size_t)kResultSize, sizeof(float) * kTraceStride * (size_t)c->last_trace_n);      // (one item + the "row_reach" record in front: lisreg_batch_fetch)
        if (stats) *stats = st;
        rc = st.status;
    }
    c->trace_cap = saved_cap;
    c->prepared = false;
    return rc;
}

int lisreg_get_trace(lisreg_ctx* c, float* buf, int max_iters)
{
    if (!c || !buf || max_iters <= 0) return 0;
    const int n = std::min(max_iters, c->last_trace_n);
    if (n > 0) memcpy(buf, c->last_trace.data(), sizeof(float) * kTraceStride * (size_t)n);
    return n;
}

int lisreg_get_counters(lisreg_ctx* c, unsigned long long* out, int n)
{
    if (!c || !out || n < 0) return LISREG_ERR_ARG;
    for (int i = 0; i < n; ++i) out[i] = 0;
    if (!c->counters.p) return LISREG_OK;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipMemcpy(out, c->counters.p, sizeof(unsigned long long) * (size_t)std::min(n, 128), hipMemcpyDeviceToHost));
    if (c->mode_now == 3 || c->mode_now == 5) {              // the graph scan packs (walked << 32 | valid) per GN iteration: unpack to the pair layout
        unsigned long long tmp[64] = { 0 };
        for (int i = 0; i < 32 && 2 * i + 1 < n; ++i) { tmp[2 * i] = out[i] >> 32; tmp[2 * i + 1] = out[i] & 0xffffffffull; }
        for (int i = 0; i < std::min(n, 64); ++i) out[i] = tmp[i];
    }
    return LISREG_OK;
}

int lisreg_get_target_index(lisreg_ctx* c, int slot, int kind, int* dims /* n, nx, ny, nz, n_cells */, float* geom /* ox, oy, oz, cell */,
                            float* sorted_out, int sorted_capacity, int* cell_start_out, int cell_capacity)
{
    if (!c || slot < 0 || (size_t)slot >= c->targets.size() || kind < 0 || kind > 1 || !c->targets[(size_t)slot].valid)
        return fail(c, LISREG_ERR_ARG, "get_target_index: no such target");
    const Target& t = c->targets[(size_t)slot];
    const GridIndex& g = t.g[kind];
    if (dims) { dims[0] = t.n[kind]; dims[1] = g.nx; dims[2] = g.ny; dims[3] = g.nz; dims[4] = t.n_cells[kind]; }
    if (geom) { geom[0] = g.ox; geom[1] = g.oy; geom[2] = g.oz; geom[3] = g.cell; }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (sorted_out) {
        if (sorted_capacity < t.n[kind]) return fail(c, LISREG_ERR_ARG, "get_target_index: sorted_capacity too small");
        if (t.n[kind] > 0) HIPCHK(c, hipMemcpy(sorted_out, t.sorted[kind].p, sizeof(float4) * (size_t)t.n[kind], hipMemcpyDeviceToHost));
    }
    if (cell_start_out) {
        if (cell_capacity < t.n_cells[kind] + 1) return fail(c, LISREG_ERR_ARG, "get_target_index: cell_capacity too small");
        HIPCHK(c, hipMemcpy(cell_start_out, t.cell_start[kind].p, sizeof(int) * ((size_t)t.n_cells[kind] + 1), hipMemcpyDeviceToHost));
    }
    return LISREG_OK;
}

int lisreg_get_target_graph(lisreg_ctx* c, int slot, int kind, int* k_out, float* rows_out, float* meta_out, int capacity_points)
{
    if (!c || slot < 0 || (size_t)slot >= c->targets.size() || kind < 0 || kind > 1 || !c->targets[(size_t)slot].valid)
        return fail(c, LISREG_ERR_ARG, "get_target_graph: no such target");
    Target& t = c->targets[(size_t)slot];
    if (k_out) *k_out = kGraphK;
    if (!rows_out && !meta_out) return LISREG_OK;
    if (capacity_points < t.n[kind]) return fail(c, LISREG_ERR_ARG, "get_target_graph: capacity_points too small");
    HIPCHK(c, hipSetDevice(c->device));
    if (!t.graph_valid[kind] || !t.g[kind].nbr) {            // build it now (it is built on demand otherwise)
        int rc = ensure_graph(c, t, kind, true);
        if (rc) return rc;
        c->grids_dirty = true;
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    const size_t n = (size_t)t.n[kind];
    if (rows_out && n) HIPCHK(c, hipMemcpy(rows_out, t.nbr[kind].p, sizeof(float4) * kGraphK * n, hipMemcpyDeviceToHost));
    if (meta_out && n) HIPCHK(c, hipMemcpy(meta_out, t.nbr_meta[kind].p, sizeof(float2) * n, hipMemcpyDeviceToHost));
    return LISREG_OK;
}

int lisreg_get_target_cell_rows(lisreg_ctx* c, int slot, int kind, int* n_rows, int* k_out, int* table_out, int capacity_cells,
                                float* rows_out, float* meta_out, int capacity_rows)
{
    if (!c || slot < 0 || (size_t)slot >= c->targets.size() || kind < 0 || kind > 1 || !c->targets[(size_t)slot].valid)
        return fail(c, LISREG_ERR_ARG, "get_target_cell_rows: no such target");
    HIPCHK(c, hipSetDevice(c->device));
    Target& t = c->targets[(size_t)slot];
    if (k_out) *k_out = kGraphK;
    if (!t.crow_valid[kind] || !t.g[kind].crow_tab) {
        int rc = ensure_crows(c, t, kind);
        if (rc) return rc;
        c->grids_dirty = true;
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    int rows = 0;
    if (t.n[kind] > 0) HIPCHK(c, hipMemcpy(&rows, t.crow_scan[kind].as<int>() + t.n_cells[kind], sizeof(int), hipMemcpyDeviceToHost));
    rows = std::min(rows, t.crow_cap[kind]);
    if (n_rows) *n_rows = rows;
    if (table_out) {
        if (capacity_cells < t.n_cells[kind]) return fail(c, LISREG_ERR_ARG, "get_target_cell_rows: capacity_cells too small");
        HIPCHK(c, hipMemcpy(table_out, t.crow_tab[kind].p, sizeof(int) * (size_t)t.n_cells[kind], hipMemcpyDeviceToHost));
    }
    if ((rows_out || meta_out) && capacity_rows < rows) return fail(c, LISREG_ERR_ARG, "get_target_cell_rows: capacity_rows too small");
    if (rows_out && rows) {
        HIPCHK(c, hipMemcpy(rows_out, t.crow[kind].p, sizeof(float4) * kGraphK * (size_t)rows, hipMemcpyDeviceToHost));
    }
    if (meta_out && rows) HIPCHK(c, hipMemcpy(meta_out, t.crow_meta[kind].p, sizeof(float2) * (size_t)rows, hipMemcpyDeviceToHost));
    return LISREG_OK;
}

int lisreg_get_neighbors(lisreg_ctx* c, int* out, int n_elems)
{
    if (!c || !out || n_elems < 0) return LISREG_ERR_ARG;
    if (!c->dump_neighbors || !c->dbg_nn.p || n_elems != c->n_elems || (c->mode_now != 1 && c->mode_now != 3 && c->mode_now != 5))
        return fail(c, LISREG_ERR_ARG, "get_neighbors: set option dump_neighbors before preparing the batch (search modes 1, 3); n_elems must be the batch's source point count");
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipMemcpy(out, c->dbg_nn.p, sizeof(int) * 6 * (size_t)n_elems, hipMemcpyDeviceToHost));
    return LISREG_OK;
}

int lisreg_test_fit_models(lisreg_ctx* c, int kind, int n, const float* neighbours, const float* queries, const lisreg_params* params,
                           int exact, float* out)
{
    if (!c) return LISREG_ERR_ARG;
    if ((kind != 0 && kind != 1) || n < 0 || !params || (n > 0 && (!neighbours || !queries || !out)))
        return fail(c, LISREG_ERR_ARG, "test_fit_models: bad arguments");
    if (n == 0) return LISREG_OK;
    HIPCHK(c, hipSetDevice(c->device));
    DevBuf nb, q, o;
    HIPCHK(c, nb.ensure(sizeof(float) * 15 * (size_t)n));
    HIPCHK(c, q.ensure(sizeof(float) * 3 * (size_t)n));
    HIPCHK(c, o.ensure(sizeof(float) * 10 * (size_t)n));
    HIPCHK(c, hipMemcpyAsync(nb.p, neighbours, sizeof(float) * 15 * (size_t)n, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(q.p, queries, sizeof(float) * 3 * (size_t)n, hipMemcpyHostToDevice, c->stream));
    DevParams prm = make_dev_params(*params);
    prm.exact = exact ? 1 : 0;
    if (exact) launch_test_fit_exact(kind, n, nb.as<float>(), q.as<float>(), prm, o.as<float>(), c->stream);
    else launch_test_fit(kind, n, nb.as<float>(), q.as<float>(), prm, o.as<float>(), c->stream);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(out, o.p, sizeof(float) * 10 * (size_t)n, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    nb.release(); q.release(); o.release();
    return LISREG_OK;
}

int lisreg_set_profiling(lisreg_ctx* c, int enable)
{
    if (!c) return LISREG_ERR_ARG;
    c->profiling = enable != 0;
    return LISREG_OK;
}

int lisreg_get_timing(lisreg_ctx* c, double out[5])
{
    if (!c || !out) return LISREG_ERR_ARG;
    for (int i = 0; i < 5; ++i) out[i] = c->timing[i];
    return LISREG_OK;
}

// ---- §8 f-1: voxel-grid down-sampling and cloud transform --------------------------------------------------------------
int lisreg_voxel_downsample(lisreg_ctx* c, const void* in, int n, int stride, int fmt, float leaf, void* out,
                            int out_capacity, int* n_out)
{
    if (!c) return LISREG_ERR_ARG;
    if (!n_out || n < 0 || !(leaf > 0.f) || (n > 0 && (!in || !out))) return fail(c, LISREG_ERR_ARG, "voxel_downsample: bad arguments");
    const bool dev = fmt == LISREG_FMT_DEVICE || fmt == LISREG_FMT_DEVICE_XYZI;
    if (!dev && (stride < 12 || (fmt == LISREG_FMT_XYZIL && stride < 22)))
        return fail(c, LISREG_ERR_ARG, "voxel_downsample: bad stride");
    *n_out = 0;
    if (n == 0) return LISREG_OK;
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t st = c->stream;
    const bool has_intensity = !dev && stride >= 20;
    // ---- stage the input as float4 (x,y,z, intensity | payload) [+ labels] ---------------------------------------
    const float4* pts = nullptr;
    const uint32_t* labels = nullptr;
    std::vector<float4> h_pts;
    std::vector<uint32_t> h_lab;
    if (dev) pts = static_cast<const float4*>(in);
    else {
        h_pts.resize((size_t)n);
        if (fmt == LISREG_FMT_XYZIL) h_lab.resize((size_t)n);
        const unsigned char* b = static_cast<const unsigned char*>(in);
        for (int i = 0; i < n; ++i) {
            const unsigned char* r = b + (size_t)i * (size_t)stride;
            float v[3], it = 0.f;
            memcpy(v, r, 12);
            if (has_intensity) memcpy(&it, r + 16, 4);
            h_pts[(size_t)i] = make_float4(v[0], v[1], v[2], it);
            if (fmt == LISREG_FMT_XYZIL) { uint16_t l; memcpy(&l, r + 20, 2); h_lab[(size_t)i] = l; }
        }
        HIPCHK(c, c->vox_in.ensure(sizeof(float4) * (size_t)n));
        HIPCHK(c, hipMemcpyAsync(c->vox_in.p, h_pts.data(), sizeof(float4) * (size_t)n, hipMemcpyHostToDevice, st));
        pts = c->vox_in.as<float4>();
        if (fmt == LISREG_FMT_XYZIL) {
            HIPCHK(c, c->vox_lab.ensure(sizeof(uint32_t) * (size_t)n));
            HIPCHK(c, hipMemcpyAsync(c->vox_lab.p, h_lab.data(), sizeof(uint32_t) * (size_t)n, hipMemcpyHostToDevice, st));
            labels = c->vox_lab.as<uint32_t>();
        }
    }
    // ---- getMinMax3D + grid geometry (voxel_grid.hpp) -------------------------------------------------------------
    float bb[6];
    HIPCHK(c, c->bbox_dev.ensure(sizeof(float) * 8));
    HIPCHK(c, c->bbox_scratch.ensure(sizeof(float) * 6 * 256));
    launch_bbox(pts, n, c->bbox_dev.as<float>(), c->bbox_scratch.as<float>(), st);
    HIPCHK(c, hipMemcpyAsync(bb, c->bbox_dev.p, sizeof bb, hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipStreamSynchronize(st));
    for (int k = 0; k < 6; ++k)       // the reference strips non-finite returns before any filter (pcl::removeNaNFromPointCloud)
        if (!std::isfinite(bb[k])) return fail(c, LISREG_ERR_ARG, "voxel_downsample: the cloud has infinite coordinates");
    const float inv = 1.0f / leaf;
    const long long dx = (long long)((bb[3] - bb[0]) * inv) + 1, dy = (long long)((bb[4] - bb[1]) * inv) + 1,
                    dz = (long long)((bb[5] - bb[2]) * inv) + 1;
    if (dx * dy * dz > 2147483647LL) {           // "Leaf size is too small for the input dataset": output = input
        if (n > out_capacity) { *n_out = n; return fail(c, LISREG_ERR_ARG, "voxel_downsample: out_capacity too small"); }
        if (dev) { if (out != in) HIPCHK(c, hipMemcpyAsync(out, in, sizeof(float4) * (size_t)n, hipMemcpyDeviceToDevice, st)); }
        else if (out != in) memmove(out, in, (size_t)n * (size_t)stride);
        HIPCHK(c, hipStreamSynchronize(st));
        *n_out = n;
        return LISREG_LEAF_TOO_SMALL;
    }
    VoxelDesc d;
    int div_b[3];
    const int min_b[3] = { (int)floorf(bb[0] * inv), (int)floorf(bb[1] * inv), (int)floorf(bb[2] * inv) };
    const int max_b[3] = { (int)floorf(bb[3] * inv), (int)floorf(bb[4] * inv), (int)floorf(bb[5] * inv) };
    for (int k = 0; k < 3; ++k) div_b[k] = max_b[k] - min_b[k] + 1;
    d.inv_leaf = inv; d.min_b0 = min_b[0]; d.min_b1 = min_b[1]; d.min_b2 = min_b[2];
    d.mul1 = div_b[0]; d.mul2 = div_b[0] * div_b[1];
    const long long total = (long long)div_b[0] * div_b[1] * div_b[2];
    const long long max_buckets = 1LL << 22;
    d.span = (uint32_t)std::max(1LL, (total + max_buckets - 1) / max_buckets);
    const int n_buckets = (int)((total + d.span - 1) / d.span);
    // ---- sort by voxel index, count voxels ------------------------------------------------------------------------
    int rc = ensure_sort_scratch(c, (size_t)n, (size_t)std::max(n_buckets, n) + 1);
    if (rc) return rc;
    HIPCHK(c, c->vox_order.ensure(sizeof(int) * (size_t)n));
    HIPCHK(c, c->vox_sidx.ensure(sizeof(uint32_t) * (size_t)n));
    HIPCHK(c, c->vox_head.ensure(sizeof(int) * ((size_t)n + 1)));
    HIPCHK(c, c->vox_slot.ensure(sizeof(int) * ((size_t)n + 2)));
    launch_voxel_sort(pts, n, d, n_buckets, sort_buffers(c), c->vox_order.as<int>(), c->vox_sidx.as<uint32_t>(),
                      c->vox_head.as<int>(), c->vox_slot.as<int>(), st);
    int n_vox = 0;
    HIPCHK(c, hipMemcpyAsync(&n_vox, c->vox_slot.as<int>() + n, sizeof(int), hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipStreamSynchronize(st));
    *n_out = n_vox;
    if (n_vox > out_capacity) return fail(c, LISREG_ERR_ARG, "voxel_downsample: out_capacity too small (see *n_out)");
    // ---- centroids ---------------------------------------------------------------------------------------------------
    HIPCHK(c, c->vox_start.ensure(sizeof(int) * ((size_t)n_vox + 2)));
    // in place (out inside the input records — lisreg_localmap_extract grids a class cloud onto itself): the centroid kernels read
    // pts[order[..]] while other threads write out[v], so the result is formed in scratch and copied over the input afterwards
    const bool aliased = dev && (const char*)out < (const char*)in + sizeof(float4) * (size_t)n &&
                         (const char*)in < (const char*)out + sizeof(float4) * (size_t)std::max(out_capacity, 1);
    float4* out_pts = dev && !aliased ? static_cast<float4*>(out) : nullptr;
    if (!out_pts) { HIPCHK(c, c->vox_out.ensure(sizeof(float4) * (size_t)std::max(n_vox, 1))); out_pts = c->vox_out.as<float4>(); }
    uint32_t* out_lab = nullptr;
    if (fmt == LISREG_FMT_XYZIL) { HIPCHK(c, c->vox_outlab.ensure(sizeof(uint32_t) * (size_t)n_vox)); out_lab = c->vox_outlab.as<uint32_t>(); }
    launch_voxel_centroids(n, n_vox, pts, labels, fmt == LISREG_FMT_DEVICE ? 1 : 0 /* label vote on the payload, else .w averaged */, c->vox_order.as<int>(), c->vox_head.as<int>(),
                           c->vox_slot.as<int>(), c->vox_start.as<int>(), out_pts, out_lab, st);
    HIPCHK(c, hipGetLastError());
    if (!dev) {
        std::vector<float4> r((size_t)n_vox);
        std::vector<uint32_t> rl(fmt == LISREG_FMT_XYZIL ? (size_t)n_vox : 0);
        HIPCHK(c, hipMemcpyAsync(r.data(), out_pts, sizeof(float4) * (size_t)n_vox, hipMemcpyDeviceToHost, st));
        if (out_lab) HIPCHK(c, hipMemcpyAsync(rl.data(), out_lab, sizeof(uint32_t) * (size_t)n_vox, hipMemcpyDeviceToHost, st));
        HIPCHK(c, hipStreamSynchronize(st));
        unsigned char* o = static_cast<unsigned char*>(out);
        for (int i = 0; i < n_vox; ++i) {
            unsigned char* q = o + (size_t)i * (size_t)stride;
            memset(q, 0, (size_t)stride);
            memcpy(q, &r[(size_t)i], 12);
            if (has_intensity) memcpy(q + 16, &r[(size_t)i].w, 4);
            if (out_lab) { const uint16_t l = (uint16_t)rl[(size_t)i]; memcpy(q + 20, &l, 2); }
        }
    } else {
        if (aliased && n_vox > 0) HIPCHK(c, hipMemcpyAsync(out, out_pts, sizeof(float4) * (size_t)n_vox, hipMemcpyDeviceToDevice, st));
        HIPCHK(c, hipStreamSynchronize(st));
    }
    return LISREG_OK;
}

// K device clouds through ONE launch sequence (one sort keyed by (cloud, voxel index), one centroid launch) with three host round trips
// in all — the K bounding boxes, the K voxel counts — instead of ~12 launches and three round trips per cloud: the five class grids of a
// key frame (subMapOptmizationNode.cpp:806-811) or of extractSlidingCloud (:1385-1389) are launch-bound, not bandwidth-bound.
// Results are those of K lisreg_voxel_downsample calls, bit for bit (same sort order inside every cloud, same sequential sums).
int lisreg_voxel_downsample_multi(lisreg_ctx* c, int k, const void* const* in, const int* n, const float* leaf, int fmt,
                                  void* const* out, const int* out_capacity, int* n_out)
{
    if (!c) return LISREG_ERR_ARG;
    if (k < 0 || (k > 0 && (!in || !n || !leaf || !out || !out_capacity || !n_out))) return fail(c, LISREG_ERR_ARG, "voxel_downsample_multi: bad arguments");
    if (fmt != LISREG_FMT_DEVICE && fmt != LISREG_FMT_DEVICE_XYZI) return fail(c, LISREG_ERR_ARG, "voxel_downsample_multi: device records only (LISREG_FMT_DEVICE / _DEVICE_XYZI)");
    for (int s = 0; s < k; ++s) {
        if (n[s] < 0 || !(leaf[s] > 0.f) || (n[s] > 0 && (!in[s] || !out[s]))) return fail(c, LISREG_ERR_ARG, "voxel_downsample_multi: bad cloud");
        n_out[s] = 0;
    }
    auto one_by_one = [&]() -> int {
        for (int s = 0; s < k; ++s) {
            int rc = lisreg_voxel_downsample(c, in[s], n[s], 16, fmt, leaf[s], out[s], out_capacity[s], &n_out[s]);
            if (rc != LISREG_OK && rc != LISREG_LEAF_TOO_SMALL) return rc;
        }
        return LISREG_OK;
    };
    long long total_n = 0;
    int live = 0;
    for (int s = 0; s < k; ++s) { total_n += n[s]; live += n[s] > 0; }
    if (live <= 1 || k > kVoxelMultiMax || total_n > 2000000000LL) return one_by_one();
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t st = c->stream;
    const int N = (int)total_n;
    // ---- concatenate, K bounding boxes, one round trip ---------------------------------------------------------------------------
    HIPCHK(c, c->vox_in.ensure(sizeof(float4) * (size_t)N));
    HIPCHK(c, c->bbox_dev.ensure(sizeof(float) * 6 * kVoxelMultiMax));
    HIPCHK(c, c->bbox_scratch.ensure(sizeof(float) * 6 * 256 * kVoxelMultiMax));
    VoxelMulti m;
    memset(&m, 0, sizeof m);
    m.k = k;
    float4* cat = c->vox_in.as<float4>();
    for (int s = 0, o = 0; s < k; ++s) {
        m.off[s] = o;
        o += n[s];
        m.off[s + 1] = o;
    }
    {
        BboxJobs jobs;
        memset(&jobs, 0, sizeof jobs);
        jobs.k = k;
        for (int s = 0; s < k; ++s) { jobs.pts[s] = static_cast<const float4*>(in[s]); jobs.n[s] = n[s]; }
        launch_concat_jobs(jobs, m, cat, st);                  // one launch instead of K copies
    }
    launch_bbox_multi(cat, m, c->bbox_dev.as<float>(), c->bbox_scratch.as<float>(), st);
    float bb[6 * kVoxelMultiMax];
    HIPCHK(c, hipMemcpyAsync(bb, c->bbox_dev.p, sizeof(float) * 6 * (size_t)k, hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipStreamSynchronize(st));
    // ---- per-cloud geometry (voxel_grid.hpp), one common bucket span ----------------------------------------------------------
    long long totals[kVoxelMultiMax] = { 0 }, sum_total = 0;
    for (int s = 0; s < k; ++s) {
        if (n[s] == 0) continue;
        const float* b = bb + 6 * s;
        for (int q = 0; q < 6; ++q) if (!std::isfinite(b[q])) return fail(c, LISREG_ERR_ARG, "voxel_downsample_multi: a cloud has infinite coordinates");
        const float inv = 1.0f / leaf[s];
        const long long dx = (long long)((b[3] - b[0]) * inv) + 1, dy = (long long)((b[4] - b[1]) * inv) + 1, dz = (long long)((b[5] - b[2]) * inv) + 1;
        if (dx * dy * dz > 2147483647LL) return one_by_one();          // "leaf size too small" for one of them: the single-cloud path knows what to do
        const int min_b[3] = { (int)floorf(b[0] * inv), (int)floorf(b[1] * inv), (int)floorf(b[2] * inv) };
        const int max_b[3] = { (int)floorf(b[3] * inv), (int)floorf(b[4] * inv), (int)floorf(b[5] * inv) };
        int div_b[3];
        for (int q = 0; q < 3; ++q) div_b[q] = max_b[q] - min_b[q] + 1;
        VoxelDesc& d = m.d[s];
        d.inv_leaf = inv; d.min_b0 = min_b[0]; d.min_b1 = min_b[1]; d.min_b2 = min_b[2];
        d.mul1 = div_b[0]; d.mul2 = div_b[0] * div_b[1];
        totals[s] = (long long)div_b[0] * div_b[1] * div_b[2];
        sum_total += totals[s];
    }
    if (sum_total >= (1LL << 32)) return one_by_one();                 // the joint voxel index has to fit 32 bits
    // every cloud its own bucket span (a cloud with a tiny leaf must not coarsen the others' buckets: the rank pass is quadratic inside
    // a bucket), at most 2^22 buckets in all
    const long long max_buckets = (1LL << 22) / k;
    long long nb = 0, ib = 0;
    for (int s = 0; s < k; ++s) {
        const uint32_t span = (uint32_t)std::max(1LL, (totals[s] + max_buckets - 1) / max_buckets);
        m.d[s].span = span;
        m.bucket_base[s] = (int)nb;
        m.idx_base[s] = (uint32_t)ib;
        nb += (totals[s] + span - 1) / span;
        ib += totals[s];
    }
    m.bucket_base[k] = (int)nb;
    m.idx_base[k] = (uint32_t)ib;
    const int n_buckets = (int)std::max(nb, 1LL);
    // ---- one sort, the K voxel counts in one round trip -------------------------------------------------------------------------
    int rc = ensure_sort_scratch(c, (size_t)N, (size_t)std::max(n_buckets, N) + 1);
    if (rc) return rc;
    HIPCHK(c, c->vox_order.ensure(sizeof(int) * (size_t)N));
    HIPCHK(c, c->vox_sidx.ensure(sizeof(uint32_t) * (size_t)N));
    HIPCHK(c, c->vox_head.ensure(sizeof(int) * ((size_t)N + 1)));
    HIPCHK(c, c->vox_slot.ensure(sizeof(int) * ((size_t)N + 2)));
    launch_voxel_sort_multi(cat, N, m, n_buckets, sort_buffers(c), c->vox_order.as<int>(), c->vox_sidx.as<uint32_t>(),
                            c->vox_head.as<int>(), c->vox_slot.as<int>(), st);
    int vo[kVoxelMultiMax + 1];
    // the sorted sequence is cloud by cloud: cloud s starts at sorted position off[s]; its voxels start at slot[off[s]]
    HIPCHK(c, c->mp_cnt.ensure(sizeof(int) * (kVoxelMultiMax + 1)));
    launch_multi_bounds(c->vox_slot.as<int>(), m, c->mp_cnt.as<int>(), st);
    HIPCHK(c, hipMemcpyAsync(vo, c->mp_cnt.p, sizeof(int) * (size_t)(k + 1), hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipStreamSynchronize(st));
    const int n_vox = vo[k];
    for (int s = 0; s < k; ++s) {
        n_out[s] = vo[s + 1] - vo[s];
        if (n_out[s] > out_capacity[s]) return fail(c, LISREG_ERR_ARG, "voxel_downsample_multi: out_capacity too small (see n_out)");
    }
    // ---- one centroid launch, the slices handed out ------------------------------------------------------------------------------
    HIPCHK(c, c->vox_start.ensure(sizeof(int) * ((size_t)n_vox + 2)));
    HIPCHK(c, c->vox_out.ensure(sizeof(float4) * (size_t)std::max(n_vox, 1)));
    launch_voxel_centroids(N, n_vox, cat, nullptr, fmt == LISREG_FMT_DEVICE ? 1 : 0, c->vox_order.as<int>(), c->vox_head.as<int>(),
                           c->vox_slot.as<int>(), c->vox_start.as<int>(), c->vox_out.as<float4>(), nullptr, st);
    HIPCHK(c, hipGetLastError());
    VoxelHandOut ho;
    memset(&ho, 0, sizeof ho);
    ho.k = k;
    for (int s = 0; s <= k; ++s) ho.vo[s] = vo[s];
    for (int s = 0; s < k; ++s) ho.out[s] = static_cast<float4*>(out[s]);
    launch_hand_out(c->vox_out.as<float4>(), ho, st);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipStreamSynchronize(st));           // like the single-cloud call: the outputs are complete on return
    return LISREG_OK;
}

// pcl's `*cloud += *other` for device records: K clouds end to end into `out` (one launch on the context's stream, nothing waited for —
// every later call of this context is ordered behind it).  currentCloudInit's surf source = dynamic + building + ground (:866-889).
int lisreg_concat_device(lisreg_ctx* c, int k, const void* const* in, const int* n, void* out, int* n_out)
{
    if (!c) return LISREG_ERR_ARG;
    if (k < 0 || k > kVoxelMultiMax || (k > 0 && (!in || !n))) return fail(c, LISREG_ERR_ARG, "concat_device: bad arguments (at most 8 clouds)");
    VoxelMulti m;
    BboxJobs jobs;
    memset(&m, 0, sizeof m); memset(&jobs, 0, sizeof jobs);
    m.k = jobs.k = k;
    long long total = 0;
    for (int s = 0; s < k; ++s) {
        if (n[s] < 0 || (n[s] > 0 && !in[s])) return fail(c, LISREG_ERR_ARG, "concat_device: NULL cloud with n > 0");
        m.off[s] = (int)total; jobs.pts[s] = static_cast<const float4*>(in[s]); jobs.n[s] = n[s];
        total += n[s];
    }
    if (total > 2000000000LL || (total > 0 && !out)) return fail(c, LISREG_ERR_ARG, "concat_device: bad output");
    m.off[k] = (int)total;
    if (n_out) *n_out = (int)total;
    HIPCHK(c, hipSetDevice(c->device));
    launch_concat_jobs(jobs, m, static_cast<float4*>(out), c->stream);
    HIPCHK(c, hipGetLastError());
    return LISREG_OK;
}

int lisreg_transform_cloud(lisreg_ctx* c, const void* in, int n, int stride, int fmt, const float T[6], void* out)
{
    if (!c) return LISREG_ERR_ARG;
    if (n < 0 || !T || (n > 0 && (!in || !out))) return fail(c, LISREG_ERR_ARG, "transform_cloud: bad arguments");
    if (fmt != LISREG_FMT_DEVICE && stride < 12) return fail(c, LISREG_ERR_ARG, "transform_cloud: bad stride");
    if (n == 0) return LISREG_OK;
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t st = c->stream;
    float M[12];
    pose_to_matrix_host(T, M);                       // pcl::getTransformation (common.cpp:140-142)
    if (fmt == LISREG_FMT_DEVICE) {                  // the matrix travels as a kernel argument
        launch_transform_cloud_m(static_cast<const float4*>(in), n, M, static_cast<float4*>(out), st);
        HIPCHK(c, hipGetLastError());
        HIPCHK(c, hipStreamSynchronize(st));
        return LISREG_OK;
    }
    HIPCHK(c, c->vox_M.ensure(sizeof M));
    HIPCHK(c, hipMemcpyAsync(c->vox_M.p, M, sizeof M, hipMemcpyHostToDevice, st));
    std::vector<float4> h((size_t)n);
    const unsigned char* b = static_cast<const unsigned char*>(in);
    for (int i = 0; i < n; ++i) { float v[3]; memcpy(v, b + (size_t)i * (size_t)stride, 12); h[(size_t)i] = make_float4(v[0], v[1], v[2], 0.f); }
    HIPCHK(c, c->vox_in.ensure(sizeof(float4) * (size_t)n));
    HIPCHK(c, hipMemcpyAsync(c->vox_in.p, h.data(), sizeof(float4) * (size_t)n, hipMemcpyHostToDevice, st));
    launch_transform_cloud(c->vox_in.as<float4>(), n, c->vox_M.as<float>(), c->vox_in.as<float4>(), st);
    HIPCHK(c, hipMemcpyAsync(h.data(), c->vox_in.p, sizeof(float4) * (size_t)n, hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipStreamSynchronize(st));
    unsigned char* o = static_cast<unsigned char*>(out);
    for (int i = 0; i < n; ++i) {
        if (o != b) memcpy(o + (size_t)i * (size_t)stride, b + (size_t)i * (size_t)stride, (size_t)stride);   // other fields copied
        memcpy(o + (size_t)i * (size_t)stride, &h[(size_t)i], 12);
    }
    return LISREG_OK;
}

// ---- §8 f-2: range-image projection + feature extraction ----------------------------------------------------------------
int lisreg_default_feature_params(lisreg_feature_params* p)
{
    if (!p) return LISREG_ERR_ARG;
    p->n_scan = 64; p->horizon_scan = 1800; p->downsample_rate = 2;        // config/params.yaml:68-72
    p->min_range = 0.0f; p->max_range = 70.0f;                            // :73-74
    p->edge_threshold = 1.0f; p->surf_threshold = 0.1f;                   // :117-118
    return LISREG_OK;
}

int lisreg_extract_features(lisreg_ctx* c, const void* cloud, int n, int stride, int fmt, const lisreg_feature_params* P,
                            lisreg_feature_out* out)
{
    return lisreg_extract_features_deskew(c, cloud, n, stride, fmt, P, nullptr, out);
}

int lisreg_extract_features_deskew(lisreg_ctx* c, const void* cloud, int n, int stride, int fmt, const lisreg_feature_params* P,
                                   const lisreg_deskew* dk, lisreg_feature_out* out)
{
    if (!c) return LISREG_ERR_ARG;
    if (!P || !out || n < 0 || (n > 0 && !cloud)) return fail(c, LISREG_ERR_ARG, "extract_features: bad arguments");
    const bool deskew = dk && dk->enabled && n > 0;
    if (deskew) {
        if (dk->imu_pointer_cur < 1 || dk->imu_pointer_cur > (1 << 20) || !dk->imu_time || !dk->imu_rot_x || !dk->imu_rot_y || !dk->imu_rot_z)
            return fail(c, LISREG_ERR_ARG, "extract_features: de-skew needs IMU tables with imu_pointer_cur >= 1");
        if (fmt == LISREG_FMT_DEVICE && !dk->time_device) return fail(c, LISREG_ERR_ARG, "extract_features: de-skew of device records needs time_device");
        if (fmt == LISREG_FMT_XYZIRT && stride < 28) return fail(c, LISREG_ERR_ARG, "extract_features: de-skew needs the time field (stride >= 28)");
    }
    if (fmt != LISREG_FMT_XYZIRT && fmt != LISREG_FMT_DEVICE) return fail(c, LISREG_ERR_ARG, "extract_features: fmt must be XYZIRT or DEVICE");
    if (fmt == LISREG_FMT_XYZIRT && stride < 22) return fail(c, LISREG_ERR_ARG, "extract_features: XYZIRT needs stride >= 22");
    if (P->n_scan < 1 || P->n_scan > 1024 || P->horizon_scan < 16 || P->horizon_scan > 4096 || P->downsample_rate < 1)
        return fail(c, LISREG_ERR_ARG, "extract_features: n_scan in [1,1024], horizon_scan in [16,4096], downsample_rate >= 1");
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t st = c->stream;
    const bool dev = fmt == LISREG_FMT_DEVICE;
    const int H = P->n_scan, W = P->horizon_scan, hw = H * W;
    const size_t L = (size_t)hw + 16;
    HIPCHK(c, c->ft_owner.ensure(sizeof(int) * (size_t)hw));      HIPCHK(c, c->ft_flag.ensure(sizeof(int) * L));
    HIPCHK(c, c->ft_pos.ensure(sizeof(int) * 2 * (L + 1)));       HIPCHK(c, c->ft_scan.ensure(sizeof(int) * (L / 2048 + 8)));
    HIPCHK(c, c->ft_col.ensure(sizeof(int) * L));                 HIPCHK(c, c->ft_range.ensure(sizeof(float) * L));
    HIPCHK(c, c->ft_src.ensure(sizeof(int) * L));                 HIPCHK(c, c->ft_curv.ensure(sizeof(float) * L));
    HIPCHK(c, c->ft_picked.ensure(sizeof(int) * L));              HIPCHK(c, c->ft_label.ensure(sizeof(int) * L));
    HIPCHK(c, c->ft_rlists.ensure(sizeof(int) * (size_t)H * 3 * 128));
    HIPCHK(c, c->ft_rcounts.ensure(sizeof(int) * (size_t)H * 4)); HIPCHK(c, c->ft_lists.ensure(sizeof(int) * 4 * L));
    HIPCHK(c, c->ft_counts.ensure(sizeof(int) * 8));
    FeatureBuffers fb;
    fb.owner = c->ft_owner.as<int>(); fb.flag = c->ft_flag.as<int>(); fb.pos = c->ft_pos.as<int>(); fb.scan_tmp = c->ft_scan.as<int>();
    fb.col = c->ft_col.as<int>(); fb.range = c->ft_range.as<float>(); fb.src = c->ft_src.as<int>(); fb.curv = c->ft_curv.as<float>();
    fb.picked = c->ft_picked.as<int>(); fb.label = c->ft_label.as<int>(); fb.ring_lists = c->ft_rlists.as<int>();
    fb.ring_counts = c->ft_rcounts.as<int>(); fb.lists = c->ft_lists.as<int>(); fb.counts = c->ft_counts.as<int>();
    // ---- stage the sweep -----------------------------------------------------------------------------------------
    const float4* pts = nullptr;
    const uint32_t* rings = nullptr;
    std::vector<float4> h_pts;
    std::vector<uint32_t> h_rings;
    if (dev) pts = static_cast<const float4*>(cloud);
    else if (n > 0) {
        h_pts.resize((size_t)n); h_rings.resize((size_t)n);
        const unsigned char* b = static_cast<const unsigned char*>(cloud);
        for (int i = 0; i < n; ++i) {
            const unsigned char* r = b + (size_t)i * (size_t)stride;
            float v[3], it = 0.f; uint16_t ring;
            memcpy(v, r, 12); memcpy(&it, r + 16, 4); memcpy(&ring, r + 20, 2);
            h_pts[(size_t)i] = make_float4(v[0], v[1], v[2], it); h_rings[(size_t)i] = ring;
        }
        HIPCHK(c, c->vox_in.ensure(sizeof(float4) * (size_t)n));
        HIPCHK(c, c->ft_rings.ensure(sizeof(uint32_t) * (size_t)n));
        HIPCHK(c, hipMemcpyAsync(c->vox_in.p, h_pts.data(), sizeof(float4) * (size_t)n, hipMemcpyHostToDevice, st));
        HIPCHK(c, hipMemcpyAsync(c->ft_rings.p, h_rings.data(), sizeof(uint32_t) * (size_t)n, hipMemcpyHostToDevice, st));
        pts = c->vox_in.as<float4>(); rings = c->ft_rings.as<uint32_t>();
    }
    launch_extract_features(pts, rings, n, *P, fb, st);
    HIPCHK(c, hipGetLastError());
    // ---- IMU de-skew: only the coordinates handed back change (ranges, columns and the selection use the raw points) ----
    const float4* out_pts = pts;
    std::vector<float4> h_dsk;
    if (deskew) {
        const size_t m = (size_t)dk->imu_pointer_cur + 1;
        HIPCHK(c, c->ft_dsk_tab.ensure(sizeof(double) * 4 * m + 64));
        HIPCHK(c, c->ft_dsk_pts.ensure(sizeof(float4) * (size_t)n));
        HIPCHK(c, c->ft_dsk_misc.ensure(64));
        double* tab = c->ft_dsk_tab.as<double>();
        const double* srcs[4] = { dk->imu_time, dk->imu_rot_x, dk->imu_rot_y, dk->imu_rot_z };
        for (int k = 0; k < 4; ++k) HIPCHK(c, hipMemcpyAsync(tab + (size_t)k * m, srcs[k], sizeof(double) * m, hipMemcpyHostToDevice, st));
        const float* times_dev = dk->time_device;
        std::vector<float> h_time;
        if (!dev) {
            h_time.resize((size_t)n);
            const unsigned char* b = static_cast<const unsigned char*>(cloud);
            for (int i = 0; i < n; ++i) memcpy(&h_time[(size_t)i], b + (size_t)i * (size_t)stride + 24, 4);
            HIPCHK(c, c->ft_dsk_time.ensure(sizeof(float) * (size_t)n));
            HIPCHK(c, hipMemcpyAsync(c->ft_dsk_time.p, h_time.data(), sizeof(float) * (size_t)n, hipMemcpyHostToDevice, st));
            times_dev = c->ft_dsk_time.as<float>();
        }
        HIPCHK(c, hipMemcpyAsync(c->ft_dsk_pts.p, pts, sizeof(float4) * (size_t)n, hipMemcpyDeviceToDevice, st));
        DeskewTables T{ tab, tab + m, tab + 2 * m, tab + 3 * m, dk->imu_pointer_cur, dk->time_scan_cur };
        launch_deskew(fb.owner, hw, times_dev, T, c->ft_dsk_misc.as<int>(), c->ft_dsk_misc.as<float>() + 4, c->ft_dsk_pts.as<float4>(), st);
        HIPCHK(c, hipGetLastError());
        HIPCHK(c, hipStreamSynchronize(st));               // h_time and the caller's tables are done with
        out_pts = c->ft_dsk_pts.as<float4>();
        if (!dev) {
            h_dsk.resize((size_t)n);
            HIPCHK(c, hipMemcpyAsync(h_dsk.data(), c->ft_dsk_pts.p, sizeof(float4) * (size_t)n, hipMemcpyDeviceToHost, st));
        }
    }
    int counts[8];
    HIPCHK(c, hipMemcpyAsync(counts, fb.counts, sizeof counts, hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipStreamSynchronize(st));
    // ---- hand the five clouds back in the caller's layout ------------------------------------------------------------
    struct Slot { void* buf; int cap; int* n; const int* idx; int cnt; };
    Slot slots[5] = { { out->deskewed, out->cap_deskewed, &out->n_deskewed, fb.src, counts[0] },
                      { out->corner, out->cap_corner, &out->n_corner, fb.lists + 0 * L, counts[1] },
                      { out->surface, out->cap_surface, &out->n_surface, fb.lists + 1 * L, counts[2] },
                      { out->corner_sharp, out->cap_corner_sharp, &out->n_corner_sharp, fb.lists + 2 * L, counts[3] },
                      { out->surface_sharp, out->cap_surface_sharp, &out->n_surface_sharp, fb.lists + 3 * L, counts[4] } };
    for (auto& sl : slots) *sl.n = sl.cnt;
    for (auto& sl : slots)
        if (sl.buf && sl.cnt > sl.cap) return fail(c, LISREG_ERR_ARG, "extract_features: an output buffer is too small (counts written back)");
    std::vector<int> h_idx;
    for (auto& sl : slots) {
        if (!sl.buf || sl.cnt == 0) continue;
        if (dev) launch_gather_points(out_pts, sl.idx, sl.cnt, static_cast<float4*>(sl.buf), st);
        else {
            h_idx.resize((size_t)sl.cnt);
            HIPCHK(c, hipMemcpyAsync(h_idx.data(), sl.idx, sizeof(int) * (size_t)sl.cnt, hipMemcpyDeviceToHost, st));
            HIPCHK(c, hipStreamSynchronize(st));
            const unsigned char* b = static_cast<const unsigned char*>(cloud);
            unsigned char* o = static_cast<unsigned char*>(sl.buf);
            for (int i = 0; i < sl.cnt; ++i) {
                memcpy(o + (size_t)i * (size_t)stride, b + (size_t)h_idx[(size_t)i] * (size_t)stride, (size_t)stride);
                if (deskew) memcpy(o + (size_t)i * (size_t)stride, &h_dsk[(size_t)h_idx[(size_t)i]], 12);     // newPoint.x/y/z (:451-456)
            }
        }
    }
    HIPCHK(c, hipStreamSynchronize(st));
    return LISREG_OK;
}

// S sweeps in one pass: the sweeps are stacked into ONE range image of S x H rows (grid of the selection kernel = sweeps x
// rings), every flat pass of the single-sweep pipeline runs once over the stack with per-sweep end guards, and one gather per
// output list hands every sweep its slice.  Device records in, device records out; the only host round trip is the (S + 1) x 5
// list boundaries the caller needs anyway.  Results are identical to S single calls (tests/test_features.py).
int lisreg_extract_features_batch(lisreg_ctx* c, int n_sweeps, const void* const* sweeps, const int* n, const lisreg_feature_params* P,
                                  lisreg_feature_out* outs)
{
    if (!c) return LISREG_ERR_ARG;
    if (n_sweeps < 0 || (n_sweeps > 0 && (!sweeps || !n || !outs)) || !P) return fail(c, LISREG_ERR_ARG, "extract_features_batch: bad arguments");
    if (n_sweeps == 0) return LISREG_OK;
    if (P->n_scan < 1 || P->n_scan > 1024 || P->horizon_scan < 16 || P->horizon_scan > 4096 || P->downsample_rate < 1)
        return fail(c, LISREG_ERR_ARG, "extract_features_batch: n_scan in [1,1024], horizon_scan in [16,4096], downsample_rate >= 1");
    if (n_sweeps > 256 || (long long)n_sweeps * P->n_scan > 32768) return fail(c, LISREG_ERR_ARG, "extract_features_batch: at most 256 sweeps and 32768 rows per call");
    std::vector<int> off((size_t)n_sweeps + 1, 0);
    for (int s = 0; s < n_sweeps; ++s) {
        if (n[s] < 0 || (n[s] > 0 && !sweeps[s])) return fail(c, LISREG_ERR_ARG, "extract_features_batch: NULL sweep with n > 0");
        if ((long long)off[(size_t)s] + n[s] > 2000000000LL) return fail(c, LISREG_ERR_ARG, "extract_features_batch: too many points");
        off[(size_t)s + 1] = off[(size_t)s] + n[s];
    }
    const int N = off[(size_t)n_sweeps];
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t st = c->stream;
    const int Hs = P->n_scan, W = P->horizon_scan, H = Hs * n_sweeps, hw = H * W, hw_sweep = Hs * W;
    const size_t L = (size_t)hw + 16;
    HIPCHK(c, c->ft_owner.ensure(sizeof(int) * (size_t)hw));      HIPCHK(c, c->ft_flag.ensure(sizeof(int) * L));
    HIPCHK(c, c->ft_pos.ensure(sizeof(int) * 2 * (L + 1)));       HIPCHK(c, c->ft_scan.ensure(sizeof(int) * (L / 2048 + 8)));
    HIPCHK(c, c->ft_col.ensure(sizeof(int) * L));                 HIPCHK(c, c->ft_range.ensure(sizeof(float) * L));
    HIPCHK(c, c->ft_src.ensure(sizeof(int) * L));                 HIPCHK(c, c->ft_curv.ensure(sizeof(float) * L));
    HIPCHK(c, c->ft_picked.ensure(sizeof(int) * L));              HIPCHK(c, c->ft_label.ensure(sizeof(int) * L));
    HIPCHK(c, c->ft_rlists.ensure(sizeof(int) * (size_t)H * 3 * 128));
    HIPCHK(c, c->ft_rcounts.ensure(sizeof(int) * (size_t)H * 4)); HIPCHK(c, c->ft_lists.ensure(sizeof(int) * 4 * L));
    HIPCHK(c, c->ft_counts.ensure(sizeof(int) * 8));
    HIPCHK(c, c->ft_cat.ensure(sizeof(float4) * (size_t)std::max(N, 1)));
    HIPCHK(c, c->ft_rings.ensure(sizeof(uint32_t) * (size_t)std::max(N, 1)));
    // layout: (S + 1) x 5 list boundaries, then — at the next 16-byte boundary — the 5 x S gather jobs (16 bytes each)
    const size_t jobs_off = (sizeof(int) * 5 * ((size_t)n_sweeps + 1) + 15) & ~(size_t)15;
    HIPCHK(c, c->ft_bounds.ensure(jobs_off + 16 * 5 * (size_t)n_sweeps));
    FeatureBuffers fb;
    fb.owner = c->ft_owner.as<int>(); fb.flag = c->ft_flag.as<int>(); fb.pos = c->ft_pos.as<int>(); fb.scan_tmp = c->ft_scan.as<int>();
    fb.col = c->ft_col.as<int>(); fb.range = c->ft_range.as<float>(); fb.src = c->ft_src.as<int>(); fb.curv = c->ft_curv.as<float>();
    fb.picked = c->ft_picked.as<int>(); fb.label = c->ft_label.as<int>(); fb.ring_lists = c->ft_rlists.as<int>();
    fb.ring_counts = c->ft_rcounts.as<int>(); fb.lists = c->ft_lists.as<int>(); fb.counts = c->ft_counts.as<int>();
    float4* cat = c->ft_cat.as<float4>();
    for (int s = 0; s < n_sweeps; ++s)
        if (n[s] > 0) HIPCHK(c, hipMemcpyAsync(cat + off[(size_t)s], sweeps[s], sizeof(float4) * (size_t)n[s], hipMemcpyDeviceToDevice, st));
    launch_feature_batch_rows(cat, N, off.data(), n_sweeps, Hs, P->downsample_rate, c->ft_rings.as<uint32_t>(), st);
    lisreg_feature_params Pst = *P;
    Pst.n_scan = H; Pst.downsample_rate = 1;                    // rows are stack rows; the ring filter was applied by k_feat_batch_rows
    launch_extract_features(cat, c->ft_rings.as<uint32_t>(), N, Pst, fb, st, n_sweeps);
    int* Bdev = c->ft_bounds.as<int>();
    launch_feature_batch_bounds(n_sweeps, Hs, hw_sweep, fb, hw, Bdev, st);
    HIPCHK(c, hipGetLastError());
    std::vector<int> B(5 * ((size_t)n_sweeps + 1));
    HIPCHK(c, hipMemcpyAsync(B.data(), Bdev, sizeof(int) * B.size(), hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipStreamSynchronize(st));
    // ---- per sweep: counts, capacity check, one gather job per list ---------------------------------------------------------
    struct Job { void* dst; int begin, count; };
    std::vector<Job> jobs(5 * (size_t)n_sweeps);
    int max_count[5] = { 0, 0, 0, 0, 0 };
    for (int s = 0; s < n_sweeps; ++s) {
        lisreg_feature_out& o = outs[s];
        // B columns: extracted, corner, corner_sharp, surface_sharp, surface
        const int cnt[5] = { B[(s + 1) * 5 + 0] - B[s * 5 + 0], B[(s + 1) * 5 + 1] - B[s * 5 + 1], B[(s + 1) * 5 + 4] - B[s * 5 + 4],
                             B[(s + 1) * 5 + 2] - B[s * 5 + 2], B[(s + 1) * 5 + 3] - B[s * 5 + 3] };      // deskewed, corner, surface, corner_sharp, surface_sharp
        const int beg[5] = { B[s * 5 + 0], B[s * 5 + 1], B[s * 5 + 4], B[s * 5 + 2], B[s * 5 + 3] };
        void* bufs[5] = { o.deskewed, o.corner, o.surface, o.corner_sharp, o.surface_sharp };
        const int caps[5] = { o.cap_deskewed, o.cap_corner, o.cap_surface, o.cap_corner_sharp, o.cap_surface_sharp };
        o.n_deskewed = cnt[0]; o.n_corner = cnt[1]; o.n_surface = cnt[2]; o.n_corner_sharp = cnt[3]; o.n_surface_sharp = cnt[4];
        for (int k = 0; k < 5; ++k) {
            if (bufs[k] && cnt[k] > caps[k]) return fail(c, LISREG_ERR_ARG, "extract_features_batch: an output buffer is too small (counts written back)");
            jobs[(size_t)k * n_sweeps + s] = Job{ bufs[k], beg[k], bufs[k] ? cnt[k] : 0 };
            if (bufs[k]) max_count[k] = std::max(max_count[k], cnt[k]);
        }
    }
    static_assert(sizeof(Job) == 16, "gather job = one 16-byte slot");
    Job* jobs_dev = reinterpret_cast<Job*>(reinterpret_cast<unsigned char*>(Bdev) + jobs_off);     // hipMalloc'ed base is 256-byte aligned
    HIPCHK(c, hipMemcpyAsync(jobs_dev, jobs.data(), sizeof(Job) * jobs.size(), hipMemcpyHostToDevice, st));
    const int* idx[5] = { fb.src, fb.lists + 0 * L, fb.lists + 1 * L, fb.lists + 2 * L, fb.lists + 3 * L };
    for (int k = 0; k < 5; ++k)
        launch_feature_batch_gather(cat, idx[k], jobs_dev + (size_t)k * n_sweeps, n_sweeps, max_count[k], st);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipStreamSynchronize(st));           // `jobs` is a local
    return LISREG_OK;
}

int lisreg_semantic_split(lisreg_ctx* c, const void* cloud, int n, int stride, int fmt, const uint32_t* using_label,
                          lisreg_semantic_out* out)
{
    if (!c) return LISREG_ERR_ARG;
    if (!out || n < 0 || (n > 0 && !cloud)) return fail(c, LISREG_ERR_ARG, "semantic_split: bad arguments");
    if (fmt != LISREG_FMT_XYZIL && fmt != LISREG_FMT_DEVICE) return fail(c, LISREG_ERR_ARG, "semantic_split: fmt must be XYZIL or DEVICE");
    if (fmt == LISREG_FMT_XYZIL && stride < 22) return fail(c, LISREG_ERR_ARG, "semantic_split: XYZIL needs stride >= 22");
    static const uint32_t kUsingLabel[32] = { 0, 10, 10, 10, 10, 10, 10, 10, 10, 40, 40, 40, 70, 50, 50, 70, 81, 70, 81, 81 };   // label.yaml:177-196
    const uint32_t* map = using_label ? using_label : kUsingLabel;
    for (int k = 0; k < 5; ++k) out->n[k] = 0;
    if (n == 0) return LISREG_OK;
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t st = c->stream;
    const bool dev = fmt == LISREG_FMT_DEVICE;
    const float4* pts = nullptr;
    const uint32_t* labels = nullptr;
    std::vector<float4> h_pts;
    std::vector<uint32_t> h_lab;
    if (dev) pts = static_cast<const float4*>(cloud);
    else {
        h_pts.resize((size_t)n); h_lab.resize((size_t)n);
        const unsigned char* b = static_cast<const unsigned char*>(cloud);
        for (int i = 0; i < n; ++i) {
            const unsigned char* r = b + (size_t)i * (size_t)stride;
            float v[3]; uint16_t l; memcpy(v, r, 12); memcpy(&l, r + 20, 2);
            h_pts[(size_t)i] = make_float4(v[0], v[1], v[2], 0.f); h_lab[(size_t)i] = l;
        }
        HIPCHK(c, c->vox_in.ensure(sizeof(float4) * (size_t)n));
        HIPCHK(c, c->vox_lab.ensure(sizeof(uint32_t) * (size_t)n));
        HIPCHK(c, hipMemcpyAsync(c->vox_in.p, h_pts.data(), sizeof(float4) * (size_t)n, hipMemcpyHostToDevice, st));
        HIPCHK(c, hipMemcpyAsync(c->vox_lab.p, h_lab.data(), sizeof(uint32_t) * (size_t)n, hipMemcpyHostToDevice, st));
        pts = c->vox_in.as<float4>(); labels = c->vox_lab.as<uint32_t>();
    }
    HIPCHK(c, c->vox_head.ensure(sizeof(int) * (5 * (size_t)n + 1)));
    HIPCHK(c, c->vox_slot.ensure(sizeof(int) * (5 * (size_t)n + 2)));
    HIPCHK(c, c->scan_tmp.ensure(sizeof(int) * (5 * (size_t)n / 2048 + 8)));
    HIPCHK(c, c->ft_lists.ensure(sizeof(int) * 5 * (size_t)n));
    HIPCHK(c, c->ft_counts.ensure(sizeof(int) * 8));
    launch_semantic_split(pts, labels, n, map, c->vox_head.as<int>(), c->vox_slot.as<int>(), c->scan_tmp.as<int>(),
                          c->ft_lists.as<int>(), c->ft_counts.as<int>(), st);
    HIPCHK(c, hipGetLastError());
    int counts[5];
    HIPCHK(c, hipMemcpyAsync(counts, c->ft_counts.p, sizeof counts, hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipStreamSynchronize(st));
    for (int k = 0; k < 5; ++k) out->n[k] = counts[k];
    for (int k = 0; k < 5; ++k)
        if (out->cloud[k] && counts[k] > out->cap[k]) return fail(c, LISREG_ERR_ARG, "semantic_split: an output buffer is too small (counts written back)");
    std::vector<int> h_idx;
    if (dev) {
        SemanticGather sg;
        for (int k = 0; k < 5; ++k) { sg.out[k] = static_cast<float4*>(out->cloud[k]); sg.count[k] = out->cloud[k] ? counts[k] : 0; }
        launch_semantic_gather(pts, c->ft_lists.as<int>(), n, sg, st);
    }
    for (int k = 0; k < 5; ++k) {
        if (dev || !out->cloud[k] || counts[k] == 0) continue;
        const int* idx = c->ft_lists.as<int>() + (size_t)k * n;
        {
            h_idx.resize((size_t)counts[k]);
            HIPCHK(c, hipMemcpyAsync(h_idx.data(), idx, sizeof(int) * (size_t)counts[k], hipMemcpyDeviceToHost, st));
            HIPCHK(c, hipStreamSynchronize(st));
            const unsigned char* b = static_cast<const unsigned char*>(cloud);
            unsigned char* o = static_cast<unsigned char*>(out->cloud[k]);
            for (int i = 0; i < counts[k]; ++i) memcpy(o + (size_t)i * (size_t)stride, b + (size_t)h_idx[(size_t)i] * (size_t)stride, (size_t)stride);
        }
    }
    HIPCHK(c, hipStreamSynchronize(st));
    return LISREG_OK;
}

// ---- RCCL pose gather: librccl is loaded lazily so single-GPU users never pay for it ------------
// ONE RCCL per process, and never in the global symbol scope.  A host process may carry an RCCL of its own already (a PyTorch wheel
// bundles librccl.so.1 next to ITS librocm_smi64 — soname .so.7, the system's is .so.1, so the loader keeps both): the copy already
// loaded is reused (RTLD_NOLOAD by soname); only a process without one gets the system's library, RTLD_LOCAL.  Round 3 loaded it
// RTLD_GLOBAL: the system librocm_smi64's globals then interposed those of the wheel's copy imported later, both static destructors
// freed the same std::map at exit, and glibc aborted the process ("double free or corruption", exit status 134) after every test had
// passed.  tests/test_teardown.py runs that sequence in a subprocess.
static void* rccl_dlopen()
{
    void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD);
    if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
    return h;
}

static int rccl_load(lisreg_ctx* c)
{
    if (c->rccl.handle) return LISREG_OK;
    void* h = rccl_dlopen();
    if (!h) return fail(c, LISREG_ERR_COMM, std::string("dlopen(librccl.so): ") + dlerror());
    c->rccl.handle = h;
    c->rccl.GetUniqueId = (int (*)(void*))dlsym(h, "ncclGetUniqueId");
    c->rccl.AllGather = (int (*)(const void*, void*, size_t, int, void*, hipStream_t))dlsym(h, "ncclAllGather");
    c->rccl.CommDestroy = (int (*)(void*))dlsym(h, "ncclCommDestroy");
    if (!c->rccl.GetUniqueId || !c->rccl.AllGather || !c->rccl.CommDestroy || !dlsym(h, "ncclCommInitRank"))
        return fail(c, LISREG_ERR_COMM, "librccl.so lacks the expected nccl* symbols");
    return LISREG_OK;
}

int lisreg_comm_unique_id(unsigned char id[128])
{
    if (!id) return LISREG_ERR_ARG;
    void* h = rccl_dlopen();                      // reference-counted by the loader; the library stays for the life of the process
    if (!h) return fail(nullptr, LISREG_ERR_COMM, "dlopen(librccl.so) failed");
    auto f = (int (*)(void*))dlsym(h, "ncclGetUniqueId");
    if (!f || f(id) != 0) return fail(nullptr, LISREG_ERR_COMM, "ncclGetUniqueId failed");
    return LISREG_OK;
}

namespace { struct UniqueId128 { char b[128]; }; }

int lisreg_comm_init(lisreg_ctx* c, int rank, int nranks, const unsigned char id[128])
{
    if (!c || !id || nranks < 1 || rank < 0 || rank >= nranks) return LISREG_ERR_ARG;
    int rc = rccl_load(c);
    if (rc) return rc;
    HIPCHK(c, hipSetDevice(c->device));
    UniqueId128 uid;
    memcpy(uid.b, id, 128);
    auto init = (int (*)(void**, int, UniqueId128, int))dlsym(c->rccl.handle, "ncclCommInitRank");
    if (init(&c->comm, nranks, uid, rank) != 0) return fail(c, LISREG_ERR_COMM, "ncclCommInitRank failed");
    c->comm_nranks = nranks;
    return LISREG_OK;
}

int lisreg_gather_results(lisreg_ctx* c, const void* local_device, int n_local, void* out_device)
{
    if (!c || !local_device || !out_device || n_local < 0) return LISREG_ERR_ARG;
    if (!c->comm) return fail(c, LISREG_ERR_COMM, "gather_results: call lisreg_comm_init first");
    const int ncclFloat32 = 7;
    if (c->rccl.AllGather(local_device, out_device, (size_t)n_local * kResultSize, ncclFloat32, c->comm, c->stream) != 0)
        return fail(c, LISREG_ERR_COMM, "ncclAllGather failed");
    return LISREG_OK;
}

void lisreg_comm_destroy(lisreg_ctx* c)
{
    if (!c || !c->comm) return;
    if (c->rccl.CommDestroy) c->rccl.CommDestroy(c->comm);
    c->comm = nullptr;
}

}  // extern "C"
