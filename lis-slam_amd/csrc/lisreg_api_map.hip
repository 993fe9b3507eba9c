// lisreg_api_map.hip — C-ABI entry points of SURVEY.md §8 f-3 (local-map maintenance, /root/reference/src/include/subMap.h):
// the k = 1 map index, the dynamic-point filter, the box crop and the cloud bounds.  Kernels: lisreg_nn1.hip.
// Every filter is "flag -> exclusive scan -> index list -> gather", so the survivors keep the input order like the
// reference's push_back loops.  No CPU fallback: without a HIP device these fail with LISREG_ERR_HIP.
#include "lisreg_ctx.hpp"

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstring>

using namespace lisreg;

namespace {

int bad(lisreg_ctx* c, const char* msg) { return ctx_fail(c, LISREG_ERR_ARG, msg); }

// the query / input cloud as 16-B device records: the caller's own memory for LISREG_FMT_DEVICE, else uploaded to mp_pts
int stage_cloud(lisreg_ctx* c, const void* cloud, int n, int stride, int fmt, const float4** out)
{
    if (fmt == LISREG_FMT_DEVICE) { *out = static_cast<const float4*>(cloud); return LISREG_OK; }
    std::vector<lisreg_dpoint> h((size_t)std::max(n, 1));
    pack_cloud(cloud, n, stride, fmt, h.data());
    HIPCHK(c, c->mp_pts.ensure(sizeof(float4) * (size_t)std::max(n, 1)));
    HIPCHK(c, hipMemcpyAsync(c->mp_pts.p, h.data(), sizeof(float4) * (size_t)n, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));      // h is a local
    *out = c->mp_pts.as<float4>();
    return LISREG_OK;
}

int check_cloud(lisreg_ctx* c, const void* cloud, int n, int stride, int fmt, const char* who)
{
    if (n < 0 || (n > 0 && !cloud)) return bad(c, (std::string(who) + ": NULL cloud with n > 0").c_str());
    if (fmt != LISREG_FMT_DEVICE && fmt != LISREG_FMT_XYZI && fmt != LISREG_FMT_XYZIL && fmt != LISREG_FMT_XYZIRT)
        return bad(c, (std::string(who) + ": unknown fmt").c_str());
    if (fmt != LISREG_FMT_DEVICE && stride < 12) return bad(c, (std::string(who) + ": stride < 12").c_str());
    if (fmt == LISREG_FMT_XYZIL && stride < 22) return bad(c, (std::string(who) + ": XYZIL needs stride >= 22").c_str());
    return LISREG_OK;
}

// flags (mp_flag) -> survivors of `cloud` in `out`, input layout and order
int emit_survivors(lisreg_ctx* c, const void* cloud, const float4* pts, int n, int stride, int fmt, void* out, int* n_out)
{
    hipStream_t st = c->stream;
    HIPCHK(c, c->mp_pos.ensure(sizeof(int) * ((size_t)n + 2)));
    HIPCHK(c, c->mp_idx.ensure(sizeof(int) * ((size_t)n + 1)));
    HIPCHK(c, c->mp_cnt.ensure(sizeof(int) * 4));
    HIPCHK(c, c->scan_tmp.ensure(sizeof(int) * ((size_t)n / 2048 + 8)));
    launch_compact(n, c->mp_flag.as<int>(), c->mp_pos.as<int>(), c->scan_tmp.as<int>(), c->mp_idx.as<int>(), c->mp_cnt.as<int>(), st);
    HIPCHK(c, hipGetLastError());
    int m = 0;
    HIPCHK(c, hipMemcpyAsync(&m, c->mp_cnt.p, sizeof m, hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipStreamSynchronize(st));
    *n_out = m;
    if (m == 0) return LISREG_OK;
    if (fmt == LISREG_FMT_DEVICE) {
        if (out == cloud) {                       // in place: gather through a scratch copy
            HIPCHK(c, c->mp_out.ensure(sizeof(float4) * (size_t)m));
            launch_gather_points(pts, c->mp_idx.as<int>(), m, c->mp_out.as<float4>(), st);
            HIPCHK(c, hipMemcpyAsync(out, c->mp_out.p, sizeof(float4) * (size_t)m, hipMemcpyDeviceToDevice, st));
        } else launch_gather_points(pts, c->mp_idx.as<int>(), m, static_cast<float4*>(out), st);
        HIPCHK(c, hipGetLastError());
        HIPCHK(c, hipStreamSynchronize(st));
        return LISREG_OK;
    }
    std::vector<int> idx((size_t)m);
    HIPCHK(c, hipMemcpyAsync(idx.data(), c->mp_idx.p, sizeof(int) * (size_t)m, hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipStreamSynchronize(st));
    const unsigned char* b = static_cast<const unsigned char*>(cloud);
    unsigned char* o = static_cast<unsigned char*>(out);
    for (int i = 0; i < m; ++i)                    // ascending indices: safe in place (memmove)
        memmove(o + (size_t)i * (size_t)stride, b + (size_t)idx[(size_t)i] * (size_t)stride, (size_t)stride);
    return LISREG_OK;
}

int copy_through(lisreg_ctx* c, const void* cloud, int n, int stride, int fmt, void* out)
{
    if (n == 0 || out == cloud) return LISREG_OK;
    if (fmt == LISREG_FMT_DEVICE) {
        HIPCHK(c, hipMemcpyAsync(out, cloud, sizeof(float4) * (size_t)n, hipMemcpyDeviceToDevice, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
    } else memmove(out, cloud, (size_t)n * (size_t)stride);
    return LISREG_OK;
}

}  // namespace

extern "C" {

int lisreg_map_index_set(lisreg_ctx* c, int slot, const void* cloud, int n, int stride, int fmt)
{
    if (!c) return LISREG_ERR_ARG;
    if (slot < 0 || slot > 65535) return bad(c, "map_index_set: bad slot");
    int rc = check_cloud(c, cloud, n, stride, fmt, "map_index_set");
    if (rc) return rc;
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t st = c->stream;
    MapIndex& m = c->maps[slot];
    m.valid = false;
    m.n = n;
    float bb[6] = { 0, 0, 0, 0, 0, 0 };
    if (fmt == LISREG_FMT_DEVICE) m.raw_ptr = static_cast<const float4*>(cloud);
    else {
        std::vector<lisreg_dpoint> h((size_t)std::max(n, 1));
        pack_cloud(cloud, n, stride, fmt, h.data());
        HIPCHK(c, m.raw.ensure(sizeof(float4) * (size_t)std::max(n, 1)));
        if (n > 0) HIPCHK(c, hipMemcpy(m.raw.p, h.data(), sizeof(float4) * (size_t)n, hipMemcpyHostToDevice));
        m.raw_ptr = m.raw.as<float4>();
    }
    if (n > 0) {
        HIPCHK(c, c->bbox_dev.ensure(sizeof(float) * 8));
        HIPCHK(c, c->bbox_scratch.ensure(sizeof(float) * 6 * 256));
        launch_bbox(m.raw_ptr, n, c->bbox_dev.as<float>(), c->bbox_scratch.as<float>(), st);
        HIPCHK(c, hipMemcpyAsync(bb, c->bbox_dev.p, sizeof bb, hipMemcpyDeviceToHost, st));
        HIPCHK(c, hipStreamSynchronize(st));
    }
    for (int d = 0; d < 6; ++d)
        if (n > 0 && !std::isfinite(bb[d])) return bad(c, "map_index_set: the cloud has infinite coordinates");
    if (n > 0 && !(bb[0] <= bb[3] && bb[1] <= bb[4] && bb[2] <= bb[5])) return bad(c, "map_index_set: the cloud has no finite point (every coordinate is NaN)");
    make_grid(bb, n, &m.g, &m.n_cells);
    HIPCHK(c, m.cell_start.ensure(sizeof(int) * ((size_t)m.n_cells + 2)));
    HIPCHK(c, m.sorted.ensure(sizeof(float4) * (size_t)std::max(n, 1)));
    rc = ensure_sort_scratch(c, (size_t)std::max(n, 1), (size_t)m.n_cells);
    if (rc) return rc;
    m.g.pts = m.sorted.as<float4>();
    m.g.cell_start = m.cell_start.as<int>();
    launch_build_target(m.raw_ptr, n, m.g, m.sorted.as<float4>(), m.cell_start.as<int>(), m.n_cells, sort_buffers(c), st);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, m.g_dev.ensure(sizeof(GridIndex)));
    HIPCHK(c, hipMemcpyAsync(m.g_dev.p, &m.g, sizeof(GridIndex), hipMemcpyHostToDevice, st));
    HIPCHK(c, hipStreamSynchronize(st));
    m.valid = true;
    return LISREG_OK;
}

// setInputTarget for every candidate of a loop-closure batch at once (subMapOptmizationNode.cpp:2793 inside the candidate loop): one
// bounding-box launch, one read-back, ONE bucket-sort launch sequence over all clouds — instead of n x (8 launches of a few microseconds
// and two host synchronisations).  Same index per map as lisreg_map_index_set, bit for bit (records by (cell, original index)).
int lisreg_map_index_set_batch(lisreg_ctx* c, int n_maps, const int* slots, const void* const* clouds, const int* counts, int stride, int fmt)
{
    if (!c) return LISREG_ERR_ARG;
    if (n_maps < 0 || (n_maps > 0 && (!slots || !clouds || !counts))) return bad(c, "map_index_set_batch: NULL argument");
    long long total = 0, total_cells = 0;
    for (int k = 0; k < n_maps; ++k) {
        if (slots[k] < 0 || slots[k] > 65535) return bad(c, "map_index_set_batch: bad slot");
        for (int j = 0; j < k; ++j) if (slots[j] == slots[k]) return bad(c, "map_index_set_batch: a slot is named twice");
        const int rc = check_cloud(c, clouds[k], counts[k], stride, fmt, "map_index_set_batch");
        if (rc) return rc;
        total += counts[k];
    }
    if (n_maps == 0) return LISREG_OK;
    if (total > 0x7fffffffLL / 2) return bad(c, "map_index_set_batch: more than 2^30 points in one batch");
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t st = c->stream;
    // The clouds in device memory: the caller's own records, or — host clouds — packed into ONE staging buffer of the context and uploaded
    // with one copy (the build below is the only reader; it is through when this call returns).
    std::vector<const float4*> src((size_t)n_maps, nullptr);
    if (fmt == LISREG_FMT_DEVICE) {
        for (int k = 0; k < n_maps; ++k) src[(size_t)k] = static_cast<const float4*>(clouds[k]);
    } else {
        std::vector<lisreg_dpoint> h((size_t)std::max<long long>(total, 1));
        long long off = 0;
        for (int k = 0; k < n_maps; ++k) { pack_cloud(clouds[k], counts[k], stride, fmt, h.data() + off); off += counts[k]; }
        HIPCHK(c, c->map_stage.ensure(sizeof(float4) * (size_t)std::max<long long>(total, 1)));
        if (total > 0) HIPCHK(c, hipMemcpy(c->map_stage.p, h.data(), sizeof(float4) * (size_t)total, hipMemcpyHostToDevice));
        off = 0;
        for (int k = 0; k < n_maps; ++k) { src[(size_t)k] = c->map_stage.as<float4>() + off; off += counts[k]; }
    }
    // bounding boxes of all clouds: one launch pair, one read-back
    std::vector<CloudRef> refs((size_t)n_maps);
    for (int k = 0; k < n_maps; ++k) refs[(size_t)k] = CloudRef{ src[(size_t)k], counts[k], 0 };
    HIPCHK(c, c->map_tab.ensure(sizeof(CloudRef) * (size_t)n_maps));
    HIPCHK(c, c->bbox_dev.ensure(sizeof(float) * 6 * (size_t)n_maps + 64));
    HIPCHK(c, c->bbox_scratch.ensure(sizeof(float) * 6 * 64 * (size_t)n_maps + 6 * 256 * sizeof(float)));
    HIPCHK(c, hipMemcpyAsync(c->map_tab.p, refs.data(), sizeof(CloudRef) * (size_t)n_maps, hipMemcpyHostToDevice, st));
    launch_bbox_refs(c->map_tab.as<CloudRef>(), n_maps, c->bbox_dev.as<float>(), c->bbox_scratch.as<float>(), st);
    HIPCHK(c, hipGetLastError());
    std::vector<float> bbs((size_t)n_maps * 6);
    HIPCHK(c, hipMemcpyAsync(bbs.data(), c->bbox_dev.p, sizeof(float) * 6 * (size_t)n_maps, hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipStreamSynchronize(st));            // (also: refs is a local)
    // Every cloud is checked and every grid laid out BEFORE any map of the batch is touched: an unusable cloud (infinite coordinates, no
    // finite point, too many cells) fails the call with every named slot — those that held an index before — as it was.
    std::vector<GridIndex> grids((size_t)n_maps);
    std::vector<int> ncells((size_t)n_maps, 1);
    for (int k = 0; k < n_maps; ++k) {
        const float* bb = &bbs[(size_t)k * 6];
        const int n = counts[k];
        float zero[6] = { 0, 0, 0, 0, 0, 0 };
        if (n > 0) {
            for (int d = 0; d < 6; ++d) if (!std::isfinite(bb[d])) return bad(c, "map_index_set_batch: a cloud has infinite coordinates");
            if (!(bb[0] <= bb[3] && bb[1] <= bb[4] && bb[2] <= bb[5])) return bad(c, "map_index_set_batch: a cloud has no finite point (every coordinate is NaN)");
        }
        memset(&grids[(size_t)k], 0, sizeof(GridIndex));
        make_grid(n > 0 ? bb : zero, n, &grids[(size_t)k], &ncells[(size_t)k]);
        total_cells += ncells[(size_t)k];
        if (total_cells > 0x7fffffffLL / 2) return bad(c, "map_index_set_batch: the grids of this batch have more than 2^30 cells in all — set them in smaller groups");
    }
    total_cells = 0;
    std::vector<MapIndex*> ms((size_t)n_maps);
    for (int k = 0; k < n_maps; ++k) { ms[(size_t)k] = &c->maps[slots[k]]; ms[(size_t)k]->valid = false; ms[(size_t)k]->n = counts[k]; ms[(size_t)k]->raw_ptr = src[(size_t)k]; }
    // one table for the batched build
    std::vector<TargetSeg> tsegs((size_t)n_maps);
    std::vector<BlockDesc> tblocks, tchunks;
    int tflat = 0, tstrip = 0, max_units = 0, max_ucells = 0;
    bool strips_fit = true;
    for (int k = 0; k < n_maps; ++k) {
        MapIndex& m = *ms[(size_t)k];
        const int n = counts[k];
        m.g = grids[(size_t)k]; m.n_cells = ncells[(size_t)k];
        HIPCHK(c, m.cell_start.ensure(sizeof(int) * ((size_t)m.n_cells + 2)));
        HIPCHK(c, m.sorted.ensure(sizeof(float4) * (size_t)std::max(n, 1)));
        HIPCHK(c, m.g_dev.ensure(sizeof(GridIndex)));
        m.g.pts = m.sorted.as<float4>();
        m.g.cell_start = m.cell_start.as<int>();
        TargetSeg& ts = tsegs[(size_t)k];
        memset(&ts, 0, sizeof ts);
        ts.raw = m.raw_ptr; ts.sorted_out = m.sorted.as<float4>(); ts.cell_start_out = m.cell_start.as<int>();
        ts.n = n; ts.n_cells = m.n_cells;
        ts.flat_base = tflat; ts.bucket_base = (int)total_cells;
        ts.ox = m.g.ox; ts.oy = m.g.oy; ts.oz = m.g.oz; ts.inv_cell = m.g.inv_cell;
        ts.nx = m.g.nx; ts.ny = m.g.ny; ts.nz = m.g.nz;
        ts.grid_id = k;
        // strip form of the build (the one a prepared batch's targets take, lisreg_batch_prepare): same layout rule
        ts.strip_base = tstrip;
        ts.ystrip = std::max(1, std::min(ts.ny, ((c->strip_cells > 0 ? c->strip_cells : 2048) + ts.nz / 2) / std::max(ts.nz, 1)));      // (map batches: many maps, the big strips)
        if (ts.nx > 0 && ts.nx <= kMaxStrips) {
            const int max_nstrips = std::max(1, kMaxStrips / ts.nx);
            ts.ystrip = std::max(ts.ystrip, (ts.ny + max_nstrips - 1) / max_nstrips);
        }
        ts.nstrips = (ts.ny + ts.ystrip - 1) / ts.ystrip;
        if (n > 0) {
            const int units = ts.nx * ts.nstrips;
            tstrip += units; max_units = std::max(max_units, units); max_ucells = std::max(max_ucells, ts.ystrip * ts.nz);
            strips_fit = strips_fit && units <= kMaxStrips;
        }
        for (int s0 = 0; s0 < n; s0 += kBlockQ) tblocks.push_back(BlockDesc{ k, s0, std::min(kBlockQ, n - s0), 0 });
        for (int s0 = 0; s0 < n; s0 += kPartChunkHost) tchunks.push_back(BlockDesc{ k, s0, std::min(kPartChunkHost, n - s0), 0 });
        tflat += n; total_cells += m.n_cells;
    }
    // Two or more maps whose grids fit it take the strip form (4.9 -> 1.2 ms for the 256 targets of 200 k points of bench.py --workload
    // cfg4_icp: LDS histograms instead of one random global atomic per point, profiles/r05_kernel_experiments.md section 12); same records,
    // same tables (tests/test_index_build.py, tests/test_mapfilter.py).  index_build 0 keeps the bucket sort.
    strips_fit = strips_fit && ((size_t)max_ucells + 1) * 4 + (size_t)c->strip_cap * 6 + 8192 <= kStripLdsLarge;
    const bool strips = strips_fit && c->index_build != 0 && n_maps >= 2 && tstrip > 0;
    int rc = ensure_sort_scratch(c, (size_t)std::max(tflat, 1), strips ? 1 : (size_t)std::max<long long>(total_cells, 1));
    if (rc) return rc;
    const std::vector<BlockDesc>& tb = strips ? tchunks : tblocks;
    HIPCHK(c, c->map_tsegs.ensure(sizeof(TargetSeg) * (size_t)n_maps));
    HIPCHK(c, c->map_tblocks.ensure(sizeof(BlockDesc) * std::max<size_t>(tb.size(), 1)));
    if (strips) { HIPCHK(c, c->strip_tab.ensure(sizeof(int) * (3 * ((size_t)tstrip + 4) + (size_t)tstrip / 2048 + 8))); c->strip_zero_ints = 0; }      // (this build leaves the counters as they are)
    HIPCHK(c, hipMemcpyAsync(c->map_tsegs.p, tsegs.data(), sizeof(TargetSeg) * (size_t)n_maps, hipMemcpyHostToDevice, st));
    if (!tb.empty()) HIPCHK(c, hipMemcpyAsync(c->map_tblocks.p, tb.data(), sizeof(BlockDesc) * tb.size(), hipMemcpyHostToDevice, st));
    for (int k = 0; k < n_maps; ++k)
        HIPCHK(c, hipMemcpyAsync(ms[(size_t)k]->g_dev.p, &ms[(size_t)k]->g, sizeof(GridIndex), hipMemcpyHostToDevice, st));
    if (strips) {
        StripBuffers sl;
        sl.cnt = c->strip_tab.as<int>(); sl.fill = sl.cnt + (tstrip + 1); sl.start = sl.fill + (tstrip + 1);
        sl.scan_tmp = sl.start + (tstrip + 2);
        sl.tmp_pts = c->tmp_pts.as<float4>(); sl.slot_idx = c->elem_bucket.as<uint32_t>(); sl.slot_pos = c->elem_sub.as<uint32_t>();
        if (!c->side_stream) {                    // created once; failure just means the two strip variants run back to back
            if (hipStreamCreateWithFlags(&c->side_stream, hipStreamNonBlocking) != hipSuccess) c->side_stream = nullptr;
            if (c->side_stream && (hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming) != hipSuccess ||
                                   hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming) != hipSuccess)) {
                (void)hipStreamDestroy(c->side_stream); c->side_stream = nullptr;
            }
        }
        sl.side = c->side_stream; sl.ev_fork = c->ev_fork; sl.ev_join = c->ev_join;
        for (int k = 0; k < n_maps; ++k)          // an empty cloud has no strip: its one-cell table is [0, 0]
            if (counts[k] <= 0) HIPCHK(c, hipMemsetAsync(ms[(size_t)k]->cell_start.p, 0, sizeof(int) * ((size_t)ms[(size_t)k]->n_cells + 1), st));
        if (launch_build_targets_strips(c->map_tblocks.as<BlockDesc>(), (int)tchunks.size(), c->map_tsegs.as<TargetSeg>(), n_maps, tstrip,
                                        max_units, max_ucells, c->strip_cap, sl, st))
            return ctx_fail(c, LISREG_ERR_HIP, "map_index_set_batch: strip index build: LDS configuration refused");
    } else
        launch_build_targets_batched(c->map_tblocks.as<BlockDesc>(), (int)tblocks.size(), c->map_tsegs.as<TargetSeg>(), n_maps, tflat,
                                     (int)std::max<long long>(total_cells, 1), sort_buffers(c), st);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipStreamSynchronize(st));            // the tables are locals
    for (int k = 0; k < n_maps; ++k) ms[(size_t)k]->valid = true;
    return LISREG_OK;
}

int lisreg_nearest(lisreg_ctx* c, int slot, const void* query, int n, int stride, int fmt, float max_dist, int* idx_out,
                   float* sqd_out)
{
    if (!c) return LISREG_ERR_ARG;
    if (slot < 0 || c->maps.count(slot) == 0 || !c->maps[slot].valid)
        return ctx_fail(c, LISREG_ERR_NO_TARGET, "nearest: no map index in this slot");
    int rc = check_cloud(c, query, n, stride, fmt, "nearest");
    if (rc) return rc;
    if (!(max_dist >= 0.f)) return bad(c, "nearest: max_dist must be >= 0");
    if (n > 0 && (!idx_out || !sqd_out)) return bad(c, "nearest: NULL output");
    if (n == 0) return LISREG_OK;
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t st = c->stream;
    const MapIndex& m = c->maps[slot];
    const float4* q = nullptr;
    rc = stage_cloud(c, query, n, stride, fmt, &q);
    if (rc) return rc;
    max_dist = std::min(max_dist, 1.8e19f);        // squared below
    if (fmt == LISREG_FMT_DEVICE) {
        launch_nn1(q, n, m.g_dev.as<GridIndex>(), max_dist, idx_out, sqd_out, st);
        HIPCHK(c, hipGetLastError());
        HIPCHK(c, hipStreamSynchronize(st));
        return LISREG_OK;
    }
    HIPCHK(c, c->mp_idx.ensure(sizeof(int) * (size_t)n));
    HIPCHK(c, c->mp_d2.ensure(sizeof(float) * (size_t)n));
    launch_nn1(q, n, m.g_dev.as<GridIndex>(), max_dist, c->mp_idx.as<int>(), c->mp_d2.as<float>(), st);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(idx_out, c->mp_idx.p, sizeof(int) * (size_t)n, hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipMemcpyAsync(sqd_out, c->mp_d2.p, sizeof(float) * (size_t)n, hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipStreamSynchronize(st));
    return LISREG_OK;
}

int lisreg_dynamic_filter(lisreg_ctx* c, int slot, const void* cloud, int n, int stride, int fmt, float center_radius,
                          float dist_thre_min, float dist_thre_max, float near_dist_thre, void* out, int* n_out)
{
    if (!c) return LISREG_ERR_ARG;
    if (slot < 0 || c->maps.count(slot) == 0 || !c->maps[slot].valid)
        return ctx_fail(c, LISREG_ERR_NO_TARGET, "dynamic_filter: no map index in this slot");
    int rc = check_cloud(c, cloud, n, stride, fmt, "dynamic_filter");
    if (rc) return rc;
    if (!n_out || (n > 0 && !out)) return bad(c, "dynamic_filter: NULL output");
    if (center_radius != center_radius || dist_thre_min != dist_thre_min || dist_thre_max != dist_thre_max || near_dist_thre != near_dist_thre)
        return bad(c, "dynamic_filter: NaN threshold");
    HIPCHK(c, hipSetDevice(c->device));
    const MapIndex& m = c->maps[slot];
    if (n <= 10 || m.n <= 0) {                      // subMap.h:1071-1072 returns false and leaves the cloud alone
        *n_out = n;
        rc = copy_through(c, cloud, n, stride, fmt, out);
        return rc ? rc : (n <= 10 ? LISREG_NOT_ENOUGH_FEATURES : LISREG_OK);
    }
    const float4* pts = nullptr;
    rc = stage_cloud(c, cloud, n, stride, fmt, &pts);
    if (rc) return rc;
    HIPCHK(c, c->mp_flag.ensure(sizeof(int) * ((size_t)n + 1)));
    launch_dynamic_flags(pts, n, m.g_dev.as<GridIndex>(), center_radius, near_dist_thre, dist_thre_min, dist_thre_max,
                         c->mp_flag.as<int>(), c->stream);
    HIPCHK(c, hipGetLastError());
    return emit_survivors(c, cloud, pts, n, stride, fmt, out, n_out);
}

int lisreg_bbx_filter(lisreg_ctx* c, const void* cloud, int n, int stride, int fmt, const double bounds[6], int delete_box,
                      void* out, int* n_out)
{
    if (!c) return LISREG_ERR_ARG;
    int rc = check_cloud(c, cloud, n, stride, fmt, "bbx_filter");
    if (rc) return rc;
    if (!bounds || !n_out || (n > 0 && !out)) return bad(c, "bbx_filter: NULL argument");
    *n_out = 0;
    if (n == 0) return LISREG_OK;
    HIPCHK(c, hipSetDevice(c->device));
    const float4* pts = nullptr;
    rc = stage_cloud(c, cloud, n, stride, fmt, &pts);
    if (rc) return rc;
    HIPCHK(c, c->mp_flag.ensure(sizeof(int) * ((size_t)n + 1)));
    launch_bbx_flags(pts, n, bounds, delete_box, c->mp_flag.as<int>(), c->stream);
    HIPCHK(c, hipGetLastError());
    return emit_survivors(c, cloud, pts, n, stride, fmt, out, n_out);
}

int lisreg_cloud_bounds(lisreg_ctx* c, const void* cloud, int n, int stride, int fmt, double bounds[6])
{
    if (!c) return LISREG_ERR_ARG;
    int rc = check_cloud(c, cloud, n, stride, fmt, "cloud_bounds");
    if (rc) return rc;
    if (!bounds) return bad(c, "cloud_bounds: NULL bounds");
    for (int d = 0; d < 3; ++d) { bounds[d] = DBL_MAX; bounds[3 + d] = -DBL_MAX; }
    if (n == 0) return LISREG_OK;
    HIPCHK(c, hipSetDevice(c->device));
    const float4* pts = nullptr;
    rc = stage_cloud(c, cloud, n, stride, fmt, &pts);
    if (rc) return rc;
    float bb[6];
    HIPCHK(c, c->bbox_dev.ensure(sizeof(float) * 8));
    HIPCHK(c, c->bbox_scratch.ensure(sizeof(float) * 6 * 256));
    launch_bbox(pts, n, c->bbox_dev.as<float>(), c->bbox_scratch.as<float>(), c->stream);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(bb, c->bbox_dev.p, sizeof bb, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    for (int d = 0; d < 6; ++d) bounds[d] = (double)bb[d];       // float extremes widened, as the reference's double min/max
    return LISREG_OK;
}

// ---- §8 f-4 ------------------------------------------------------------------------------------------------------
int lisreg_icp_default_params(int kind, lisreg_icp_params* p)
{
    if (!p || (kind != 0 && kind != 1)) return LISREG_ERR_ARG;
    memset(p, 0, sizeof *p);
    if (kind == 0) { p->max_corr_dist = 10; p->max_iters = 30; p->transformation_epsilon = 1e-4; p->euclidean_fitness_epsilon = 1e-4; }   // subMapOptmizationNode.cpp:2765-2769
    else { p->max_corr_dist = 0.2; p->max_iters = 50; p->transformation_epsilon = 1e-5; p->euclidean_fitness_epsilon = 1e-5; }             // :1445-1449, :4401-4405
    p->prev_mse = DBL_MAX;
    return LISREG_OK;
}

// ---- the ICP core: n alignments, each a source against the map index of ITS slot, one launch sequence per iteration ------------------
namespace {
struct IcpHostItem {
    const float4* src;      // device records
    int           n, slot;
    const float*  guess;    // row-major 4 x 4 or NULL
    double        prev_mse; // correspondences_prev_mse_ going in
    int           defer;    // the first MSE comparison is made by the caller (chained batch)
};

int icp_run(lisreg_ctx* c, const std::vector<IcpHostItem>& its, const lisreg_icp_params* P, std::vector<IcpState>& hs)
{
    hipStream_t st = c->stream;
    const int n_items = (int)its.size();
    hs.assign((size_t)n_items, IcpState());
    if (n_items == 0) return LISREG_OK;
    long long total = 0;
    for (const IcpHostItem& it : its) total += it.n;
    if (total > 0x7fffffffLL / 4) return bad(c, "icp: more than 2^29 source points in one batch");
    for (const IcpHostItem& it : its)
        if (c->maps[it.slot].n >= (1 << 28)) return bad(c, "icp: a target of 2^28 points or more (the search addresses its records by 32-bit byte offsets)");
    const int q = icp_batch_lanes(total);
    std::vector<IcpItem> hi((size_t)n_items + 1);          // + the end sentinel
    HIPCHK(c, c->icp_cur.ensure((sizeof(float4) + sizeof(int)) * (size_t)std::max<long long>(total, 1)));
    int* nn_base = reinterpret_cast<int*>(c->icp_cur.as<float4>() + std::max<long long>(total, 1));
    long long off = 0; int blk = 0;
    for (int k = 0; k < n_items; ++k) {
        const IcpHostItem& it = its[(size_t)k];
        IcpState& h = hs[(size_t)k];
        memset(&h, 0, sizeof h);
        for (int j = 0; j < 16; ++j) h.F[j] = it.guess ? it.guess[j] : ((j % 5 == 0) ? 1.f : 0.f);
        for (int j = 0; j < 16; ++j) h.Tm[j] = h.F[j];        // the first pass moves the working copy by the guess (icp.hpp:129-137)
        h.prev_mse = it.prev_mse; h.cur_mse = DBL_MAX; h.first_mse = -1.0; h.defer_first = it.defer;
        IcpItem& d = hi[(size_t)k];
        d.src = it.src; d.cur = c->icp_cur.as<float4>() + off; d.nn = nn_base + off; d.grid = c->maps[it.slot].g_dev.as<GridIndex>();
        d.n = it.n; d.blk0 = blk; d.nblk = icp_batch_blocks(it.n, q); d.pad_ = 0;
        off += it.n; blk += d.nblk;
    }
    memset(&hi[(size_t)n_items], 0, sizeof(IcpItem));
    hi[(size_t)n_items].blk0 = blk;
    const int total_blocks = blk;
    HIPCHK(c, c->icp_state.ensure(sizeof(IcpState) * (size_t)n_items + 64));
    HIPCHK(c, c->icp_items.ensure(sizeof(IcpItem) * 2 * ((size_t)n_items + 1)));      // the batch's table + its compacted form (below)
    HIPCHK(c, c->icp_partials.ensure(sizeof(double) * 17 * (size_t)std::max(total_blocks, 1)));
    IcpState* sd = c->icp_state.as<IcpState>();
    int* n_done_dev = reinterpret_cast<int*>(reinterpret_cast<char*>(c->icp_state.p) + sizeof(IcpState) * (size_t)n_items);
    HIPCHK(c, hipMemcpyAsync(sd, hs.data(), sizeof(IcpState) * (size_t)n_items, hipMemcpyHostToDevice, st));
    HIPCHK(c, hipMemcpyAsync(c->icp_items.p, hi.data(), sizeof(IcpItem) * ((size_t)n_items + 1), hipMemcpyHostToDevice, st));
    HIPCHK(c, hipMemsetAsync(n_done_dev, 0, sizeof(int), st));
    // (double) d2 > max_distance^2 rejects (correspondence_estimation.hpp): the largest float that still passes
    const double max_d2 = P->max_corr_dist * P->max_corr_dist;
    float cap2 = max_d2 >= 3.0e38 ? 3.0e38f : (float)max_d2;
    if ((double)cap2 > max_d2) cap2 = std::nextafterf(cap2, 0.f);
    const IcpItem* di = c->icp_items.as<IcpItem>();
    // Alignments end at different iterations (configs[3]: between the 8th and the 19th of at most 30), and a workgroup of a finished one
    // costs its launch slot (460 k empty wavefronts = 0.10 ms per launch).  Whenever the finished-counter read at the end of a chunk has
    // moved, the blocks of the alignments still iterating are renumbered end to end (finished ones get no block; same order, same blocks per
    // alignment, so the same partial rows in the same order: same sums) and the next launches shrink to them.  The fitness pass takes the
    // full table again.
    IcpItem* dc = c->icp_items.as<IcpItem>() + ((size_t)n_items + 1);
    const IcpItem* dcur = di;
    int cur_blocks = total_blocks;
    std::vector<IcpItem> hc;
    int n_done = 0, n_done_seen = 0;
    for (int it = 0; it < P->max_iters && n_done < n_items;) {
        // four iterations between two looks at the counter; two once alignments have begun to finish (a look costs ~30 us, a launch of
        // nothing but finished alignments 100)
        const int chunk = std::min(n_done > 0 ? 2 : 4, P->max_iters - it);
        for (int k = 0; k < chunk; ++k) {
            ctx_prof_mark(c, 0);
            launch_icp_assoc(dcur, n_items, cur_blocks, q, sd, cap2, c->icp_partials.as<double>(), st);
            ctx_prof_mark(c, 1);
            launch_icp_solve(dcur, n_items, q, sd, c->icp_partials.as<double>(), P->max_iters, P->transformation_epsilon,
                             P->euclidean_fitness_epsilon, n_done_dev, st);
            ctx_prof_mark(c, -1);
        }
        HIPCHK(c, hipGetLastError());
        it += chunk;
        HIPCHK(c, hipMemcpyAsync(&n_done, n_done_dev, sizeof n_done, hipMemcpyDeviceToHost, st));
        HIPCHK(c, hipStreamSynchronize(st));
        if (n_done > n_done_seen && n_done < n_items && it < P->max_iters) {
            n_done_seen = n_done;
            HIPCHK(c, hipMemcpyAsync(hs.data(), sd, sizeof(IcpState) * (size_t)n_items, hipMemcpyDeviceToHost, st));
            HIPCHK(c, hipStreamSynchronize(st));
            hc = hi;
            int b = 0;
            for (int k = 0; k < n_items; ++k) {
                hc[(size_t)k].blk0 = b;
                if (hs[(size_t)k].done) hc[(size_t)k].nblk = 0;
                b += hc[(size_t)k].nblk;
            }
            hc[(size_t)n_items].blk0 = b;
            HIPCHK(c, hipMemcpyAsync(dc, hc.data(), sizeof(IcpItem) * ((size_t)n_items + 1), hipMemcpyHostToDevice, st));
            HIPCHK(c, hipStreamSynchronize(st));      // (hc is pageable: the copy has left it when this returns)
            dcur = dc; cur_blocks = b;
        }
    }
    launch_icp_fitness_batch(di, n_items, total_blocks, q, sd, c->icp_partials.as<double>(), st);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(hs.data(), sd, sizeof(IcpState) * (size_t)n_items, hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipStreamSynchronize(st));
    ctx_prof_collect(c);
    return LISREG_OK;
}

void icp_fill_result(const IcpState& h, lisreg_icp_result* res)
{
    memcpy(res->final_transform, h.F, sizeof h.F);
    res->converged = h.converged; res->iters = h.iters; res->state = h.state; res->n_corr_last = h.n_corr;
    res->fitness = h.fit_n > 0 ? h.fit_sum / (double)h.fit_n : DBL_MAX;
    res->prev_mse = h.prev_mse;
}
}  // namespace

int lisreg_icp_align(lisreg_ctx* c, int slot, const void* source, int n, int stride, int fmt, const lisreg_icp_params* P,
                     const float* guess, lisreg_icp_result* res, void* aligned_out)
{
    if (!c) return LISREG_ERR_ARG;
    if (slot < 0 || c->maps.count(slot) == 0 || !c->maps[slot].valid)
        return ctx_fail(c, LISREG_ERR_NO_TARGET, "icp_align: no map index in this slot (setInputTarget)");
    int rc = check_cloud(c, source, n, stride, fmt, "icp_align");
    if (rc) return rc;
    if (!P || !res) return bad(c, "icp_align: NULL params / result");
    if (!(P->max_corr_dist >= 0) || P->max_iters < 1) return bad(c, "icp_align: bad max_corr_dist / max_iters");
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t st = c->stream;
    const float4* src = nullptr;
    rc = stage_cloud(c, source, n, stride, fmt, &src);
    if (rc) return rc;
    std::vector<IcpHostItem> its(1);
    its[0] = IcpHostItem{ src, n, slot, guess, P->prev_mse, 0 };
    std::vector<IcpState> hs;
    rc = icp_run(c, its, P, hs);
    if (rc) return rc;
    const IcpState& h = hs[0];
    icp_fill_result(h, res);
    if (aligned_out && n > 0) {                       // `output` of align(): the source under the final transformation
        HIPCHK(c, c->vox_M.ensure(sizeof(float) * 12));
        HIPCHK(c, hipMemcpyAsync(c->vox_M.p, h.F, sizeof(float) * 12, hipMemcpyHostToDevice, st));     // rows 0..2 of F = [R|t]
        if (fmt == LISREG_FMT_DEVICE) {
            launch_transform_cloud(src, n, c->vox_M.as<float>(), static_cast<float4*>(aligned_out), st);
            HIPCHK(c, hipStreamSynchronize(st));
        } else {
            HIPCHK(c, c->mp_out.ensure(sizeof(float4) * (size_t)n));
            launch_transform_cloud(src, n, c->vox_M.as<float>(), c->mp_out.as<float4>(), st);
            std::vector<float4> hp((size_t)n);
            HIPCHK(c, hipMemcpyAsync(hp.data(), c->mp_out.p, sizeof(float4) * (size_t)n, hipMemcpyDeviceToHost, st));
            HIPCHK(c, hipStreamSynchronize(st));
            const unsigned char* b = static_cast<const unsigned char*>(source);
            unsigned char* o = static_cast<unsigned char*>(aligned_out);
            for (int i = 0; i < n; ++i) {
                if (o != b) memcpy(o + (size_t)i * (size_t)stride, b + (size_t)i * (size_t)stride, (size_t)stride);
                memcpy(o + (size_t)i * (size_t)stride, &hp[(size_t)i], 12);
            }
        }
    }
    return LISREG_OK;
}

// The candidate loop of detectLoopClosureForSubMap (subMapOptmizationNode.cpp:2776-2840: setInputTarget / setInputSource / align /
// getFitnessScore / hasConverged / getFinalTransformation per candidate) as ONE call.
int lisreg_icp_align_batch(lisreg_ctx* c, const lisreg_icp_item* items, int n_items, int stride, int fmt, const lisreg_icp_params* P,
                           int chain_prev_mse, lisreg_icp_result* results)
{
    if (!c) return LISREG_ERR_ARG;
    if (n_items < 0 || (n_items > 0 && (!items || !results))) return bad(c, "icp_align_batch: NULL items / results");
    if (!P) return bad(c, "icp_align_batch: NULL params");
    if (!(P->max_corr_dist >= 0) || P->max_iters < 1) return bad(c, "icp_align_batch: bad max_corr_dist / max_iters");
    long long total = 0;
    for (int k = 0; k < n_items; ++k) {
        if (items[k].slot < 0 || c->maps.count(items[k].slot) == 0 || !c->maps[items[k].slot].valid)
            return ctx_fail(c, LISREG_ERR_NO_TARGET, "icp_align_batch: an item names a slot without a map index (setInputTarget)");
        const int rc = check_cloud(c, items[k].source, items[k].n, stride, fmt, "icp_align_batch");
        if (rc) return rc;
        total += items[k].n;
    }
    if (n_items == 0) return LISREG_OK;
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t st = c->stream;
    std::vector<IcpHostItem> its((size_t)n_items);
    if (fmt == LISREG_FMT_DEVICE) {
        for (int k = 0; k < n_items; ++k) its[(size_t)k].src = static_cast<const float4*>(items[k].source);
    } else {
        // all host sources through one packed upload
        std::vector<lisreg_dpoint> h((size_t)std::max<long long>(total, 1));
        long long off = 0;
        for (int k = 0; k < n_items; ++k) { pack_cloud(items[k].source, items[k].n, stride, fmt, h.data() + off); off += items[k].n; }
        HIPCHK(c, c->mp_pts.ensure(sizeof(float4) * (size_t)std::max<long long>(total, 1)));
        if (total > 0) HIPCHK(c, hipMemcpyAsync(c->mp_pts.p, h.data(), sizeof(float4) * (size_t)total, hipMemcpyHostToDevice, st));
        HIPCHK(c, hipStreamSynchronize(st));      // h is a local
        off = 0;
        for (int k = 0; k < n_items; ++k) { its[(size_t)k].src = c->mp_pts.as<float4>() + off; off += items[k].n; }
    }
    for (int k = 0; k < n_items; ++k) {
        IcpHostItem& it = its[(size_t)k];
        it.n = items[k].n; it.slot = items[k].slot; it.guess = items[k].guess;
        it.prev_mse = P->prev_mse;
        it.defer = (chain_prev_mse && k > 0) ? 1 : 0;
    }
    std::vector<IcpState> hs;
    int rc = icp_run(c, its, P, hs);
    if (rc) return rc;
    for (int k = 0; k < n_items; ++k) icp_fill_result(hs[(size_t)k], &results[k]);
    if (!chain_prev_mse) return LISREG_OK;
    // The reference's ICP object is `static`: DefaultConvergenceCriteria keeps correspondences_prev_mse_ from one align() to the next, so
    // candidate k's FIRST MSE comparison is against what candidate k - 1 left behind.  The batch ran items 1.. with that comparison
    // left out; here it is made, in order, with the values now known.  An item it would have stopped (a first-iteration MSE within the
    // absolute / relative bound of its predecessor's last one — rare) is aligned again alone with the right value going in.
    double carry = results[0].prev_mse;
    for (int k = 1; k < n_items; ++k) {
        const IcpState& h = hs[(size_t)k];
        if (h.first_mse < 0.0) { results[k].prev_mse = carry; continue; }      // ended before any MSE test: the value passes through
        const double diff = std::fabs(h.first_mse - carry);
        if (diff < 1e-12 || diff / carry < P->euclidean_fitness_epsilon) {
            std::vector<IcpHostItem> one(1, its[(size_t)k]);
            one[0].prev_mse = carry; one[0].defer = 0;
            std::vector<IcpState> h1;
            rc = icp_run(c, one, P, h1);
            if (rc) return rc;
            icp_fill_result(h1[0], &results[k]);
        }
        carry = results[k].prev_mse;
    }
    return LISREG_OK;
}

int lisreg_icp_gn_match(lisreg_ctx* c, int slot, const void* source, int n, int stride, int fmt, unsigned max_iterations,
                        float max_correspond_distance, const float predict_pose[16], lisreg_icpgn_result* res, void* transformed_out)
{
    if (!c) return LISREG_ERR_ARG;
    if (slot < 0 || c->maps.count(slot) == 0 || !c->maps[slot].valid)
        return ctx_fail(c, LISREG_ERR_NO_TARGET, "icp_gn_match: no map index in this slot (SetTargetCloud)");
    int rc = check_cloud(c, source, n, stride, fmt, "icp_gn_match");
    if (rc) return rc;
    if (!predict_pose || !res) return bad(c, "icp_gn_match: NULL predict_pose / result");
    if (max_iterations > 100000u) return bad(c, "icp_gn_match: max_iterations out of range");
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t st = c->stream;
    const MapIndex& m = c->maps[slot];
    const float4* src = nullptr;
    rc = stage_cloud(c, source, n, stride, fmt, &src);
    if (rc) return rc;
    IcpState h;
    memset(&h, 0, sizeof h);
    memcpy(h.F, predict_pose, sizeof h.F);
    const int nb = icp_blocks(n);
    HIPCHK(c, c->icp_state.ensure(sizeof(IcpState)));
    HIPCHK(c, c->icp_partials.ensure(sizeof(double) * 22 * (size_t)std::max(nb, 1)));
    HIPCHK(c, hipMemcpyAsync(c->icp_state.p, &h, sizeof h, hipMemcpyHostToDevice, st));
    IcpState* sd = c->icp_state.as<IcpState>();
    // registration.cpp:50 compares the SQUARED distance with max_correspond_distance itself
    const float cap2 = max_correspond_distance >= 0.f ? std::min(max_correspond_distance, 3.0e38f) : -1.f;
    for (unsigned it = 0; it < max_iterations; ++it) {
        if (cap2 >= 0.f) launch_icpgn_iteration(src, n, m.g_dev.as<GridIndex>(), sd, cap2, c->icp_partials.as<double>(), st);
        if ((it & 63u) == 63u) HIPCHK(c, hipStreamSynchronize(st));      // keep the queue bounded for long runs
    }
    HIPCHK(c, hipGetLastError());
    launch_icp_fitness(src, n, m.g_dev.as<GridIndex>(), sd, c->icp_partials.as<double>(), st);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(&h, sd, sizeof h, hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipStreamSynchronize(st));
    memcpy(res->final_transform, h.F, sizeof h.F);
    res->steps_applied = h.iters; res->n_corr_last = h.n_corr; res->reserved = 0;
    res->fitness = h.fit_n > 0 ? (float)(h.fit_sum / (double)h.fit_n) : FLT_MAX;
    if (transformed_out && n > 0) {
        HIPCHK(c, c->vox_M.ensure(sizeof(float) * 12));
        HIPCHK(c, hipMemcpyAsync(c->vox_M.p, h.F, sizeof(float) * 12, hipMemcpyHostToDevice, st));
        if (fmt == LISREG_FMT_DEVICE) {
            launch_transform_cloud(src, n, c->vox_M.as<float>(), static_cast<float4*>(transformed_out), st);
            HIPCHK(c, hipStreamSynchronize(st));
        } else {
            HIPCHK(c, c->mp_out.ensure(sizeof(float4) * (size_t)n));
            launch_transform_cloud(src, n, c->vox_M.as<float>(), c->mp_out.as<float4>(), st);
            std::vector<float4> hp((size_t)n);
            HIPCHK(c, hipMemcpyAsync(hp.data(), c->mp_out.p, sizeof(float4) * (size_t)n, hipMemcpyDeviceToHost, st));
            HIPCHK(c, hipStreamSynchronize(st));
            const unsigned char* b = static_cast<const unsigned char*>(source);
            unsigned char* o = static_cast<unsigned char*>(transformed_out);
            for (int i = 0; i < n; ++i) {
                if (o != b) memcpy(o + (size_t)i * (size_t)stride, b + (size_t)i * (size_t)stride, (size_t)stride);
                memcpy(o + (size_t)i * (size_t)stride, &hp[(size_t)i], 12);
            }
        }
    }
    return LISREG_OK;
}

}  // extern "C"
