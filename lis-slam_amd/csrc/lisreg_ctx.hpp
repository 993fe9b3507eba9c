// lisreg_ctx.hpp — the context behind the C ABI (shared by the lisreg_api*.hip translation units; not installed).
#pragma once
#include <algorithm>
#include "lisreg_internal.hpp"

#include <atomic>
#include <map>
#include <string>
#include <vector>

namespace lisreg {

struct DevBuf {
    void*  p = nullptr;
    size_t cap = 0;
    hipError_t ensure(size_t bytes)
    {
        if (bytes <= cap) return hipSuccess;
        const size_t old_cap = cap;
        if (p) (void)hipFree(p);
        p = nullptr; cap = 0;
        // doubling: a buffer that follows a growing map (the sliding local map, the key-frame ring) is reallocated a handful of times, not
        // every few frames — hipFree waits for the device and hipMalloc costs 0.1-1 ms, which showed as spikes in the frame loops
        // the head-room is capped (a GB-scale buffer growing by a byte must not ask for twice itself), and a refused request falls back
        // to the size that is actually needed
        size_t want = std::max(bytes + bytes / 4 + 256, std::min(2 * old_cap, bytes + ((size_t)256 << 20)));
        hipError_t e = hipMalloc(&p, want);
        if (e != hipSuccess && want > bytes + 256) { (void)hipGetLastError(); want = bytes + 256; e = hipMalloc(&p, want); }
        if (e == hipSuccess) cap = want; else p = nullptr;
        return e;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
    template <class T> T* as() const { return reinterpret_cast<T*>(p); }
};

struct Target {
    bool      valid = false;
    int       n[2] = { 0, 0 };
    int       n_cells[2] = { 1, 1 };
    GridIndex g[2];
    lisreg::DevBuf    raw[2], sorted[2], cell_start[2];
    lisreg::DevBuf    nbr[2], nbr_meta[2];                // k-NN graph of the sorted points (search_mode 3)
    bool      graph_valid[2] = { false, false };
    lisreg::DevBuf    crow[2], crow_meta[2], crow_tab[2], crow_need[2], crow_omask[2], crow_scan[2], crow_scan_tmp[2];   // cell rows (search_mode 5)
    lisreg::DevBuf    crow_qmark[2], crow_reach[2];    // query marks of the running batch and their two-cell dilation (option "row_reach"; GridIndex::qmark)
    float     bbox[2][6] = { { 0 }, { 0 } };           // the cloud's bounding box (the grid is made from it, with or without a margin)
    int       grid_margin[2] = { 0, 0 };               // cells the grid reaches past the cloud on every side (the cell rows want two)
    int       omask_zero[2] = { 0, 0 };                // cells at the head of crow_omask known to be zero (launch_crow_classify / _build)
    int       crow_cap[2] = { 0, 0 };                  // rows allocated (= rows the classified index asked for when it was last sized)
    bool      crow_valid[2] = { false, false };
    bool      crow_chosen[2] = { false, false };       // a batch took the cell rows for this slot before: its next target is indexed with the margin at once
    bool      crow_too_big[2] = { false, false };      // the rows this target asks for exceed "cell_rows_max_mb" (auto: the batch takes the graph instead)
    bool      raw_external[2] = { false, false };     // LISREG_FMT_DEVICE: caller's memory, not ours
    const float4* raw_ptr[2] = { nullptr, nullptr };
    unsigned long long gen = 0;                        // bumped by every set_target of this slot (who built what is in here?)
};

// k = 1 search index over one cloud (lisreg_map_index_set)
struct MapIndex {
    bool      valid = false;
    int       n = 0, n_cells = 1;
    GridIndex g;
    DevBuf    raw, sorted, cell_start, g_dev;
    const float4* raw_ptr = nullptr;
};

// device-resident sliding local map (lisreg_api_localmap.hip)
struct LocalMap {
    bool   valid = false;
    DevBuf cls[5];                 // dynamic, pole, ground, building, outlier (map frame)
    int    n[5] = { 0, 0, 0, 0, 0 };
    DevBuf tgt[2];                 // corner / surf registration targets of the last extract
    int    n_tgt[2] = { 0, 0 };
    int    feature_point_num = 0;
    double bound[6] = { 0, 0, 0, 0, 0, 0 };
};

// the <= 19 newest key frames of the odometry node in the map frame (lisreg_api_localmap.hip: lisreg_keyframes_*)
struct KeyframeRing {
    bool valid = false;
    bool payload_is_label = false;  // fourth channel of the kept records: label (vote) or intensity (average) in the voxel grids
    struct Frame { DevBuf cloud[2]; int n[2] = { 0, 0 }; };     // corner, surf
    std::vector<Frame> frames;      // oldest first
    DevBuf cat[2], tgt[2];
    int    n_tgt[2] = { 0, 0 };
    // the target built by the last lisreg_keyframes_target: still current while no frame was pushed, the leaf sizes are the same and the
    // registration slot still holds it (the reference rebuilds the identical clouds and kd-trees for every sweep, :185-207)
    bool   built = false;
    float  built_leaf[2] = { 0.f, 0.f };
    int    built_slot = -1;
    unsigned long long built_gen = 0;
};

struct PackPool;           // host feeder threads (lisreg_api_feed.hip)
struct PackChunk { const unsigned char* src; lisreg_dpoint* dst; int n, stride, fmt; int pinned; };      // <= 64 k points of one host cloud; pinned: the DMA engine may read src

struct RcclApi {
    void* handle = nullptr;
    int (*GetUniqueId)(void*) = nullptr;
    int (*CommInitRank)(void**, int, const void* /* ncclUniqueId by value, 128 B */, int) = nullptr;
    int (*AllGather)(const void*, void*, size_t, int, void*, hipStream_t) = nullptr;
    int (*CommDestroy)(void*) = nullptr;
};

}  // namespace lisreg

struct lisreg_ctx {
    int          device = 0;
    hipStream_t  own_stream = nullptr, stream = nullptr;
    std::string  err;
    std::vector<lisreg::Target> targets;
    unsigned long long   target_gen = 0;
    lisreg::DevBuf       grids_dev;
    bool         grids_dirty = true;
    unsigned char* grids_host = nullptr;    // pinned staging of the GridIndex table (upload_grids does not wait for the stream)
    size_t       grids_host_cap = 0;
    hipEvent_t   grids_done = nullptr;      // the last upload has left the staging buffer
    // sort scratch (shared by target build and source sort; stream-ordered so reuse is safe)
    lisreg::DevBuf hist, bucket_start, scan_tmp, elem_bucket, elem_sub, tmp_bucket, tmp_sub, tmp_idx, tmp_pts, bbox_dev, bbox_scratch;
    // batch
    lisreg::DevBuf blocks, segs, items, sorted_all, order_all, partials, results, trace, src_upload, raw_upload, dbg_nn, blocks_q, coef, coef_ok, nn, counters, tseg_dev, tblk_dev, tchunk_dev, strip_tab, done_dev, xcd_tab,
           vox_in, vox_lab, vox_order, vox_sidx, vox_head, vox_slot, vox_start, vox_out, vox_outlab, vox_M,
           ft_owner, ft_flag, ft_pos, ft_scan, ft_col, ft_range, ft_src, ft_curv, ft_picked, ft_label, ft_rlists, ft_rcounts,
           ft_lists, ft_counts, ft_rings, ft_gather, ft_cat, ft_bounds, ft_dsk_tab, ft_dsk_pts, ft_dsk_misc, ft_dsk_time;
    // host feeder (lisreg_api_feed.hip): clouds packed to 16-byte records by a few threads into pinned staging, uploaded on a copy stream
    lisreg::PackPool* pack_pool = nullptr;
    int          feeder_threads = 8;
    unsigned char* pack_host[2] = { nullptr, nullptr };
    size_t       pack_cap[2] = { 0, 0 };
    lisreg::DevBuf pack_dev[2];
    int            feeder_engine = 1;                    // 0: the copy engine takes no chunks; 1: when idle (default); 2: whenever a packed chunk is not ready; 3: the same and one chunk up front, ready or not (tests)
    int            pack_stolen = 0, pack_chunks_n = 0;   // last lisreg_stage_host_items: chunks the copy engine took / all chunks
    lisreg::DevBuf pack_raw[2];                // structs that crossed the link as they are (chunks the copy engine took over), packed on the device
    unsigned char* up_host = nullptr;          // pinned staging of lisreg_upload_cloud
    size_t         up_cap = 0;
    hipEvent_t   pack_copied[2] = { nullptr, nullptr };   // the uploads into device buffer b are done (recorded on copy_stream)
    hipEvent_t   pack_free[2] = { nullptr, nullptr };     // the batch reading device buffer b has run (recorded on stream)
    hipEvent_t   pack_pending = nullptr;                  // uploads the next prepared batch has to wait for
    hipEvent_t   pack_raw_done = nullptr;                 // the copy engine's last read of the CALLER's pinned memory (chunks it took over)
    int          pack_flip = 0, pack_last = -1, pack_in_use = -1;
    hipStream_t  copy_stream = nullptr;
    hipStream_t  pack_stream = nullptr;                   // chunks the copy engine takes as they are: raw copy + k_pack_cloud (never on copy_stream: see lisreg_stage_host_items)
    hipEvent_t   pack_kernels_done = nullptr;             // recorded on pack_stream behind the last k_pack_cloud of a staging call
    std::vector<lisreg::PackChunk> pack_chunks;
    std::vector<std::atomic<int>> pack_done;
    std::map<int, lisreg::MapIndex> maps;             // by slot (sparse: the local maps keep theirs at 60000 + id)
    std::vector<lisreg::LocalMap> localmaps;
    std::vector<lisreg::KeyframeRing> keyrings;
    lisreg::DevBuf lm_in, lm_tmp, lm_bbox, exact_trig;
    lisreg::DevBuf map_stage;                          // lisreg_map_index_set_batch: host clouds of a batch, packed, in one upload
    lisreg::DevBuf mp_pts, mp_flag, mp_pos, mp_idx, mp_cnt, mp_d2, mp_out, icp_state, icp_partials, icp_cur, icp_items, map_tab, map_tsegs, map_tblocks;
    int*      done_host = nullptr;          // pinned
    unsigned char* stage_host = nullptr;    // pinned staging of the per-batch tables
    size_t    stage_cap = 0;
    hipEvent_t stage_done = nullptr;
    float*    fetch_host = nullptr;         // pinned landing area of results (+ trace)
    size_t    fetch_cap = 0;
    int       early_stop_chunk = -1;        // iterations between host looks at the finished-counter; -1 auto, 0 never
    int       fetch_trace_records = 0;      // lisreg_align: trace records copied out together with the results
    std::vector<lisreg::TargetSeg> h_tsegs;
    std::vector<lisreg::BlockDesc> h_tblocks;
    std::vector<lisreg::BlockDesc> h_tchunks;          // the same targets in chunks of kPartChunkHost points (strip form of the build)
    int       t_strips = 0, t_max_units = 0, t_max_ucells = 0;
    int       index_build = 2;                          // 0 bucket sort, 1 strip form (error if a grid does not fit it), 2 strip form whenever it fits
    int       strip_cells = 0, strip_cap = 2048;        // cells per strip aimed at (0: 1024 for a batch of one or two targets, else 2048); points per strip of the small-workgroup variant
    bool      strip_now = false;
    int         row_reach = 1;                          // option "row_reach": rows only for the cells the batch's queries come within two cells of (runs that rebuild their targets)
    bool        reach_ready = false;                    // lisreg_batch_prepare made the reach words of this batch's targets
    bool        xcd_cached = false;                     // lisreg_batch_prepare made the dispatch-order table of this batch (runs reuse it)
    int         reach_backoff = 0;                      // batches still to be prepared without the marks after a run that missed too often
    int         strip_zero_ints = 0;                    // leading ints of strip_tab known to be zero (a strip build hands its counters back clean)
    bool        items_reset = false;                    // the device registrations are in their start-of-run state (prepare, or the run before: launch_finalize)
    int         reach_miss_seen = 0, runs_since_fetch = 0;
    int         reach_miss_last = 0;                    // query-iterations of the last fetched run that found their cell without rows
    bool        reach_now = false;                      // the last run built its rows that way
    hipStream_t side_stream = nullptr;                  // strip build: the big-strip kernel runs here, forked from / joined to `stream`
    hipEvent_t  ev_fork = nullptr, ev_join = nullptr;
    hipEvent_t  ev_ab = nullptr, ev_ba = nullptr;       // interleaved runs: "half A's / half B's correspondence launch is through"
    int       t_elems = 0, t_buckets = 0;
    bool      count_searches = false;
    bool      dump_neighbors = false;   // tests: keep the five neighbour ids of every query of the last iteration run
    int       search_mode = 4;           // 0 LDS-staged box, 1 per-lane cell walk, 3 k-NN graph scan, 5 cell rows,
                                         // 4 auto: 3 when the prepared batch asks enough queries per target point to pay for the graph, else 1
    int       mode_now = 1;              // front-end of the prepared batch
    int       lanes_q = 1;               // lanes per query of the prepared batch (8 for small walk-mode batches)
    bool      lanes_per_query_auto = true;
    int       cell_min_ratio = 110;      // auto: query-iterations per target point from which the cell rows pay (they cost more than the graph to build and
                                         // halve the first iterations of a batch).  Round 6, rows filtered by the query marks: 16 scans against 200 k points
                                         // (92) draw, 24 (138) win 1.5 %, 32 (184) win 7 % (170 in rounds 4-5: 24 scans lost 4 %, 32 won 2 %)
    int       cell_rows_max_mb = 16384;  // auto: cell rows only while the targets' rows are expected to fit this (about 5 KB per target point)
    int       graph_min_ratio = 60;      // auto: query-iterations per target point from which the graph build pays (measured break-even ~55)
    int       interleave = 0;            // "interleave": big batches that run a fixed number of iterations are cut in two halves iterating on two
                                         // streams, each half's solves underneath the other half's correspondence launch (run_impl).  0 off (default),
                                         // 1 the halves' launches alternate through events, 2 free-running.  Measured on configs[1] (round 5):
                                         // 24.09 k reg/s off, 21.7 k alternating (a cross-stream hand-off costs more than the solve it hides),
                                         // 24.5 k free-running (+1.7 %, but two launches then share the chip and their durations stop being a
                                         // launch's own: 122 us per half against 98) — bit-identical results either way (tests)
    bool      interleaved_now = false;   // what the last run did
    int       interleave_min_blocks = 8192;   // workgroups from which a run is interleaved (each half should still fill the chip; tests lower it)
    int       xcd_order = 2;             // XCD-aware dispatch order of the correspondence launches: 0 off, 1 on (graph front-end), 2 auto (graph front-end, >= 32 registrations, >= 2048 blocks)
    bool      xcd_now = false;           // what the last run used
    int       feeder_numa = 1;           // packing threads bound to the CPUs of the device's NUMA node (those this process owns)
    int       feeder_node = -1, feeder_cpus = 0;       // what that found: the node, the CPUs bound to
    int       cell_anchor_until = 1;     // graph front-end: GN iterations 1 .. this also try an anchor out of the query's own grid column
    int       graph_hops = 3;            // neighbour lists scanned per query (anchor, then nearest found, ...) before the walk takes over
    int       graph_wide_until = 1;      // search_mode 3: GN iterations 0..this run the centre-first variant of the fall-back walk
    int       crow_wide_until = -1;      // search_mode 5: the same for the cell rows — never (1.4 % of the queries walk at iteration 0: the plain kernel's
                                         // iteration 0 takes 317 us against 330 with the centre-first variant, iteration 1 201 against 205; same neighbours)
    bool      canonical_ties = false;    // "canonical_ties" (always on with exact_arithmetic)
    bool      exact = false;             // "exact_arithmetic": the correspondence launches and the pose cache run the reference's arithmetic (lisreg_assoc.hip)
    int       sort_sources = 2;          // 0: keep the caller order, 1: 2-D column sort, 2: auto (probe the order at prepare time)
    bool      sort_now = false;          // decision for the prepared batch
    int       probe_items = -1, probe_elems = -1, probe_age = 0; bool probe_verdict = false;   // the batch shape the order was last probed on (auto): batches of a stream are alike
    float     first_pass_r = 0.45f;
    int       last_launches = 0;         // Gauss-Newton iterations the last fetched batch ran (its slowest item): where run_impl looks first
    int       wide_from = 0;
    int       wide_from_small = 1;       // the same for batches searched with eight lanes per query (single frames, sequential use): their
                                         // first guess is usually a frame step off, and the radius-limited first pass of the plain walk
                                         // then settles iteration 0 in 160-200 us instead of 230-310 (replay of configs[2]); 1 % slower on
                                         // a single frame with a 2-degree error (configs[0] stand-in)
    int       wide_until = 2;            // GN iterations wide_from..wide_until walk centre-first (no seeds, or seeds a pose step off)
    std::vector<lisreg::BlockDesc> h_blocks;      // 256-query workgroups: partial rows, sorts, probes
    std::vector<lisreg::BlockDesc> h_blocks_q;    // lanes_q > 1: kBlockQ / lanes_q queries per workgroup of the search kernel
    std::vector<lisreg::Segment>   h_segs;
    std::vector<lisreg::ItemState> h_items;
    std::vector<float>     h_results;
    int       n_items = 0, n_blocks = 0, n_segs = 0, n_elems = 0, n_buckets = 0, trace_cap = 0;
    lisreg::DevParams prm;
    lisreg_params params;
    bool      prepared = false;
    bool      rebuild_targets_each_run = false;
    std::vector<int> batch_slots;           // target slots used by the prepared batch
    int       degenerate = 0;               // isDegenerate member (odomEstimationNode.cpp:67)
    // profiling
    bool      profiling = false;
    std::vector<hipEvent_t> ev;
    std::vector<int>        ev_kind;        // kind of the interval STARTING at event i: 0 assoc, 1 solve, 2 index, -1 none
    std::vector<int>        ev_sidx;        // stream the event was recorded on: 0 the context's, 1 the side stream (interleaved runs)
    double    timing[5] = { 0, 0, 0, 0, 0 };
    // last align trace (host copy)
    std::vector<float> last_trace;
    int       last_trace_n = 0;
    // RCCL
    lisreg::RcclApi   rccl;
    void*     comm = nullptr;
    int       comm_nranks = 0;
};

namespace lisreg {
void feeder_destroy(lisreg_ctx* c);
int  ctx_fail(lisreg_ctx* c, int code, const std::string& msg);
// pack PCL structs (stride/format of common.h:9,25-35) into 16-B device records
void pack_cloud(const void* cloud, int n, int stride, int fmt, lisreg_dpoint* out);
// grid geometry from a bounding box; cell edge grows if the box would need too many cells
void make_grid(const float bb[6], int n, GridIndex* g, int* n_cells, int margin_cells = 0);
SortBuffers sort_buffers(lisreg_ctx* c);
int  ensure_sort_scratch(lisreg_ctx* c, size_t n_elems, size_t n_buckets);
void ctx_prof_mark(lisreg_ctx* c, int kind_of_next_interval);      // 0 correspondence kernel, 1 solve, 2 index build, -1 nothing
void ctx_prof_collect(lisreg_ctx* c);
}  // namespace lisreg

#define HIPCHK(c, expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) \
    return lisreg::ctx_fail((c), LISREG_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); } while (0)
