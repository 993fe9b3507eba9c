// lisreg_api_localmap.hip — the device-resident sliding local map of SURVEY.md §8 f-3: SubMapManager::insert_local_map
// (/root/reference/src/include/subMap.h:979-1059) and SubMapOptmizationNode::extractSlidingCloud
// (src/node/subMapOptmizationNode.cpp:1369-1432) as ONE chain over 16-byte device records, ending in the registration
// target of scan2SubMapOptimization (:1509-1541).  Between frames the five class clouds never leave HBM; per call the host
// sees only counts and bounding boxes.  Every step is one of the primitives of lisreg_api.hip / lisreg_api_map.hip in
// LISREG_FMT_DEVICE form, in the reference's order:
//   insert : transformPointCloud of the frame's dynamic / pole / ground / building clouds (the outlier transform is commented
//            out in the reference, :1003, so nothing is appended to that class) -> optional map-based dynamic removal of the
//            frame's dynamic points against tree_dynamic (:1007-1026) -> append_feature -> feature_point_num -> bound
//   extract: cur_bbx moved by the current pose (transform_bbx) -> intersection with the map bound + 2 m -> in-place voxel
//            grids 0.1 / 0.05 / 0.4 / 0.2 / 0.6 -> bbx_filter of every class -> corner target = pole,
//            surf target = ground + building + dynamic (:1408-1419) -> both target indexes built.
// Also here: the key-frame ring of the odometry node (lisreg_keyframes_*), the same idea for odomEstimationNode's target.
// No CPU fallback: without a HIP device these fail with LISREG_ERR_HIP.
#include "lisreg_ctx.hpp"

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstring>
#include <string>

using namespace lisreg;

namespace {

constexpr int kMapSlotBase = 60000;      // lisreg_map_index_set slots 60000.. are the local maps' tree_dynamic

int bad(lisreg_ctx* c, const char* msg) { return ctx_fail(c, LISREG_ERR_ARG, msg); }

// grow a device buffer, keeping its first `keep` bytes
int grow_keep(lisreg_ctx* c, DevBuf& b, size_t bytes, size_t keep)
{
    if (bytes <= b.cap) return LISREG_OK;
    DevBuf nb;
    // a class cloud of a sliding map: start at 16 MB (a million records) and double — a reallocation in the frame loop is a device-wide wait
    // plus an allocation, 0.5 ms alone and tens of ms in a process that also hosts another HIP runtime user
    // (the head-room is a wish: if the device refuses it, the size that is needed will do)
    if (nb.ensure(std::max(std::max(bytes + bytes / 2, std::min(2 * b.cap, bytes + ((size_t)256 << 20))), (size_t)16 << 20)) != hipSuccess) {
        (void)hipGetLastError();
        HIPCHK(c, nb.ensure(bytes));
    }
    if (keep > 0 && b.p) HIPCHK(c, hipMemcpyAsync(nb.p, b.p, keep, hipMemcpyDeviceToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    b.release();
    b = nb;
    return LISREG_OK;
}

// bbx_filter of the five class clouds + the assembly of both registration targets as ONE launch sequence (extractSlidingCloud
// :1391-1419): flags over the concatenated index space, one scan, the survivors compacted class by class into scratch, then handed out —
// back into their class cloud (the map keeps the cropped classes) and into the target they belong to.  The host reads the six class
// boundaries once, after everything is enqueued.
struct CropClasses {
    const float4* in[5];        // class clouds (dynamic, pole, ground, building, outlier)
    float4*       out[5];       // the same buffers (the hand-out runs after the compaction has left them)
    int           off[6];       // class k = concatenated positions [off[k], off[k + 1])
    float4*       tgt[2];       // corner target = pole; surf target = ground, building, dynamic
    double        box[6];
};

__global__ __launch_bounds__(256) void k_crop_flags(CropClasses cc, int* __restrict__ flag)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= cc.off[5]) return;
    int k = 0;
#pragma unroll
    for (int j = 1; j < 5; ++j) if (i >= cc.off[j]) k = j;
    const float4 p = cc.in[k][i - cc.off[k]];
    // bbx_filter (subMap.h:1131-1144): float coordinate against double bounds, strictly inside
    const bool in = (double)p.x > cc.box[0] && (double)p.x < cc.box[3] && (double)p.y > cc.box[1] && (double)p.y < cc.box[4] &&
                    (double)p.z > cc.box[2] && (double)p.z < cc.box[5];
    flag[i] = in ? 1 : 0;
}

__global__ __launch_bounds__(256) void k_crop_compact(CropClasses cc, const int* __restrict__ flag, const int* __restrict__ pos,
                                                      float4* __restrict__ scratch, int* __restrict__ bounds_out)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < 6) bounds_out[i] = pos[cc.off[i]];                 // survivors before class i (pos has n + 1 entries)
    if (i >= cc.off[5] || !flag[i]) return;
    int k = 0;
#pragma unroll
    for (int j = 1; j < 5; ++j) if (i >= cc.off[j]) k = j;
    scratch[pos[i]] = cc.in[k][i - cc.off[k]];
}

__global__ __launch_bounds__(256) void k_crop_hand_out(CropClasses cc, const float4* __restrict__ scratch, const int* __restrict__ bounds)
{
    const int s = blockIdx.x * 256 + threadIdx.x;
    const int b0 = bounds[0], b1 = bounds[1], b2 = bounds[2], b3 = bounds[3], b4 = bounds[4], b5 = bounds[5];
    if (s >= b5) return;
    const float4 p = scratch[s];
    const int k = s >= b4 ? 4 : s >= b3 ? 3 : s >= b2 ? 2 : s >= b1 ? 1 : 0;
    const int base = k == 4 ? b4 : k == 3 ? b3 : k == 2 ? b2 : k == 1 ? b1 : b0;
    cc.out[k][s - base] = p;
    if (k == 1) cc.tgt[0][s - b1] = p;
    else if (k == 2) cc.tgt[1][s - b2] = p;
    else if (k == 3) cc.tgt[1][(b3 - b2) + (s - b3)] = p;
    else if (k == 0) cc.tgt[1][(b3 - b2) + (b4 - b3) + s] = p;
}

LocalMap* get_map(lisreg_ctx* c, int id, bool create)
{
    if (id < 0 || id > 1023) return nullptr;
    if ((size_t)id >= c->localmaps.size()) { if (!create) return nullptr; c->localmaps.resize((size_t)id + 1); }
    return &c->localmaps[(size_t)id];
}

void fill_info(const LocalMap& m, lisreg_localmap_info* info)
{
    if (!info) return;
    for (int k = 0; k < 5; ++k) info->n[k] = m.n[k];
    info->feature_point_num = m.feature_point_num;
    for (int d = 0; d < 6; ++d) info->bound[d] = m.bound[d];
    info->n_target_corner = m.n_tgt[0]; info->n_target_surf = m.n_tgt[1];
}

// get_cloud_bbx over the five classes (merge_feature_points + get_cloud_bbx_cpt, subMap.h:1047-1049): float extremes
// widened to double, {DBL_MAX, -DBL_MAX} when the map is empty
int update_bound(lisreg_ctx* c, LocalMap& m)
{
    for (int d = 0; d < 3; ++d) { m.bound[d] = DBL_MAX; m.bound[3 + d] = -DBL_MAX; }
    HIPCHK(c, c->lm_bbox.ensure(sizeof(float) * 6 * 5));
    HIPCHK(c, c->bbox_scratch.ensure(sizeof(float) * 6 * 64 * 5));
    float bb[30];
    bool any = false;
    BboxJobs jobs;
    memset(&jobs, 0, sizeof jobs);
    jobs.k = 5;
    for (int k = 0; k < 5; ++k) { jobs.pts[k] = m.cls[k].as<float4>(); jobs.n[k] = m.n[k]; any = any || m.n[k] > 0; }
    if (!any) return LISREG_OK;
    launch_bbox_jobs(jobs, c->lm_bbox.as<float>(), c->bbox_scratch.as<float>(), c->stream);     // five boxes, two launches
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(bb, c->lm_bbox.p, sizeof bb, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    for (int k = 0; k < 5; ++k)
        if (m.n[k] > 0)
            for (int d = 0; d < 3; ++d) {
                m.bound[d] = std::min(m.bound[d], (double)bb[6 * k + d]);
                m.bound[3 + d] = std::max(m.bound[3 + d], (double)bb[6 * k + 3 + d]);
            }
    return LISREG_OK;
}

}  // namespace

extern "C" {

int lisreg_localmap_default_params(lisreg_localmap_params* p)
{
    if (!p) return LISREG_ERR_ARG;
    memset(p, 0, sizeof *p);
    p->max_num_pts = 80000;                          // subMapOptmizationNode.cpp:605
    p->dynamic_removal_on = 1;                       // :608
    p->dynamic_removal_center_radius = 30.0f;        // :609
    p->dynamic_dist_thre_min = 0.3f;                 // :610
    p->dynamic_dist_thre_max = 3.0f;                 // :611
    p->near_dist_thre = 0.03f;                       // :612
    const float leaf[5] = { 0.1f, 0.05f, 0.4f, 0.2f, 0.6f };            // :1385-1389 (dynamic, pole, ground, building, outlier)
    const float box[6] = { -70.f, -70.f, -10.f, 70.f, 70.f, 20.f };     // :1377-1379
    memcpy(p->leaf, leaf, sizeof leaf);
    memcpy(p->crop_box, box, sizeof box);
    p->crop_pad = 2.0f;                              // :1384
    return LISREG_OK;
}

int lisreg_localmap_reset(lisreg_ctx* c, int map_id)
{
    if (!c) return LISREG_ERR_ARG;
    LocalMap* m = get_map(c, map_id, true);
    if (!m) return bad(c, "localmap_reset: bad map id");
    for (int k = 0; k < 5; ++k) m->n[k] = 0;
    m->n_tgt[0] = m->n_tgt[1] = 0;
    m->feature_point_num = 0;
    for (int d = 0; d < 3; ++d) { m->bound[d] = DBL_MAX; m->bound[3 + d] = -DBL_MAX; }
    m->valid = true;
    return LISREG_OK;
}

// insert_local_map (n_classes = 4: the outlier transform is commented out there, subMap.h:1003) and insert_submap (n_classes = 5,
// subMap.h:878-882) share everything up to the bound
static int insert_classes(lisreg_ctx* c, LocalMap* m, int map_id, const void* const clouds[5], const int n[5], int stride, int fmt,
                          const float pose[6], const lisreg_localmap_params* P, int n_classes)
{
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t st = c->stream;
    // the reference widens the band first (subMap.h:884 / :1006): max(dynamic_dist_thre_max, (float)(dynamic_dist_thre_min + 0.1))
    const float thre_max = std::max(P->dynamic_dist_thre_max, (float)((double)P->dynamic_dist_thre_min + 0.1));
    float Mpose[12] = { 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0 };
    if (pose) lisreg_pose_to_matrix(pose, Mpose);      // pcl::getTransformation, as lisreg_transform_cloud forms it
    for (int k = 0; k < n_classes; ++k) {            // dynamic, pole, ground, building (, outlier)
        if (n[k] == 0) continue;
        // stage the frame's class cloud as device records, then transformPointCloud(.., &pose) into lm_tmp
        const float4* src = nullptr;
        if (fmt == LISREG_FMT_DEVICE) src = static_cast<const float4*>(clouds[k]);
        else {
            const size_t bytes = (size_t)n[k] * (size_t)stride;
            HIPCHK(c, c->raw_upload.ensure(bytes + 32));
            HIPCHK(c, c->lm_in.ensure(sizeof(float4) * (size_t)n[k]));
            HIPCHK(c, hipMemcpyAsync(c->raw_upload.p, clouds[k], bytes, hipMemcpyHostToDevice, st));
            launch_pack_cloud(c->raw_upload.p, (size_t)n[k], stride, fmt == LISREG_FMT_XYZIL, c->lm_in.as<float4>(), st);
            HIPCHK(c, hipStreamSynchronize(st));     // pageable sources are free to change after the call
            src = c->lm_in.as<float4>();
        }
        int rc = LISREG_OK;
        const bool filtered = pose && k == 0 && P->dynamic_removal_on && m->feature_point_num > P->max_num_pts / 5;
        if (!filtered) {
            // transformPointCloud straight onto the end of the class cloud (append_feature, subMap.h:742-753); the stream orders it
            rc = grow_keep(c, m->cls[k], sizeof(float4) * ((size_t)m->n[k] + (size_t)n[k] + 1), sizeof(float4) * (size_t)m->n[k]);
            if (rc) return rc;
            if (pose) launch_transform_cloud_m(src, n[k], Mpose, m->cls[k].as<float4>() + m->n[k], st);
            else HIPCHK(c, hipMemcpyAsync(m->cls[k].as<float4>() + m->n[k], src, sizeof(float4) * (size_t)n[k], hipMemcpyDeviceToDevice, st));   // fisrt_submap: as it is
            HIPCHK(c, hipGetLastError());
            m->n[k] += n[k];
            continue;
        }
        HIPCHK(c, c->lm_tmp.ensure(sizeof(float4) * (size_t)n[k]));
        launch_transform_cloud_m(src, n[k], Mpose, c->lm_tmp.as<float4>(), st);
        HIPCHK(c, hipGetLastError());
        int n_add = n[k];
        {
            // tree_dynamic->setInputCloud(submap_dynamic) + map_scan_feature_pts_distance_removal (:889-892 / :1008-1012)
            rc = lisreg_map_index_set(c, kMapSlotBase + map_id, m->cls[0].p, m->n[0], 16, LISREG_FMT_DEVICE);
            if (rc) return rc;
            rc = lisreg_dynamic_filter(c, kMapSlotBase + map_id, c->lm_tmp.p, n[k], 16, LISREG_FMT_DEVICE,
                                       P->dynamic_removal_center_radius, P->dynamic_dist_thre_min, thre_max, P->near_dist_thre,
                                       c->lm_tmp.p, &n_add);
            if (rc != LISREG_OK && rc != LISREG_NOT_ENOUGH_FEATURES) return rc;
        }
        // append_feature (subMap.h:742-753)
        rc = grow_keep(c, m->cls[k], sizeof(float4) * ((size_t)m->n[k] + (size_t)n_add + 1), sizeof(float4) * (size_t)m->n[k]);
        if (rc) return rc;
        if (n_add > 0)
            HIPCHK(c, hipMemcpyAsync(m->cls[k].as<float4>() + m->n[k], c->lm_tmp.p, sizeof(float4) * (size_t)n_add, hipMemcpyDeviceToDevice, st));
        m->n[k] += n_add;
    }
    m->feature_point_num = m->n[0] + m->n[1] + m->n[2] + m->n[3] + m->n[4];
    return update_bound(c, *m);                       // ends with the call's one wait for the stream
}

static int check_insert_args(lisreg_ctx* c, const char* who, const void* const clouds[5], const int n[5], int stride, int fmt)
{
    if (fmt != LISREG_FMT_DEVICE && fmt != LISREG_FMT_XYZIL && fmt != LISREG_FMT_XYZI) return bad(c, (std::string(who) + ": unknown fmt").c_str());
    if (fmt != LISREG_FMT_DEVICE && (stride < 12 || (fmt == LISREG_FMT_XYZIL && stride < 22))) return bad(c, (std::string(who) + ": bad stride").c_str());
    for (int k = 0; k < 5; ++k) if (n[k] < 0 || (n[k] > 0 && !clouds[k])) return bad(c, (std::string(who) + ": NULL cloud with n > 0").c_str());
    return LISREG_OK;
}

int lisreg_localmap_insert(lisreg_ctx* c, int map_id, const void* const clouds[5], const int n[5], int stride, int fmt,
                           const float pose[6], const lisreg_localmap_params* P, lisreg_localmap_info* info)
{
    if (!c) return LISREG_ERR_ARG;
    if (!clouds || !n || !pose || !P) return bad(c, "localmap_insert: NULL argument");
    int rc = check_insert_args(c, "localmap_insert", clouds, n, stride, fmt);
    if (rc) return rc;
    LocalMap* m = get_map(c, map_id, true);
    if (!m) return bad(c, "localmap_insert: bad map id");
    if (!m->valid) { rc = lisreg_localmap_reset(c, map_id); if (rc) return rc; }
    rc = insert_classes(c, m, map_id, clouds, n, stride, fmt, pose, P, 4);
    if (rc) return rc;
    fill_info(*m, info);
    return LISREG_OK;
}

// ---- copy #3's side of SURVEY.md section 8 f-3: submap_t + SubMapManager::insert_submap + SubMapOptmizationNode::extractSubMapCloud ----------
// transform_bbx (subMap.h:214-228): the centre goes through the float matrix in double arithmetic, the box keeps its extents
static void transform_bbx(const double in[6], const float M[12], double out[6])
{
    double cp[3], cpo[3];
    for (int d = 0; d < 3; ++d) cp[d] = 0.5 * (in[d] + in[3 + d]);                       // local_cp = get_bound_cpt(local_bound)
    for (int r = 0; r < 3; ++r) cpo[r] = M[4 * r + 0] * cp[0] + M[4 * r + 1] * cp[1] + M[4 * r + 2] * cp[2] + M[4 * r + 3];
    for (int d = 0; d < 3; ++d) { out[3 + d] = in[3 + d] - cp[d] + cpo[d]; out[d] = in[d] - cp[d] + cpo[d]; }
}

// Eigen::Affine3f::inverse() of a pose matrix: linear part by the cofactor 3x3 inverse, t' = -L^-1 t (float)
static void affine_inverse(const float A[12], float Ai[12])
{
    const float a = A[0], b = A[1], cc = A[2], d = A[4], e = A[5], f = A[6], g = A[8], h = A[9], i = A[10];
    const float c00 = e * i - f * h, c01 = f * g - d * i, c02 = d * h - e * g;
    const float det = a * c00 + b * c01 + cc * c02, id = 1.f / det;
    const float L[9] = { c00 * id, (cc * h - b * i) * id, (b * f - cc * e) * id,
                         c01 * id, (a * i - cc * g) * id, (cc * d - a * f) * id,
                         c02 * id, (b * g - a * h) * id, (a * e - b * d) * id };
    for (int r = 0; r < 3; ++r) {
        Ai[4 * r] = L[3 * r]; Ai[4 * r + 1] = L[3 * r + 1]; Ai[4 * r + 2] = L[3 * r + 2];
        Ai[4 * r + 3] = -(L[3 * r] * A[3] + L[3 * r + 1] * A[7] + L[3 * r + 2] * A[11]);
    }
}

void lisreg_submap_crop_boxes(const double pre_local_bound[6], const float pre_pose[6], const double cur_local_bound[6],
                              const float cur_pose[6], float pad, double isect[6], double isect_local[6])
{
    float Mp[12], Mc[12], Mi[12];
    lisreg_pose_to_matrix(pre_pose, Mp);             // pclPointToAffine3f(preSubMap->submap_pose_6D_optimized) (:3988)
    lisreg_pose_to_matrix(cur_pose, Mc);             // trans2Affine3f(transformTobeMapped) (:3991)
    double pre[6], cur[6];
    transform_bbx(pre_local_bound, Mp, pre);
    transform_bbx(cur_local_bound, Mc, cur);
    for (int d = 0; d < 3; ++d) {                    // get_intersection_bbx(cur->bound, pre->bound, .., 10.0) (:3995)
        isect[d] = std::max(cur[d], pre[d]) - (double)pad;
        isect[3 + d] = std::min(cur[3 + d], pre[3 + d]) + (double)pad;
    }
    affine_inverse(Mc, Mi);                          // tran_map.inverse() (:4057), then transform_bbx of the intersection box (:4058)
    transform_bbx(isect, Mi, isect_local);
}

int lisreg_submap_insert(lisreg_ctx* c, int map_id, const void* const clouds[5], const int n[5], int stride, int fmt,
                         const float relative_pose[6], const float submap_pose[6], const lisreg_localmap_params* P, lisreg_submap_info* info)
{
    if (!c) return LISREG_ERR_ARG;
    if (!clouds || !n || !submap_pose || !P) return bad(c, "submap_insert: NULL argument");
    int rc = check_insert_args(c, "submap_insert", clouds, n, stride, fmt);
    if (rc) return rc;
    LocalMap* m = get_map(c, map_id, true);
    if (!m) return bad(c, "submap_insert: bad map id");
    if (!m->valid) { rc = lisreg_localmap_reset(c, map_id); if (rc) return rc; }
    rc = insert_classes(c, m, map_id, clouds, n, stride, fmt, relative_pose, P, 5);
    if (rc) return rc;
    if (info) {
        for (int k = 0; k < 5; ++k) info->n[k] = m->n[k];
        info->feature_point_num = m->feature_point_num;
        for (int d = 0; d < 6; ++d) info->local_bound[d] = m->bound[d];
        float M[12];
        lisreg_pose_to_matrix(submap_pose, M);       // pclPointToAffine3f(local_map->submap_pose_6D_optimized) (:968)
        transform_bbx(m->bound, M, info->bound);     // this->transform_bbx(local_bound, local_cp, bound, cp, tran_map) (:969)
    }
    return LISREG_OK;
}

int lisreg_submap_extract(lisreg_ctx* c, int pre_id, int cur_id, const float pre_pose[6], const float cur_pose[6], float pad,
                          float corner_leaf, float surf_leaf, int target_slot, lisreg_submap_extract_out* out)
{
    if (!c) return LISREG_ERR_ARG;
    if (!pre_pose || !cur_pose || !out) return bad(c, "submap_extract: NULL argument");
    LocalMap* pre = get_map(c, pre_id, false);
    LocalMap* cur = get_map(c, cur_id, false);
    if (!pre || !pre->valid || !cur || !cur->valid || pre == cur) return ctx_fail(c, LISREG_ERR_NO_TARGET, "submap_extract: no such submap pair");
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t st = c->stream;
    memset(out, 0, sizeof *out);
    lisreg_submap_crop_boxes(pre->bound, pre_pose, cur->bound, cur_pose, pad, out->isect, out->isect_local);
    // ---- target: the previous submap in the map frame, cropped (:3997-4020) ------------------------------------------------------
    // corner = pole; surf = ground + building + dynamic; transformPointCloud by submap_pose_6D_optimized is point-wise, so every
    // class is transformed straight into its place of the concatenation
    const int tn[2] = { pre->n[1], pre->n[2] + pre->n[3] + pre->n[0] };
    HIPCHK(c, pre->tgt[0].ensure(sizeof(float4) * (size_t)std::max(tn[0], 1)));
    HIPCHK(c, pre->tgt[1].ensure(sizeof(float4) * (size_t)std::max(tn[1], 1)));
    int rc;
    if (pre->n[1] > 0) { rc = lisreg_transform_cloud(c, pre->cls[1].p, pre->n[1], 16, LISREG_FMT_DEVICE, pre_pose, pre->tgt[0].p); if (rc) return rc; }
    {
        size_t off = 0;
        const int order[3] = { 2, 3, 0 };
        for (int j = 0; j < 3; ++j) {
            const int k = order[j];
            if (pre->n[k] > 0) { rc = lisreg_transform_cloud(c, pre->cls[k].p, pre->n[k], 16, LISREG_FMT_DEVICE, pre_pose, pre->tgt[1].as<float4>() + off); if (rc) return rc; }
            off += (size_t)pre->n[k];
        }
    }
    for (int k = 0; k < 2; ++k) {
        pre->n_tgt[k] = tn[k];
        if (tn[k] > 0) { rc = lisreg_bbx_filter(c, pre->tgt[k].p, tn[k], 16, LISREG_FMT_DEVICE, out->isect, 0, pre->tgt[k].p, &pre->n_tgt[k]); if (rc) return rc; }
    }
    // ---- sources: the current submap in its own frame, cropped by the box moved into that frame, voxel grids 0.2 / 0.5 (:4030-4067) -----
    // corner = pole; surf = dynamic + ground + building
    const int sn[2] = { cur->n[1], cur->n[0] + cur->n[2] + cur->n[3] };
    HIPCHK(c, cur->tgt[0].ensure(sizeof(float4) * (size_t)std::max(sn[0], 1)));
    HIPCHK(c, cur->tgt[1].ensure(sizeof(float4) * (size_t)std::max(sn[1], 1)));
    if (cur->n[1] > 0) HIPCHK(c, hipMemcpyAsync(cur->tgt[0].p, cur->cls[1].p, sizeof(float4) * (size_t)cur->n[1], hipMemcpyDeviceToDevice, st));
    {
        size_t off = 0;
        const int order[3] = { 0, 2, 3 };
        for (int j = 0; j < 3; ++j) {
            const int k = order[j];
            if (cur->n[k] > 0) HIPCHK(c, hipMemcpyAsync(cur->tgt[1].as<float4>() + off, cur->cls[k].p, sizeof(float4) * (size_t)cur->n[k], hipMemcpyDeviceToDevice, st));
            off += (size_t)cur->n[k];
        }
    }
    const float leaf[2] = { corner_leaf, surf_leaf };
    for (int k = 0; k < 2; ++k) {
        int nk = sn[k];
        if (nk > 0) { rc = lisreg_bbx_filter(c, cur->tgt[k].p, nk, 16, LISREG_FMT_DEVICE, out->isect_local, 0, cur->tgt[k].p, &nk); if (rc) return rc; }
        if (nk > 0) {                                  // voxel_downsample_pcl(Last, LastDS, leaf): an empty cloud returns false and stays empty
            HIPCHK(c, c->lm_tmp.ensure(sizeof(float4) * (size_t)nk));
            int nv = 0;
            rc = lisreg_voxel_downsample(c, cur->tgt[k].p, nk, 16, LISREG_FMT_DEVICE, leaf[k], c->lm_tmp.p, nk, &nv);
            if (rc != LISREG_OK && rc != LISREG_LEAF_TOO_SMALL) return rc;
            HIPCHK(c, hipMemcpyAsync(cur->tgt[k].p, c->lm_tmp.p, sizeof(float4) * (size_t)nv, hipMemcpyDeviceToDevice, st));
            nk = nv;
        }
        cur->n_tgt[k] = nk;
    }
    HIPCHK(c, hipStreamSynchronize(st));
    if (target_slot >= 0) {                            // kdtree{Corner,Surf}FromSubMap->setInputCloud (:4496-4497)
        rc = lisreg_set_target_slot(c, target_slot, pre->tgt[0].p, pre->n_tgt[0], pre->tgt[1].p, pre->n_tgt[1], 16, LISREG_FMT_DEVICE);
        if (rc) return rc;
    }
    out->n_target_corner = pre->n_tgt[0]; out->n_target_surf = pre->n_tgt[1];
    out->src_corner = cur->tgt[0].p; out->n_src_corner = cur->n_tgt[0];
    out->src_surf = cur->tgt[1].p; out->n_src_surf = cur->n_tgt[1];
    return LISREG_OK;
}

int lisreg_localmap_extract(lisreg_ctx* c, int map_id, const float cur_pose[6], const lisreg_localmap_params* P, int target_slot,
                            lisreg_localmap_info* info)
{
    if (!c) return LISREG_ERR_ARG;
    if (!cur_pose || !P) return bad(c, "localmap_extract: NULL argument");
    LocalMap* m = get_map(c, map_id, false);
    if (!m || !m->valid) return ctx_fail(c, LISREG_ERR_NO_TARGET, "localmap_extract: no such local map");
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t st = c->stream;
    // cur_bbx / get_bound_cpt / transform_bbx (:1376-1382, subMap.h:122-127, 235-249): double boxes, float matrix entries
    float M[12];
    lisreg_pose_to_matrix(cur_pose, M);
    double bx[6], cp[3], cpo[3];
    for (int d = 0; d < 6; ++d) bx[d] = (double)P->crop_box[d];
    for (int d = 0; d < 3; ++d) cp[d] = 0.5 * (bx[d] + bx[3 + d]);
    for (int r = 0; r < 3; ++r) cpo[r] = M[4 * r + 0] * cp[0] + M[4 * r + 1] * cp[1] + M[4 * r + 2] * cp[2] + M[4 * r + 3];
    double cur[6];
    for (int d = 0; d < 3; ++d) { cur[3 + d] = bx[3 + d] - cp[d] + cpo[d]; cur[d] = bx[d] - cp[d] + cpo[d]; }
    // get_intersection_bbx(cur_bbx, localMap->bound, bbx_intersection, 2.0) (subMap.h:176-184)
    double isect[6];
    for (int d = 0; d < 3; ++d) {
        isect[d] = std::max(cur[d], m->bound[d]) - (double)P->crop_pad;
        isect[3 + d] = std::min(cur[3 + d], m->bound[3 + d]) + (double)P->crop_pad;
    }
    // voxel_downsample_pcl(cls, cls, leaf) in place (:1385-1389; an empty class returns false and stays empty): the five grids as ONE
    // launch sequence (lisreg_voxel_downsample_multi), then bbx_filter
    {
        const void* vin[5]; void* vout[5]; int vn[5], vcap[5], vno[5];
        for (int k = 0; k < 5; ++k) { vin[k] = m->cls[k].p; vout[k] = m->cls[k].p; vn[k] = m->n[k]; vcap[k] = m->n[k]; vno[k] = 0; }
        int rc = lisreg_voxel_downsample_multi(c, 5, vin, vn, P->leaf, LISREG_FMT_DEVICE, vout, vcap, vno);
        if (rc) return rc;
        for (int k = 0; k < 5; ++k) m->n[k] = vno[k];
    }
    // bbx_filter of every class, then laserCloudCornerFromSubMap = pole; laserCloudSurfFromSubMap = ground + building + dynamic (:1408-1419)
    {
        CropClasses cc;
        int total = 0;
        for (int k = 0; k < 5; ++k) { cc.in[k] = m->cls[k].as<float4>(); cc.out[k] = m->cls[k].as<float4>(); cc.off[k] = total; total += m->n[k]; }
        cc.off[5] = total;
        for (int d = 0; d < 6; ++d) cc.box[d] = isect[d];
        HIPCHK(c, m->tgt[0].ensure(sizeof(float4) * (size_t)std::max(m->n[1], 1)));
        HIPCHK(c, m->tgt[1].ensure(sizeof(float4) * (size_t)std::max(m->n[2] + m->n[3] + m->n[0], 1)));
        cc.tgt[0] = m->tgt[0].as<float4>(); cc.tgt[1] = m->tgt[1].as<float4>();
        int bounds[6] = { 0, 0, 0, 0, 0, 0 };
        if (total > 0) {
            HIPCHK(c, c->mp_flag.ensure(sizeof(int) * ((size_t)total + 1)));
            HIPCHK(c, c->mp_pos.ensure(sizeof(int) * ((size_t)total + 2)));
            HIPCHK(c, c->mp_cnt.ensure(sizeof(int) * 8));
            HIPCHK(c, c->mp_out.ensure(sizeof(float4) * (size_t)total));
            HIPCHK(c, c->scan_tmp.ensure(sizeof(int) * ((size_t)total / 2048 + 8)));
            const unsigned nb = (unsigned)((total + 255) / 256);
            k_crop_flags<<<nb, 256, 0, st>>>(cc, c->mp_flag.as<int>());
            launch_exclusive_scan(c->mp_flag.as<int>(), c->mp_pos.as<int>(), c->scan_tmp.as<int>(), total, st);
            k_crop_compact<<<nb, 256, 0, st>>>(cc, c->mp_flag.as<int>(), c->mp_pos.as<int>(), c->mp_out.as<float4>(), c->mp_cnt.as<int>());
            k_crop_hand_out<<<nb, 256, 0, st>>>(cc, c->mp_out.as<float4>(), c->mp_cnt.as<int>());
            HIPCHK(c, hipGetLastError());
            HIPCHK(c, hipMemcpyAsync(bounds, c->mp_cnt.p, sizeof bounds, hipMemcpyDeviceToHost, st));
        }
        HIPCHK(c, hipStreamSynchronize(st));
        for (int k = 0; k < 5; ++k) m->n[k] = bounds[k + 1] - bounds[k];
    }
    m->n_tgt[0] = m->n[1];
    m->n_tgt[1] = m->n[2] + m->n[3] + m->n[0];
    if (target_slot >= 0) {                          // kdtree{Corner,Surf}FromSubMap->setInputCloud (:1517-1518)
        int rc = lisreg_set_target_slot(c, target_slot, m->tgt[0].p, m->n_tgt[0], m->tgt[1].p, m->n_tgt[1], 16, LISREG_FMT_DEVICE);
        if (rc) return rc;
    }
    fill_info(*m, info);
    if (info) for (int d = 0; d < 6; ++d) info->crop[d] = isect[d];
    return LISREG_OK;
}

int lisreg_localmap_get(lisreg_ctx* c, int map_id, int cls, void* out, int capacity, int* n_out)
{
    if (!c) return LISREG_ERR_ARG;
    LocalMap* m = get_map(c, map_id, false);
    if (!m || !m->valid) return ctx_fail(c, LISREG_ERR_NO_TARGET, "localmap_get: no such local map");
    if (cls < 0 || cls > 6 || !n_out) return bad(c, "localmap_get: bad class / NULL count");
    const void* src = cls < 5 ? m->cls[cls].p : m->tgt[cls - 5].p;
    const int n = cls < 5 ? m->n[cls] : m->n_tgt[cls - 5];
    *n_out = n;
    if (n > capacity) return bad(c, "localmap_get: capacity too small (see *n_out)");
    if (n > 0 && !out) return bad(c, "localmap_get: NULL out");
    HIPCHK(c, hipSetDevice(c->device));
    if (n > 0) HIPCHK(c, hipMemcpy(out, src, sizeof(float4) * (size_t)n, hipMemcpyDefault));      // host or device destination
    return LISREG_OK;
}

// updateInitialGuess without IMU / odometry input (odomEstimationNode.cpp:351-392, subMapOptmizationNode.cpp:984-1020): the last
// frame-to-frame increment applied once more, T_guess = T_cur * (T_last^-1 * T_cur), in Eigen's float arithmetic
// (pcl::getTransformation, Affine3f::inverse, getTranslationAndEulerAngles).
// ---- the key-frame target of the odometry node (odomEstimationNode.cpp, USING_MULTI_FRAME_TARGET) ---------------------------
// saveKeyFrames (:421-468): the frame's full corner / surf feature clouds, transformPointCloud'ed into the map frame, appended;
// fewer than 20 are kept.  laserCloudInfoHandler (:185-207): target = the kept frames concatenated NEWEST FIRST, voxel grids
// mappingCornerLeafSize / mappingSurfLeafSize, then both kd-trees (scan2SubMapOptimization :602-603).  Here the frames stay in
// HBM as 16-byte records; per call the host sees counts only.
int lisreg_keyframes_reset(lisreg_ctx* c, int ring_id)
{
    if (!c) return LISREG_ERR_ARG;
    if (ring_id < 0 || ring_id > 1023) return bad(c, "keyframes_reset: bad ring id");
    if ((size_t)ring_id >= c->keyrings.size()) c->keyrings.resize((size_t)ring_id + 1);
    KeyframeRing& r = c->keyrings[(size_t)ring_id];
    HIPCHK(c, hipStreamSynchronize(c->stream));
    for (auto& f : r.frames) { f.cloud[0].release(); f.cloud[1].release(); }
    r.frames.clear();
    r.built = false;
    r.n_tgt[0] = r.n_tgt[1] = 0;
    r.valid = true;
    return LISREG_OK;
}

int lisreg_keyframes_push(lisreg_ctx* c, int ring_id, const void* corner, int n_corner, const void* surf, int n_surf, int stride, int fmt,
                          const float pose[6], int max_keep, lisreg_keyframes_info* info)
{
    if (!c) return LISREG_ERR_ARG;
    if (!pose || n_corner < 0 || n_surf < 0 || (n_corner > 0 && !corner) || (n_surf > 0 && !surf) || max_keep < 1) return bad(c, "keyframes_push: bad argument");
    const bool devfmt = fmt == LISREG_FMT_DEVICE || fmt == LISREG_FMT_DEVICE_XYZI;
    if (!devfmt && fmt != LISREG_FMT_XYZIL && fmt != LISREG_FMT_XYZI) return bad(c, "keyframes_push: unknown fmt");
    if (!devfmt && stride < 12) return bad(c, "keyframes_push: bad stride");
    if (ring_id < 0 || (size_t)ring_id >= c->keyrings.size() || !c->keyrings[(size_t)ring_id].valid) {
        int rc = lisreg_keyframes_reset(c, ring_id);
        if (rc) return rc;
    }
    KeyframeRing& r = c->keyrings[(size_t)ring_id];
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t st = c->stream;
    KeyframeRing::Frame f;
    const void* src_h[2] = { corner, surf };
    const int n[2] = { n_corner, n_surf };
    for (int k = 0; k < 2; ++k) {
        f.n[k] = n[k];
        HIPCHK(c, f.cloud[k].ensure(sizeof(float4) * (size_t)std::max(n[k], 1)));
        if (n[k] == 0) continue;
        const float4* src = nullptr;
        if (devfmt) src = static_cast<const float4*>(src_h[k]);
        else {
            const size_t bytes = (size_t)n[k] * (size_t)stride;
            HIPCHK(c, c->raw_upload.ensure(bytes + 32));
            HIPCHK(c, c->lm_in.ensure(sizeof(float4) * (size_t)n[k]));
            HIPCHK(c, hipMemcpyAsync(c->raw_upload.p, src_h[k], bytes, hipMemcpyHostToDevice, st));
            launch_pack_cloud(c->raw_upload.p, (size_t)n[k], stride, fmt == LISREG_FMT_XYZIL, c->lm_in.as<float4>(), st);
            HIPCHK(c, hipStreamSynchronize(st));     // pageable sources are free to change after the call
            src = c->lm_in.as<float4>();
        }
        int rc = lisreg_transform_cloud(c, src, n[k], 16, LISREG_FMT_DEVICE, pose, f.cloud[k].p);      // transformPointCloud(.., &thisPose6D)
        if (rc) { f.cloud[0].release(); f.cloud[1].release(); return rc; }
    }
    r.frames.push_back(f);
    r.built = false;
    r.payload_is_label = fmt == LISREG_FMT_DEVICE || fmt == LISREG_FMT_XYZIL;       // what the ring's voxel grids do with the fourth channel
    while ((int)r.frames.size() > max_keep) {         // while (size() >= 20) erase(begin())  with max_keep = 19
        HIPCHK(c, hipStreamSynchronize(st));
        r.frames.front().cloud[0].release(); r.frames.front().cloud[1].release();
        r.frames.erase(r.frames.begin());
    }
    if (info) { info->n_keyframes = (int)r.frames.size(); info->n_target_corner = r.n_tgt[0]; info->n_target_surf = r.n_tgt[1]; }
    return LISREG_OK;
}

int lisreg_keyframes_target(lisreg_ctx* c, int ring_id, float corner_leaf, float surf_leaf, int target_slot, lisreg_keyframes_info* info)
{
    if (!c) return LISREG_ERR_ARG;
    if (ring_id < 0 || (size_t)ring_id >= c->keyrings.size() || !c->keyrings[(size_t)ring_id].valid)
        return ctx_fail(c, LISREG_ERR_NO_TARGET, "keyframes_target: no such key-frame ring");
    KeyframeRing& r = c->keyrings[(size_t)ring_id];
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t st = c->stream;
    const float leaf[2] = { corner_leaf, surf_leaf };
    if (r.built && r.built_leaf[0] == corner_leaf && r.built_leaf[1] == surf_leaf && r.built_slot == target_slot &&
        (target_slot < 0 || ((size_t)target_slot < c->targets.size() && c->targets[(size_t)target_slot].valid &&
                             c->targets[(size_t)target_slot].gen == r.built_gen))) {
        // no key frame since the last call: the concatenation, both grids and both indexes would come out the same
        if (info) { info->n_keyframes = (int)r.frames.size(); info->n_target_corner = r.n_tgt[0]; info->n_target_surf = r.n_tgt[1]; }
        return LISREG_OK;
    }
    r.built = false;
    size_t total[2] = { 0, 0 };
    for (int k = 0; k < 2; ++k) {
        for (const auto& f : r.frames) total[k] += (size_t)f.n[k];
        HIPCHK(c, r.cat[k].ensure(sizeof(float4) * std::max<size_t>(total[k], 1)));
        HIPCHK(c, r.tgt[k].ensure(sizeof(float4) * std::max<size_t>(total[k], 1)));
        size_t off = 0;
        for (size_t i = r.frames.size(); i-- > 0;) {                 // newest first (:191-194)
            const auto& f = r.frames[i];
            if (f.n[k] > 0) HIPCHK(c, hipMemcpyAsync(r.cat[k].as<float4>() + off, f.cloud[k].p, sizeof(float4) * (size_t)f.n[k], hipMemcpyDeviceToDevice, st));
            off += (size_t)f.n[k];
        }
    }
    const int vfmt = r.payload_is_label ? LISREG_FMT_DEVICE : LISREG_FMT_DEVICE_XYZI;
    int nv[2] = { 0, 0 };
    if (total[0] > 0 && total[1] > 0) {                              // both grids as one sort (lisreg_voxel_downsample_multi)
        const void* vin[2] = { r.cat[0].p, r.cat[1].p }; void* vout[2] = { r.tgt[0].p, r.tgt[1].p };
        const int vn[2] = { (int)total[0], (int)total[1] };
        int rc = lisreg_voxel_downsample_multi(c, 2, vin, vn, leaf, vfmt, vout, vn, nv);
        if (rc) return rc;
    }
    for (int k = 0; k < 2; ++k) {
        if (total[k] > 0 && nv[k] == 0) {                            // alone, or a grid the joint call could not take
            int rc = lisreg_voxel_downsample(c, r.cat[k].p, (int)total[k], 16, vfmt, leaf[k], r.tgt[k].p, (int)total[k], &nv[k]);
            if (rc == LISREG_LEAF_TOO_SMALL) {               // PCL warns and hands the input through
                HIPCHK(c, hipMemcpyAsync(r.tgt[k].p, r.cat[k].p, sizeof(float4) * total[k], hipMemcpyDeviceToDevice, st));
                nv[k] = (int)total[k];
            } else if (rc) return rc;
        }
        r.n_tgt[k] = nv[k];
    }
    HIPCHK(c, hipStreamSynchronize(st));
    if (target_slot >= 0) {
        int rc = lisreg_set_target_slot(c, target_slot, r.tgt[0].p, r.n_tgt[0], r.tgt[1].p, r.n_tgt[1], 16, LISREG_FMT_DEVICE);
        if (rc) return rc;
        r.built_gen = c->targets[(size_t)target_slot].gen;
    }
    r.built = true; r.built_leaf[0] = corner_leaf; r.built_leaf[1] = surf_leaf; r.built_slot = target_slot;
    if (info) { info->n_keyframes = (int)r.frames.size(); info->n_target_corner = r.n_tgt[0]; info->n_target_surf = r.n_tgt[1]; }
    return LISREG_OK;
}

// Eigen::Affine3f arithmetic of updateInitialGuess, step by step in float: row-major 3 x 4 [R|t]
namespace {
// Affine3f::inverse(): linear part by the cofactor 3x3 inverse, t' = -L^-1 t
void aff_inverse(const float A[12], float I[12])
{
    const float a = A[0], b = A[1], cc = A[2], d = A[4], e = A[5], f = A[6], g = A[8], h = A[9], i = A[10];
    const float c00 = e * i - f * h, c01 = f * g - d * i, c02 = d * h - e * g;
    const float det = a * c00 + b * c01 + cc * c02, id = 1.f / det;
    const float L[9] = { c00 * id, (cc * h - b * i) * id, (b * f - cc * e) * id,
                         c01 * id, (a * i - cc * g) * id, (cc * d - a * f) * id,
                         c02 * id, (b * g - a * h) * id, (a * e - b * d) * id };
    for (int r = 0; r < 3; ++r) {
        for (int q = 0; q < 3; ++q) I[4 * r + q] = L[3 * r + q];
        I[4 * r + 3] = -(L[3 * r] * A[3] + L[3 * r + 1] * A[7] + L[3 * r + 2] * A[11]);
    }
}
// Z = X * Y (affine product, accumulated left to right)
void aff_mul(const float X[12], const float Y[12], float Z[12])
{
    for (int r = 0; r < 3; ++r) {
        for (int q = 0; q < 3; ++q) Z[4 * r + q] = X[4 * r] * Y[q] + X[4 * r + 1] * Y[4 + q] + X[4 * r + 2] * Y[8 + q];
        Z[4 * r + 3] = X[4 * r] * Y[3] + X[4 * r + 1] * Y[7] + X[4 * r + 2] * Y[11] + X[4 * r + 3];
    }
}
// pcl::getTranslationAndEulerAngles: x,y,z = t; roll = atan2(m21, m22); pitch = asin(-m20); yaw = atan2(m10, m00)
void aff_to_pose(const float F[12], float T[6])
{
    T[3] = F[3]; T[4] = F[7]; T[5] = F[11];
    T[0] = atan2f(F[9], F[10]);
    T[1] = asinf(-F[8]);
    T[2] = atan2f(F[4], F[0]);
}
// T <- T * (from^-1 * to)
void apply_increment(const float from[12], const float to[12], float T[6])
{
    float inv[12], inc[12], cur[12], fin[12];
    aff_inverse(from, inv);
    aff_mul(inv, to, inc);
    lisreg_pose_to_matrix(T, cur);
    aff_mul(cur, inc, fin);
    aff_to_pose(fin, T);
}
}  // namespace

void lisreg_predict_pose(const float T_last[6], const float T_cur[6], float T_guess[6])
{
    float A[12], B[12], T[6];
    lisreg_pose_to_matrix(T_last, A);
    lisreg_pose_to_matrix(T_cur, B);
    for (int k = 0; k < 6; ++k) T[k] = T_cur[k];
    apply_increment(A, B, T);
    for (int k = 0; k < 6; ++k) T_guess[k] = T[k];
}

void lisreg_guess_state_init(lisreg_guess_state* st) { if (st) memset(st, 0, sizeof *st); }

void lisreg_update_initial_guess(int variant, int use_imu_heading_initialization, const lisreg_guess_input* in, lisreg_guess_state* st,
                                 float T[6], float T_prediction[6])
{
    if (!in || !st || !T) return;
    const float imu_pose[6] = { in->imu_roll_init, in->imu_pitch_init, in->imu_yaw_init, 0.f, 0.f, 0.f };
    float imu_now[12];
    lisreg_pose_to_matrix(imu_pose, imu_now);                       // pcl::getTransformation(0, 0, 0, roll, pitch, yaw)
    if (!st->first_trans_available) {                               // :305-318 | :902-923
        T[0] = in->imu_roll_init; T[1] = in->imu_pitch_init; T[2] = in->imu_yaw_init;
        if (!use_imu_heading_initialization) T[2] = 0.f;
        memcpy(st->last_imu_transformation, imu_now, sizeof imu_now);
        st->first_trans_available = 1;
        return;
    }
    auto predict = [&]() { if (T_prediction) for (int k = 0; k < 6; ++k) T_prediction[k] = T[k]; };
    auto constant_velocity = [&]() {                                // :351-392 | :986-1020
        if (!st->first) { for (int k = 0; k < 6; ++k) st->last_transform_tobe_mapped[k] = T[k]; st->first = 1; return; }
        float back[12], last[12];
        lisreg_pose_to_matrix(T, back);
        lisreg_pose_to_matrix(st->last_transform_tobe_mapped, last);
        for (int k = 0; k < 6; ++k) st->last_transform_tobe_mapped[k] = T[k];
        apply_increment(last, back, T);
    };
    auto imu_increment = [&]() {                                    // :394-415 | :962-982
        apply_increment(st->last_imu_transformation, imu_now, T);
        predict();
        memcpy(st->last_imu_transformation, imu_now, sizeof imu_now);
    };
    if (in->odom_available) {                                       // :322-347 | :928-959
        const float gp[6] = { in->initial_guess_roll, in->initial_guess_pitch, in->initial_guess_yaw,
                              in->initial_guess_x, in->initial_guess_y, in->initial_guess_z };
        float back[12];
        lisreg_pose_to_matrix(gp, back);
        const bool had = st->last_imu_pre_trans_available != 0;
        if (!had) {
            memcpy(st->last_imu_pre_transformation, back, sizeof back);
            st->last_imu_pre_trans_available = 1;
        } else {
            apply_increment(st->last_imu_pre_transformation, back, T);
            predict();
            memcpy(st->last_imu_pre_transformation, back, sizeof back);
        }
        if (variant == 0) {
            // copy #1 returns only from the increment branch (saving the IMU attitude on the way, :344); the very first odometry message
            // falls through to the tests below: `odomAvailable == false` fails, the IMU increment applies if there is an IMU
            if (had) { memcpy(st->last_imu_transformation, imu_now, sizeof imu_now); return; }
        } else {
            if (in->imu_available) memcpy(st->last_imu_transformation, imu_now, sizeof imu_now);      // :954-955
            return;
        }
    }
    if (variant == 0) {
        if (!in->odom_available) { constant_velocity(); return; }   // :351 tests odomAvailable alone
        if (in->imu_available) imu_increment();
    } else {
        if (in->imu_available) { imu_increment(); return; }
        constant_velocity();                                        // neither IMU nor odometry
    }
}

}  // extern "C"
