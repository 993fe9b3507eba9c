// lisreg_api_feed.hip — host feeder of the batch entry points: the caller's clouds (the reference's 32-byte PCL structs, common.h:9,25-35)
// reach HBM as 16-byte records without the padding ever crossing the host link.
//
// What the link and the host give (tests/probes/h2d_probe.hip, MI355X box of the pool, 64 scans of 64 x 1800 = 236 MB of structs):
//   the structs as they are, 128 hipMemcpyAsync (one per cloud)   5.4 ms   (43 GB/s — per-copy overhead)
//   the same bytes as ONE pinned copy                             4.1 ms   (57 GB/s)
//   packing to 16-byte records, 8 host threads                    1.7 ms   (1 thread: 12.3 ms)
//   the packed records, one copy                                  2.1 ms
// so a few feeder threads packing into a pinned staging buffer, with the chunks copied as they complete on a copy stream of their own,
// put a batch into HBM in ~2.2 ms instead of 5.4 — and under the previous batch's kernels when the caller stages batch k + 1 before it
// fetches batch k (lisreg_stage_host_items + lisreg_batch_prepare / _run / _fetch; bench.py `pcie_inclusive`).
#include "lisreg_ctx.hpp"

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstring>
#include <memory>
#include <mutex>
#include <thread>
#include <cctype>
#include <cstdio>
#include <pthread.h>
#include <sched.h>

namespace lisreg {

// The host side of the feeder is memory-bound (two passes over the batch), and the pool's hosts are multi-socket: the same binary staged a
// batch in 2.7 ms on one box and 5.9 ms on another.  The staging buffers come from hipHostMalloc, which places them on the
// NUMA node nearest the device; the packing threads are bound to the CPUs of that node that this process may run on (option
// "feeder_numa", default 1; nothing happens when sysfs says nothing or the process owns no CPU there).
static int feeder_numa_cpus(int device, std::vector<int>& cpus)
{
    cpus.clear();
    char bdf[64] = { 0 };
    if (hipDeviceGetPCIBusId(bdf, (int)sizeof bdf, device) != hipSuccess) return -1;
    for (char* p = bdf; *p; ++p) *p = (char)tolower(*p);
    char path[256];
    snprintf(path, sizeof path, "/sys/bus/pci/devices/%s/numa_node", bdf);
    int node = -1;
    if (FILE* f = fopen(path, "r")) { if (fscanf(f, "%d", &node) != 1) node = -1; fclose(f); }
    if (node < 0) return -1;
    snprintf(path, sizeof path, "/sys/devices/system/node/node%d/cpulist", node);
    std::vector<int> on_node;
    if (FILE* f = fopen(path, "r")) {
        int a = 0, b = 0; char sep = 0;
        while (fscanf(f, "%d", &a) == 1) {
            b = a;
            int ch = fgetc(f);
            if (ch == '-') { if (fscanf(f, "%d", &b) != 1) b = a; ch = fgetc(f); }
            for (int k = a; k <= b && k < CPU_SETSIZE; ++k) on_node.push_back(k);
            if (ch != ',') break;
            (void)sep;
        }
        fclose(f);
    }
    cpu_set_t mine; CPU_ZERO(&mine);
    if (sched_getaffinity(0, sizeof mine, &mine) != 0) return node;
    for (int k : on_node) if (CPU_ISSET(k, &mine)) cpus.push_back(k);
    return node;
}

struct PackPool {
    // one packing job: the workers take chunk indices from `next` until they run out; a job object lives as long as anyone holds it, so a
    // worker that wakes late (or is still leaving the previous job) never touches the tables of the next one
    struct Job {
        const PackChunk* chunks = nullptr;
        std::atomic<int>* done = nullptr;      // done[i] = 1 once chunk i is in the staging buffer
        int n_chunks = 0;
        // chunks are handed out from both ends: the packing threads take the lowest index still free, the publishing thread may take the
        // highest one for the copy engine (take_back); `span` = (front << 32) | back, free chunks are front .. back - 1
        std::atomic<unsigned long long> span{ 0 };
        std::atomic<int> active{ 0 };          // threads inside drain() on this job (taken under the pool mutex together with the job)
        void cancel() { span.store(0, std::memory_order_relaxed); }       // nothing is handed out any more
        int take_front()
        {
            unsigned long long v = span.load(std::memory_order_relaxed);
            for (;;) {
                const unsigned f = (unsigned)(v >> 32), b = (unsigned)v;
                if (f >= b) return -1;
                if (span.compare_exchange_weak(v, ((unsigned long long)(f + 1) << 32) | b, std::memory_order_relaxed)) return (int)f;
            }
        }
        int take_back()
        {
            unsigned long long v = span.load(std::memory_order_relaxed);
            for (;;) {
                const unsigned f = (unsigned)(v >> 32), b = (unsigned)v;
                if (f >= b) return -1;
                if (span.compare_exchange_weak(v, ((unsigned long long)f << 32) | (b - 1), std::memory_order_relaxed)) return (int)(b - 1);
            }
        }
    };
    std::vector<std::thread> th;
    std::mutex mu;
    std::condition_variable cv;
    std::shared_ptr<Job> job;
    unsigned long long gen = 0;
    bool quit = false;

    static void pack(const PackChunk& k)
    {
        const unsigned char* s = k.src;
        lisreg_dpoint* o = k.dst;
        const bool lab = k.fmt == LISREG_FMT_XYZIL;
        for (int i = 0; i < k.n; ++i, s += k.stride, ++o) {
            memcpy(o, s, 12);
            uint16_t l = 0;
            if (lab) memcpy(&l, s + 20, 2);
            o->payload = l;
        }
    }
    static void drain(Job& j)
    {
        for (;;) {
            const int i = j.take_front();
            if (i < 0) return;
            pack(j.chunks[i]);
            j.done[i].store(1, std::memory_order_release);
        }
    }
    void drain()                                                  // the publishing thread helps
    {
        std::shared_ptr<Job> j;
        { std::lock_guard<std::mutex> lk(mu); j = job; if (j) j->active.fetch_add(1, std::memory_order_relaxed); }
        if (j) { drain(*j); j->active.fetch_sub(1, std::memory_order_release); }
    }
    // Ends a job whatever state it is in: no chunk is handed out any more, late wakers find no job, and the call returns only when no
    // thread is inside it — after that the chunk table, the done flags, the staging buffer and the caller's clouds may go away.  Every
    // exit of the publishing functions goes through here (JobGuard), the error returns included.
    void finish(const std::shared_ptr<Job>& j)
    {
        if (!j) return;
        j->cancel();
        { std::lock_guard<std::mutex> lk(mu); if (job == j) job.reset(); }
        while (j->active.load(std::memory_order_acquire) != 0) std::this_thread::yield();
    }
    void worker()
    {
        unsigned long long seen = 0;
        for (;;) {
            std::shared_ptr<Job> j;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return quit || gen != seen; });
                if (quit) return;
                seen = gen;
                j = job;
                if (j) j->active.fetch_add(1, std::memory_order_relaxed);
            }
            if (j) { drain(*j); j->active.fetch_sub(1, std::memory_order_release); }
        }
    }
    // cpus the packing threads are bound to (the GPU's NUMA node, lisreg_feeder_numa below); empty = wherever the scheduler puts them
    std::vector<int> cpus;
    void start(int n_threads)
    {
        for (int t = (int)th.size(); t < n_threads; ++t) {
            th.emplace_back([this] { worker(); });
            if (!cpus.empty()) {
                cpu_set_t set; CPU_ZERO(&set);
                for (int cpu : cpus) if (cpu >= 0 && cpu < CPU_SETSIZE) CPU_SET(cpu, &set);
                (void)pthread_setaffinity_np(th.back().native_handle(), sizeof set, &set);
            }
        }
    }
    // publish a job; the caller then waits on done[] in order (and may call drain() itself).  The chunk table and the done flags must stay
    // valid until every done flag is set — after that no worker reads them again (an index past n_chunks ends its loop)
    std::shared_ptr<Job> run(const PackChunk* c, int n, std::atomic<int>* d)
    {
        auto j = std::make_shared<Job>();
        j->chunks = c; j->n_chunks = n; j->done = d;
        j->span.store((unsigned long long)(unsigned)n, std::memory_order_relaxed);
        {
            std::lock_guard<std::mutex> lk(mu);
            job = j;
            ++gen;
        }
        cv.notify_all();
        return j;
    }
    struct JobGuard {                          // scope of one published job
        PackPool* pool; std::shared_ptr<Job> job;
        ~JobGuard() { if (pool) pool->finish(job); }
    };
    ~PackPool()
    {
        { std::lock_guard<std::mutex> lk(mu); quit = true; }
        cv.notify_all();
        for (auto& t : th) t.join();
    }
};

void feeder_destroy(lisreg_ctx* c)
{
    // order: the copy stream drains first (its copies read the pinned staging buffers and write the device buffers), then the threads
    // are joined (nothing packs into a buffer that is about to go), then events, pinned and device memory
    if (c->copy_stream) (void)hipStreamSynchronize(c->copy_stream);
    if (c->pack_stream) (void)hipStreamSynchronize(c->pack_stream);
    delete c->pack_pool; c->pack_pool = nullptr;
    for (int b = 0; b < 2; ++b) {
        if (c->pack_copied[b]) (void)hipEventDestroy(c->pack_copied[b]);
        if (c->pack_free[b]) (void)hipEventDestroy(c->pack_free[b]);
        c->pack_copied[b] = c->pack_free[b] = nullptr;
        if (c->pack_host[b]) (void)hipHostFree(c->pack_host[b]);
        c->pack_host[b] = nullptr; c->pack_cap[b] = 0;
        c->pack_dev[b].release(); c->pack_raw[b].release();
    }
    if (c->pack_raw_done) (void)hipEventDestroy(c->pack_raw_done);
    c->pack_raw_done = nullptr; c->pack_pending = nullptr;
    if (c->up_host) (void)hipHostFree(c->up_host);
    c->up_host = nullptr; c->up_cap = 0;
    if (c->pack_kernels_done) (void)hipEventDestroy(c->pack_kernels_done);
    c->pack_kernels_done = nullptr;
    if (c->pack_stream) { (void)hipStreamDestroy(c->pack_stream); c->pack_stream = nullptr; }
    if (c->copy_stream) { (void)hipStreamDestroy(c->copy_stream); c->copy_stream = nullptr; }
}

}  // namespace lisreg

using namespace lisreg;

// Pack the host clouds of `items` to 16-byte records with the feeder threads and upload them, asynchronously, into one of two device
// buffers owned by the context.  items_out[i] = items[i] with device pointers / LISREG_FMT_DEVICE (items that are device records already
// pass through).  The uploads run on a copy stream of the context; the next lisreg_batch_prepare waits for them on the device, the host
// does not.  The caller's clouds are not referenced after the call returns.  Buffers alternate between calls, so batch k + 1 can be
// staged while batch k (staged by the previous call) is still running.
int lisreg_stage_host_items(lisreg_ctx* c, int n_items, const lisreg_item* items, lisreg_item* items_out)
{
    if (!c) return LISREG_ERR_ARG;
    if (n_items < 0 || (n_items > 0 && (!items || !items_out))) return ctx_fail(c, LISREG_ERR_ARG, "stage_host_items: bad arguments");
    size_t total = 0;
    for (int i = 0; i < n_items; ++i) {
        const lisreg_item& in = items[i];
        if (in.n_corner < 0 || in.n_surf < 0) return ctx_fail(c, LISREG_ERR_ARG, "stage_host_items: negative count");
        if (in.fmt == LISREG_FMT_DEVICE) continue;
        if (in.stride_bytes < 12 || (in.fmt == LISREG_FMT_XYZIL && in.stride_bytes < 22)) return ctx_fail(c, LISREG_ERR_ARG, "stage_host_items: bad stride");
        if ((in.n_corner > 0 && !in.src_corner) || (in.n_surf > 0 && !in.src_surf)) return ctx_fail(c, LISREG_ERR_ARG, "stage_host_items: NULL cloud");
        total += (size_t)in.n_corner + (size_t)in.n_surf;
    }
    HIPCHK(c, hipSetDevice(c->device));
    const int b = c->pack_flip;
    c->pack_flip ^= 1;
    // The copy stream carries asynchronous copies and ONE event record per call — never a device-side wait, never a kernel.  HIP maps a
    // process's streams onto a handful of hardware queues (four by default), so with a few contexts alive the copy stream shares an AQL
    // queue with some compute stream, possibly this context's own: a wait or kernel packet of the copy stream then queues up BEHIND the
    // kernels of the batch that is running, the copies ordered after it start when that batch is through, and the upload of batch k + 1
    // no longer hides underneath batch k (measured, round 6: 12.8 k reg/s against 23.7 k for the same loop with the streams on separate
    // queues — profiles/r06_pcie_overlap.md).  The copy engines themselves do not go through that queue.
    static const bool feed_legacy = getenv("LISREG_FEED_LEGACY") != nullptr;      // A/B only (tests/pcie_prio_ab.sh): rounds 3-5's device-side wait and
                                                                                  // packing kernels on the copy stream; same results, no overlap when queues alias
    if (!c->copy_stream) {
        const char* e = getenv("LISREG_COPY_PRIO");              // experiments: -1 (high) / 1 (low) put the stream into another queue pool
        if (e) HIPCHK(c, hipStreamCreateWithPriority(&c->copy_stream, hipStreamNonBlocking, atoi(e)));
        else HIPCHK(c, hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
    }
    if (!c->pack_copied[b]) HIPCHK(c, hipEventCreateWithFlags(&c->pack_copied[b], hipEventDisableTiming));
    if (!c->pack_free[b]) HIPCHK(c, hipEventCreateWithFlags(&c->pack_free[b], hipEventDisableTiming));
    else if (!feed_legacy) HIPCHK(c, hipEventSynchronize(c->pack_free[b]));      // the batch that last read device buffer b has run (two batches
                                                                                 // back in a pipelined loop: long done; a HOST wait, for the reason above)
    else HIPCHK(c, hipStreamWaitEvent(c->copy_stream, c->pack_free[b], 0));
    // the staging buffer's previous contents have left it (its copies are two calls old)
    if (c->pack_cap[b]) HIPCHK(c, hipEventSynchronize(c->pack_copied[b]));
    const size_t bytes = sizeof(lisreg_dpoint) * std::max<size_t>(total, 1);
    if (bytes > c->pack_cap[b]) {
        if (c->pack_host[b]) (void)hipHostFree(c->pack_host[b]);
        c->pack_host[b] = nullptr; c->pack_cap[b] = 0;
        HIPCHK(c, hipHostMalloc((void**)&c->pack_host[b], bytes + bytes / 8 + 4096, hipHostMallocDefault));
        c->pack_cap[b] = bytes + bytes / 8 + 4096;
    }
    HIPCHK(c, c->pack_dev[b].ensure(bytes));
    lisreg_dpoint* host = reinterpret_cast<lisreg_dpoint*>(c->pack_host[b]);
    lisreg_dpoint* dev = c->pack_dev[b].as<lisreg_dpoint>();
    // chunks of <= 64 k points, in staging order
    constexpr int kChunk = 65536;
    std::vector<PackChunk>& chunks = c->pack_chunks;
    chunks.clear();
    size_t off = 0;
    for (int i = 0; i < n_items; ++i) {
        items_out[i] = items[i];
        const lisreg_item& in = items[i];
        if (in.fmt == LISREG_FMT_DEVICE) continue;
        lisreg_item& d = items_out[i];
        const void* srcs[2] = { in.src_corner, in.src_surf };
        const int cnts[2] = { in.n_corner, in.n_surf };
        const void** dsts[2] = { &d.src_corner, &d.src_surf };
        for (int k = 0; k < 2; ++k) {
            *dsts[k] = dev + off;
            // a cloud in pinned host memory can also cross the link as it is (the copy engine reads it) and be packed on the device
            int pinned = 0;
            if (cnts[k] > 0) {
                hipPointerAttribute_t at;
                if (hipPointerGetAttributes(&at, srcs[k]) == hipSuccess && at.type == hipMemoryTypeHost) pinned = 1;
                else (void)hipGetLastError();
            }
            for (int s = 0; s < cnts[k]; s += kChunk)
                chunks.push_back(PackChunk{ static_cast<const unsigned char*>(srcs[k]) + (size_t)s * (size_t)in.stride_bytes, host + off + s,
                                            std::min(kChunk, cnts[k] - s), in.stride_bytes, in.fmt, pinned });
            off += (size_t)cnts[k];
        }
        d.fmt = LISREG_FMT_DEVICE; d.stride_bytes = (int)sizeof(lisreg_dpoint);
    }
    const int n_chunks = (int)chunks.size();
    c->pack_stolen = 0;
    if (n_chunks > 0) {
        // (no job is alive here: the previous call left through its JobGuard, so the flag array may be replaced)
        if ((int)c->pack_done.size() < n_chunks) c->pack_done = std::vector<std::atomic<int>>((size_t)n_chunks);
        for (int i = 0; i < n_chunks; ++i) c->pack_done[(size_t)i].store(0, std::memory_order_relaxed);
        // small batches are packed by the calling thread alone (a single odometry frame: waking threads costs more than it packs)
        const int want = total >= 262144 ? std::max(1, std::min(c->feeder_threads, (int)std::thread::hardware_concurrency() - 1)) : 0;
        std::shared_ptr<PackPool::Job> job;
        if (want > 0) {
            if (!c->pack_pool) { c->pack_pool = new PackPool(); if (c->feeder_numa) c->feeder_node = feeder_numa_cpus(c->device, c->pack_pool->cpus); c->feeder_cpus = (int)c->pack_pool->cpus.size(); }
            c->pack_pool->start(want);
            job = c->pack_pool->run(chunks.data(), n_chunks, c->pack_done.data());
        }
        // every way out of this block — the HIPCHK returns too — first takes the job away from the threads and waits for them
        PackPool::JobGuard guard{ want > 0 ? c->pack_pool : nullptr, job };
        // Copies follow the packing chunk by chunk, several chunks per copy (per-copy overhead is ~10 us; 4 MB copies run at link rate).
        // While the next packed chunk is not ready and the copy engine has nothing left to do, this thread hands the engine the LAST free
        // chunk as it is — 32-byte structs over the link, packed by a kernel on the copy stream — so the two ends of the batch are worked
        // on by the host threads and by the copy engine at once and the split follows their speeds (a contended host packs 2-3x slower
        // than a quiet one; the engine's rate does not change).
        constexpr int kPerCopy = 4;
        int next_copy = 0, lowest_stolen = n_chunks;
        size_t raw_off = 0;
        unsigned char* raw_dev = nullptr;
        bool force_one = job && c->feeder_engine >= 3;           // tests: one chunk goes to the engine whatever the packing threads' speed
        for (int i = 0; i < lowest_stolen; ++i) {
            if (want > 0) {
                while (force_one || !c->pack_done[(size_t)i].load(std::memory_order_acquire)) {
                    bool stole = false;
                    const bool forced = force_one;
                    force_one = false;
                    if (job && c->feeder_engine > 0 && chunks[(size_t)lowest_stolen - 1].pinned &&
                        (forced || c->feeder_engine > 1 || (hipStreamQuery(c->copy_stream) == hipSuccess && (!c->pack_stream || hipStreamQuery(c->pack_stream) == hipSuccess)))) {
                        const int k = job->take_back();
                        if (k >= 0) {
                            const PackChunk& ck = chunks[(size_t)k];
                            const size_t bytes_k = (size_t)ck.n * (size_t)ck.stride;
                            if (!raw_dev) {
                                size_t worst = 0;
                                for (int q = i; q < n_chunks; ++q) worst += (size_t)chunks[(size_t)q].n * (size_t)chunks[(size_t)q].stride + 64;
                                HIPCHK(c, c->pack_raw[b].ensure(worst));
                                raw_dev = static_cast<unsigned char*>(c->pack_raw[b].p);
                            }
                            // on a stream of their own: the packing kernel is a packet of a hardware queue that may be busy with the running
                            // batch, and the packed chunks' copies must not be ordered behind it
                            if (!c->pack_stream && !feed_legacy) HIPCHK(c, hipStreamCreateWithFlags(&c->pack_stream, hipStreamNonBlocking));
                            hipStream_t ps = feed_legacy ? c->copy_stream : c->pack_stream;
                            HIPCHK(c, hipMemcpyAsync(raw_dev + raw_off, ck.src, bytes_k, hipMemcpyHostToDevice, ps));
                            launch_pack_cloud(raw_dev + raw_off, (size_t)ck.n, ck.stride, ck.fmt == LISREG_FMT_XYZIL ? 1 : 0,
                                              reinterpret_cast<float4*>(dev + (ck.dst - host)), ps);
                            raw_off += (bytes_k + 63) & ~(size_t)63;
                            if (!c->pack_raw_done) HIPCHK(c, hipEventCreateWithFlags(&c->pack_raw_done, hipEventDisableTiming));
                            HIPCHK(c, hipEventRecord(c->pack_raw_done, ps));      // the engine has read the caller's memory up to here
                            lowest_stolen = k;
                            stole = true;
                            if (k <= i) break;                     // this very chunk went to the engine
                        }
                    } else (void)hipGetLastError();                // hipStreamQuery's "not ready" is not an error to keep
                    if (!stole) std::this_thread::yield();
                }
                if (i >= lowest_stolen) break;
            } else PackPool::pack(chunks[(size_t)i]);
            if (i + 1 - next_copy >= kPerCopy || i + 1 == lowest_stolen) {
                lisreg_dpoint* h0 = chunks[(size_t)next_copy].dst;
                lisreg_dpoint* h1 = chunks[(size_t)i].dst + chunks[(size_t)i].n;
                HIPCHK(c, hipMemcpyAsync(dev + (h0 - host), h0, sizeof(lisreg_dpoint) * (size_t)(h1 - h0), hipMemcpyHostToDevice, c->copy_stream));
                next_copy = i + 1;
            }
        }
        // packed chunks below the first stolen one that the loop left behind when it stopped
        if (next_copy < lowest_stolen) {
            for (int i = next_copy; i < lowest_stolen; ++i)
                while (want > 0 && !c->pack_done[(size_t)i].load(std::memory_order_acquire)) std::this_thread::yield();
            lisreg_dpoint* h0 = chunks[(size_t)next_copy].dst;
            lisreg_dpoint* h1 = chunks[(size_t)lowest_stolen - 1].dst + chunks[(size_t)lowest_stolen - 1].n;
            HIPCHK(c, hipMemcpyAsync(dev + (h0 - host), h0, sizeof(lisreg_dpoint) * (size_t)(h1 - h0), hipMemcpyHostToDevice, c->copy_stream));
        }
        HIPCHK(c, hipGetLastError());
        c->pack_stolen = n_chunks - lowest_stolen; c->pack_chunks_n = n_chunks;
        // "the caller's clouds are not referenced after the call returns": chunks the copy engine took are read from the caller's own
        // (pinned) memory by asynchronous copies — wait for the last of those; the packed chunks left the caller's memory on the host
        if (c->pack_stolen > 0 && c->pack_raw_done) HIPCHK(c, hipEventSynchronize(c->pack_raw_done));
    }
    if (c->pack_stolen > 0 && c->pack_stream) {   // the packing kernels' output belongs to the batch: the one wait the copy stream ever gets, at its very end
        if (!c->pack_kernels_done) HIPCHK(c, hipEventCreateWithFlags(&c->pack_kernels_done, hipEventDisableTiming));
        HIPCHK(c, hipEventRecord(c->pack_kernels_done, c->pack_stream));
        HIPCHK(c, hipStreamWaitEvent(c->copy_stream, c->pack_kernels_done, 0));
    }
    HIPCHK(c, hipEventRecord(c->pack_copied[b], c->copy_stream));
    c->pack_pending = c->pack_copied[b];          // the next batch_prepare makes the context's stream wait for it
    c->pack_last = b;
    return LISREG_OK;
}

// One host cloud into a caller-owned device buffer as 16-byte records: packed into a pinned staging buffer (by the feeder threads when
// the cloud is big enough to be worth waking them), one copy on the context's stream, and the call returns when the records are there.
// The per-frame upload of the sequential chains (a sweep of 64 x 1800 structs: 0.1 ms instead of 0.4 ms through a pageable copy of
// host-packed records).
int lisreg_upload_cloud(lisreg_ctx* c, const void* cloud, int n, int stride_bytes, int fmt, void* dev_out)
{
    if (!c) return LISREG_ERR_ARG;
    if (n < 0 || (n > 0 && (!cloud || !dev_out))) return ctx_fail(c, LISREG_ERR_ARG, "upload_cloud: bad arguments");
    if (fmt != LISREG_FMT_XYZIL && fmt != LISREG_FMT_XYZI) return ctx_fail(c, LISREG_ERR_ARG, "upload_cloud: host clouds only (LISREG_FMT_XYZI / _XYZIL)");
    if (stride_bytes < 12 || (fmt == LISREG_FMT_XYZIL && stride_bytes < 22)) return ctx_fail(c, LISREG_ERR_ARG, "upload_cloud: bad stride");
    if (n == 0) return LISREG_OK;
    HIPCHK(c, hipSetDevice(c->device));
    const size_t bytes = sizeof(lisreg_dpoint) * (size_t)n;
    if (bytes > c->up_cap) {
        HIPCHK(c, hipStreamSynchronize(c->stream));
        const size_t old_cap = c->up_cap;
        if (c->up_host) (void)hipHostFree(c->up_host);
        c->up_host = nullptr; c->up_cap = 0;
        const size_t want = std::max(std::max(bytes + bytes / 4 + 4096, 2 * old_cap), (size_t)8 << 20);     // sweeps vary by a few per cent: no re-pinning per frame
        HIPCHK(c, hipHostMalloc((void**)&c->up_host, want, hipHostMallocDefault));
        c->up_cap = want;
    }
    lisreg_dpoint* host = reinterpret_cast<lisreg_dpoint*>(c->up_host);
    constexpr int kChunk = 16384;
    std::vector<PackChunk> chunks;
    for (int s = 0; s < n; s += kChunk)
        chunks.push_back(PackChunk{ static_cast<const unsigned char*>(cloud) + (size_t)s * (size_t)stride_bytes, host + s, std::min(kChunk, n - s), stride_bytes, fmt, 0 });
    const int n_chunks = (int)chunks.size();
    const int want = n >= 4 * kChunk ? std::max(0, std::min(std::min(c->feeder_threads, n_chunks - 1), (int)std::thread::hardware_concurrency() - 1)) : 0;
    if (want > 0) {
        if ((int)c->pack_done.size() < n_chunks) c->pack_done = std::vector<std::atomic<int>>((size_t)n_chunks);
        for (int i = 0; i < n_chunks; ++i) c->pack_done[(size_t)i].store(0, std::memory_order_relaxed);
        if (!c->pack_pool) { c->pack_pool = new PackPool(); if (c->feeder_numa) c->feeder_node = feeder_numa_cpus(c->device, c->pack_pool->cpus); c->feeder_cpus = (int)c->pack_pool->cpus.size(); }
        c->pack_pool->start(want);
        // `chunks` is a local: the guard ends the job (and waits for the threads) before the table goes out of scope, on every path
        PackPool::JobGuard guard{ c->pack_pool, c->pack_pool->run(chunks.data(), n_chunks, c->pack_done.data()) };
        c->pack_pool->drain();                                   // the caller packs too
        for (int i = 0; i < n_chunks; ++i) while (!c->pack_done[(size_t)i].load(std::memory_order_acquire)) std::this_thread::yield();
    } else {
        for (const auto& k : chunks) PackPool::pack(k);
    }
    HIPCHK(c, hipMemcpyAsync(dev_out, host, bytes, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return LISREG_OK;
}
