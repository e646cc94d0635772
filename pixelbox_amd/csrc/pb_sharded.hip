// pb_sharded.hip -- the row-sharded `semantic_hashes` index behind the C ABI: ONE host process drives N GPUs.
//
// The reference is a single Rust process (engine.rs:79-145 owns everything); it cannot host one rank per GPU.  So the
// multi-GPU form of the scan lives here, under `extern "C"`: the table is split by rows over N device-resident shards
// (pb_index each), a query batch runs on every shard concurrently (one host worker thread per shard -- the searches
// block on their streams), every shard leaves its packed top-k message int64[nq][2k+1] in its own HBM, ONE
// `ncclAllGather` per batch exchanges the messages over xGMI (RCCL, communicators from `ncclCommInitAll` -- the
// single-process form), and a device kernel on shard 0 merges the N sorted lists per query in the reference's
// (dist, image_id) order.  Nothing else crosses between GPUs; no collective exists on any other path.
//
// RCCL is dlopen'ed here, at pb_sharded_create, and only when the shards sit on more than one distinct device: a
// single-GPU host never loads it.  Shards that share a device (the 1-GPU test topology: device_ids = {0, 0, 0})
// exchange their messages with device-to-device copies instead -- RCCL refuses duplicate devices in one communicator.
//
// Reference: engine.rs:363-396 (query), :228-259 (insert), :117-145 (open); SURVEY.md section 8e.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <condition_variable>
#include <functional>
#include <limits>
#include <mutex>
#include <new>
#include <thread>
#include <unordered_set>
#include <vector>

#include "pb_common.h"
#include "pb_merge_kernels.h"

namespace {

// ---- RCCL through dlopen: the five entry points the exchange needs ----
struct Rccl {
    void *lib = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
};

int load_rccl(Rccl *r) {
    // a host process that already carries an RCCL (e.g. the Python test harness with torch loaded) must not get a
    // second copy: RTLD_NOLOAD first, then the ROCm install
    for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
        r->lib = dlopen(name, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL);
        if (r->lib) break;
    }
    if (!r->lib)
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            r->lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (r->lib) break;
        }
    if (!r->lib) return pb::fail(PB_ERR_HIP, "pb_sharded: cannot load librccl.so.1 (%s) -- shards on more than one GPU need RCCL", dlerror());
#define PB_SYM(field, name)                                                                                   \
    r->field = reinterpret_cast<decltype(r->field)>(dlsym(r->lib, name));                                     \
    if (!r->field) return pb::fail(PB_ERR_HIP, "pb_sharded: librccl has no symbol %s", name)
    PB_SYM(CommInitAll, "ncclCommInitAll");
    PB_SYM(CommDestroy, "ncclCommDestroy");
    PB_SYM(AllGather, "ncclAllGather");
    PB_SYM(GroupStart, "ncclGroupStart");
    PB_SYM(GroupEnd, "ncclGroupEnd");
    PB_SYM(GetErrorString, "ncclGetErrorString");
#undef PB_SYM
    return PB_OK;
}

#define PB_NCCL(s, expr)                                                                                         \
    do {                                                                                                         \
        ncclResult_t _r = (expr);                                                                                \
        if (_r != ncclSuccess)                                                                                   \
            return pb::fail(PB_ERR_HIP, "%s:%d %s -> %s", __FILE__, __LINE__, #expr, (s)->rccl.GetErrorString(_r)); \
    } while (0)

// one worker thread per shard: runs the closures the calling thread hands it, one at a time
struct Worker {
    std::thread th;
    std::mutex mu;
    std::condition_variable cv;
    std::function<int()> job;
    bool has_job = false, done = false, quit = false;
    int rc = 0;
    char err[512] = {0};

    void loop() {
        for (;;) {
            std::function<int()> j;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return has_job || quit; });
                if (quit) return;
                j = std::move(job);
                has_job = false;
            }
            int r;
            try {  // a throw out of a std::thread terminates the host process: bad_alloc and friends become status codes
                r = j();
            } catch (const std::bad_alloc &) {
                r = pb::fail(PB_ERR_NOMEM, "shard worker: out of host memory");
            } catch (const std::exception &ex) {
                r = pb::fail(PB_ERR_INTERNAL, "shard worker: %s", ex.what());
            }
            {
                std::lock_guard<std::mutex> lk(mu);
                rc = r;
                if (r) snprintf(err, sizeof(err), "%s", pb::tls_error());  // the message lives in THIS thread's buffer
                done = true;
            }
            cv.notify_all();
        }
    }
    void start() { th = std::thread([this] { loop(); }); }
    void submit(std::function<int()> j) {
        {
            std::lock_guard<std::mutex> lk(mu);
            job = std::move(j);
            has_job = true;
            done = false;
        }
        cv.notify_all();
    }
    int wait() {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return done; });
        return rc;
    }
    void stop() {
        {
            std::lock_guard<std::mutex> lk(mu);
            quit = true;
        }
        cv.notify_all();
        if (th.joinable()) th.join();
    }
};

}  // namespace

struct pb_sharded {
    uint32_t dim = 0;
    uint64_t capacity = 0;
    int n = 0;
    std::vector<int> devices;
    std::vector<pb_index *> shards;
    std::vector<uint64_t> shard_cap;
    std::vector<hipStream_t> streams;       // exchange / merge streams, one per shard (on that shard's device)
    std::vector<int64_t *> d_packed;        // [nq_cap][2 k_cap + 1] per shard: this shard's message
    std::vector<int64_t *> d_gathered;      // [n][nq_cap][2 k_cap + 1] per shard: everyone's messages
    int64_t *d_out_ids = nullptr;           // merge output on shard 0's device
    float *d_out_dist = nullptr;
    uint32_t *d_out_count = nullptr;
    int64_t *h_out_ids = nullptr;           // pinned
    float *h_out_dist = nullptr;
    uint32_t *h_out_count = nullptr;
    uint32_t nq_cap = 0, k_cap = 0;
    bool use_rccl = false;
    Rccl rccl;
    std::vector<ncclComm_t> comms;
    std::vector<Worker *> workers;
    uint64_t n_exchanges = 0;               // all-gathers (or copy exchanges) issued
    // device-side appends (pb_sharded_append_device) copy outside `mu`, so that the embed threads of different GPUs insert
    // concurrently: what they are about to store is RESERVED under `mu` first -- rows per shard (the capacity checks count them)
    // and the ids themselves (the uniqueness checks of concurrent calls see them) -- and released when the copy has returned
    std::vector<uint64_t> pending;
    std::vector<uint64_t> base;     // rows of a shard counted as stored while it has appends in flight (pending > 0): its size when the
                                    // first of them was admitted + what has been committed since.  The shard's LIVE size cannot be used
                                    // then: between a device-side append's completion and the release of its reservation its rows are
                                    // in the live size AND in `pending`, and a concurrent call saw a full table that was not (one flaky
                                    // "index full: capacity 80" in ~20 runs of the tight-capacity ingest test)
    std::unordered_set<int64_t> inflight;
    std::mutex mu;
};

namespace {

void free_buffers(pb_sharded *s) {
    for (int g = 0; g < s->n; ++g) {
        pb::DeviceGuard guard(s->devices[g]);
        if (g < (int)s->d_packed.size()) (void)hipFree(s->d_packed[g]);
        if (g < (int)s->d_gathered.size()) (void)hipFree(s->d_gathered[g]);
    }
    s->d_packed.assign(s->n, nullptr);
    s->d_gathered.assign(s->n, nullptr);
    if (s->n) {
        pb::DeviceGuard guard(s->devices[0]);
        (void)hipFree(s->d_out_ids);
        (void)hipFree(s->d_out_dist);
        (void)hipFree(s->d_out_count);
        if (s->h_out_ids) (void)hipHostFree(s->h_out_ids);
        if (s->h_out_dist) (void)hipHostFree(s->h_out_dist);
        if (s->h_out_count) (void)hipHostFree(s->h_out_count);
    }
    s->d_out_ids = nullptr;
    s->d_out_dist = nullptr;
    s->d_out_count = nullptr;
    s->h_out_ids = nullptr;
    s->h_out_dist = nullptr;
    s->h_out_count = nullptr;
    s->nq_cap = s->k_cap = 0;
}

// message / merge buffers for batches of up to nq queries, k results (grow-only)
int ensure_buffers(pb_sharded *s, uint32_t nq, uint32_t k) {
    if (nq <= s->nq_cap && k <= s->k_cap) return PB_OK;
    const uint32_t nq_cap = std::max<uint32_t>(std::max(nq, s->nq_cap), 64), k_cap = std::max(k, s->k_cap);
    free_buffers(s);
    const size_t msg = (size_t)nq_cap * (2 * (size_t)k_cap + 1) * sizeof(int64_t);
    for (int g = 0; g < s->n; ++g) {
        pb::DeviceGuard guard(s->devices[g]);
        PB_HIP(hipMalloc(&s->d_packed[g], msg));
        PB_HIP(hipMalloc(&s->d_gathered[g], msg * s->n));
    }
    pb::DeviceGuard guard(s->devices[0]);
    PB_HIP(hipMalloc(&s->d_out_ids, (size_t)nq_cap * k_cap * sizeof(int64_t)));
    PB_HIP(hipMalloc(&s->d_out_dist, (size_t)nq_cap * k_cap * sizeof(float)));
    PB_HIP(hipMalloc(&s->d_out_count, (size_t)nq_cap * sizeof(uint32_t)));
    PB_HIP(hipHostMalloc(&s->h_out_ids, (size_t)nq_cap * k_cap * sizeof(int64_t), hipHostMallocDefault));
    PB_HIP(hipHostMalloc(&s->h_out_dist, (size_t)nq_cap * k_cap * sizeof(float), hipHostMallocDefault));
    PB_HIP(hipHostMalloc(&s->h_out_count, (size_t)nq_cap * sizeof(uint32_t), hipHostMallocDefault));
    s->nq_cap = nq_cap;
    s->k_cap = k_cap;
    return PB_OK;
}

// run fn(g) for every shard on its worker thread; first failure wins
int for_each_shard(pb_sharded *s, const std::function<int(int)> &fn) {
    for (int g = 0; g < s->n; ++g) s->workers[g]->submit([&fn, g] { return fn(g); });
    int rc = PB_OK;
    for (int g = 0; g < s->n; ++g) {
        const int r = s->workers[g]->wait();
        if (r && !rc) {
            rc = r;
            snprintf(pb::tls_error(), 512, "shard %d (device %d): %s", g, s->devices[g], s->workers[g]->err);
        }
    }
    return rc;
}

uint64_t shard_size(pb_sharded *s, int g) {
    uint64_t n = 0;
    (void)pb_index_size(s->shards[g], &n);
    return n;
}
// rows of shard g that count against the capacities (caller holds s->mu)
uint64_t shard_used(pb_sharded *s, int g) {
    if (s->pending.size() != (size_t)s->n) s->pending.assign(s->n, 0);
    if (s->base.size() != (size_t)s->n) s->base.assign(s->n, 0);
    return s->pending[g] ? s->base[g] + s->pending[g] : shard_size(s, g);
}

// exchange of the per-shard messages: after it d_gathered[0] holds message g at [g * count, (g+1) * count)
int exchange(pb_sharded *s, size_t count) {
    ++s->n_exchanges;
    if (s->use_rccl) {
        PB_NCCL(s, s->rccl.GroupStart());
        for (int g = 0; g < s->n; ++g) {
            pb::DeviceGuard guard(s->devices[g]);
            ncclResult_t r = s->rccl.AllGather(s->d_packed[g], s->d_gathered[g], count, ncclInt64, s->comms[g], s->streams[g]);
            if (r != ncclSuccess) {
                (void)s->rccl.GroupEnd();
                return pb::fail(PB_ERR_HIP, "ncclAllGather (shard %d) -> %s", g, s->rccl.GetErrorString(r));
            }
        }
        PB_NCCL(s, s->rccl.GroupEnd());
        // every device took part and received; only shard 0's copy is merged, the others are waited for so that the
        // next batch may overwrite their send buffers
        for (int g = 1; g < s->n; ++g) {
            pb::DeviceGuard guard(s->devices[g]);
            PB_HIP(hipStreamSynchronize(s->streams[g]));
        }
        return PB_OK;
    }
    // shards that share a device (test topology) or a single shard: plain device-to-device copies to the merging shard
    pb::DeviceGuard guard(s->devices[0]);
    for (int g = 0; g < s->n; ++g)
        PB_HIP(hipMemcpyAsync(s->d_gathered[0] + (size_t)g * count, s->d_packed[g], count * sizeof(int64_t), hipMemcpyDeviceToDevice,
                              s->streams[0]));
    return PB_OK;
}

void destroy(pb_sharded *s) {
    for (Worker *w : s->workers) {
        w->stop();
        delete w;
    }
    s->workers.clear();
    if (s->use_rccl)
        for (ncclComm_t c : s->comms)
            if (c) (void)s->rccl.CommDestroy(c);
    free_buffers(s);
    for (int g = 0; g < (int)s->streams.size(); ++g)
        if (s->streams[g]) {
            pb::DeviceGuard guard(s->devices[g]);
            (void)hipStreamDestroy(s->streams[g]);
        }
    for (pb_index *ix : s->shards) (void)pb_index_destroy(ix);
    // the RCCL handle is left loaded (other communicators of the process may live on)
}

}  // namespace

extern "C" {

int pb_sharded_create(pb_sharded **out, const int *device_ids, int n_devices, uint32_t dim, uint64_t capacity_rows) {
    PB_CHECK(out, PB_ERR_INVALID, "pb_sharded_create: null out pointer");
    *out = nullptr;
    PB_CHECK(device_ids && n_devices >= 1 && n_devices <= 64, PB_ERR_INVALID, "pb_sharded_create: need 1..64 device ids");
    PB_CHECK(capacity_rows >= 1, PB_ERR_INVALID, "pb_sharded_create: capacity_rows must be >= 1");
    pb_sharded *s = new (std::nothrow) pb_sharded();
    PB_CHECK(s, PB_ERR_NOMEM, "out of host memory");
    s->dim = dim;
    s->capacity = capacity_rows;
    s->n = n_devices;
    s->devices.assign(device_ids, device_ids + n_devices);
    s->d_packed.assign(s->n, nullptr);
    s->d_gathered.assign(s->n, nullptr);
    auto body = [&]() -> int {
        const uint64_t per = (capacity_rows + (uint64_t)s->n - 1) / (uint64_t)s->n;
        for (int g = 0; g < s->n; ++g) {
            pb_index *ix = nullptr;
            int rc = pb_index_create(&ix, s->devices[g], dim, per);
            if (rc) return rc;
            s->shards.push_back(ix);
            s->shard_cap.push_back(per);
            pb::DeviceGuard guard(s->devices[g]);
            hipStream_t st = nullptr;
            PB_HIP(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
            s->streams.push_back(st);
        }
        std::vector<int> uniq(s->devices);
        std::sort(uniq.begin(), uniq.end());
        const bool distinct = std::adjacent_find(uniq.begin(), uniq.end()) == uniq.end();
        // one shard: RCCL only on request (PB_SHARDED_FORCE_RCCL=1 lets a 1-GPU box run the real collective path)
        s->use_rccl = distinct && (s->n > 1 || getenv("PB_SHARDED_FORCE_RCCL") != nullptr);
        if (s->use_rccl) {
            int rc = load_rccl(&s->rccl);
            if (rc) return rc;
            s->comms.assign(s->n, nullptr);
            PB_NCCL(s, s->rccl.CommInitAll(s->comms.data(), s->n, s->devices.data()));
        }
        for (int g = 0; g < s->n; ++g) {
            Worker *w = new (std::nothrow) Worker();
            PB_CHECK(w, PB_ERR_NOMEM, "out of host memory");
            s->workers.push_back(w);
            w->start();
        }
        return PB_OK;
    };
    int rc = body();
    if (rc) {
        destroy(s);
        delete s;
        return rc;
    }
    *out = s;
    return PB_OK;
}

int pb_sharded_destroy(pb_sharded *s) {
    if (!s) return PB_OK;
    destroy(s);
    delete s;
    return PB_OK;
}

int pb_sharded_info(const pb_sharded *s, int *n_shards, int *uses_rccl, uint64_t *n_exchanges) {
    PB_CHECK(s, PB_ERR_INVALID, "pb_sharded_info: null handle");
    if (n_shards) *n_shards = s->n;
    if (uses_rccl) *uses_rccl = s->use_rccl ? 1 : 0;
    if (n_exchanges) *n_exchanges = s->n_exchanges;
    return PB_OK;
}

int pb_sharded_size(pb_sharded *s, uint64_t *n_rows, uint64_t *per_shard) {
    PB_CHECK(s && n_rows, PB_ERR_INVALID, "pb_sharded_size: null pointer");
    std::lock_guard<std::mutex> lock(s->mu);
    uint64_t tot = 0;
    for (int g = 0; g < s->n; ++g) {
        const uint64_t c = shard_size(s, g);
        if (per_shard) per_shard[g] = c;
        tot += c;
    }
    *n_rows = tot;
    return PB_OK;
}

// Engine::open (engine.rs:117-145): contiguous row ranges, ceil(n / G) rows per shard
int pb_sharded_load(pb_sharded *s, const int64_t *image_ids, const uint8_t *rows, uint64_t n) {
    PB_CHECK(s, PB_ERR_INVALID, "pb_sharded_load: null handle");
    PB_CHECK(n == 0 || (image_ids && rows), PB_ERR_INVALID, "pb_sharded_load: null ids/rows");
    PB_CHECK(n <= s->capacity, PB_ERR_CAPACITY, "pb_sharded_load: %llu rows > capacity %llu", (unsigned long long)n,
             (unsigned long long)s->capacity);
    for (uint64_t i = 1; i < n; ++i)
        PB_CHECK(image_ids[i] > image_ids[i - 1], PB_ERR_INVALID, "pb_sharded_load: image_ids must be strictly increasing (row %llu)",
                 (unsigned long long)i);
    std::lock_guard<std::mutex> lock(s->mu);
    const uint64_t per = (n + (uint64_t)s->n - 1) / (uint64_t)s->n;
    return for_each_shard(s, [&](int g) -> int {
        const uint64_t lo = std::min<uint64_t>((uint64_t)g * per, n), hi = std::min<uint64_t>(lo + per, n);
        return pb_index_load(s->shards[g], image_ids + lo, rows + lo * s->dim, hi - lo);
    });
}

// INSERT OR IGNORE (engine.rs:251-256) over the shards: a pair whose image_id is stored on ANY shard is skipped; the new
// pairs of a call go to the least-full shard (ids are carried explicitly, so placement is free), spilling to the next
// one when it fills up.
int pb_sharded_append(pb_sharded *s, const int64_t *image_ids, const uint8_t *rows, uint64_t n, uint64_t *n_inserted) {
    PB_CHECK(s, PB_ERR_INVALID, "pb_sharded_append: null handle");
    PB_CHECK(n == 0 || (image_ids && rows), PB_ERR_INVALID, "pb_sharded_append: null ids/rows");
    if (n_inserted) *n_inserted = 0;
    if (n == 0) return PB_OK;
    std::lock_guard<std::mutex> lock(s->mu);
    const size_t d = s->dim;
    // new pairs only: not stored on any shard, first occurrence within the call
    std::vector<int64_t> ids;
    std::vector<uint8_t> data;
    std::unordered_set<int64_t> seen;
    for (uint64_t i = 0; i < n; ++i) {
        int found = 0;
        for (int g = 0; g < s->n && !found; ++g) {
            int rc = pb_index_contains(s->shards[g], image_ids[i], &found);
            if (rc) return rc;
        }
        if (found || s->inflight.count(image_ids[i]) || !seen.insert(image_ids[i]).second) continue;  // (an id a device-side append is storing right now is taken)
        ids.push_back(image_ids[i]);
        data.insert(data.end(), rows + i * d, rows + (i + 1) * d);
    }
    uint64_t stored = 0, pos = 0;
    uint64_t room = s->capacity;  // the TOTAL is the contract (per-shard capacities round up: their sum may exceed it)
    for (int g = 0; g < s->n; ++g) room -= std::min<uint64_t>(room, shard_used(s, g));
    while (pos < ids.size()) {
        if (room == 0) {
            if (n_inserted) *n_inserted = stored;
            return pb::fail(PB_ERR_CAPACITY, "pb_sharded_append: index full: capacity %llu rows (%llu rows of this call stored)",
                            (unsigned long long)s->capacity, (unsigned long long)stored);
        }
        int best = -1;
        uint64_t best_free = 0, best_size = 0;
        for (int g = 0; g < s->n; ++g) {
            const uint64_t sz = shard_used(s, g), fr = s->shard_cap[g] > sz ? s->shard_cap[g] - sz : 0;
            if (fr && (best < 0 || sz < best_size)) {
                best = g;
                best_free = fr;
                best_size = sz;
            }
        }
        if (best < 0) {
            if (n_inserted) *n_inserted = stored;
            return pb::fail(PB_ERR_CAPACITY, "pb_sharded_append: every shard is full (%llu rows stored of this call)", (unsigned long long)stored);
        }
        const uint64_t take = std::min<uint64_t>(std::min<uint64_t>(best_free, room), ids.size() - pos);
        uint64_t got = 0;
        int rc = pb_index_append(s->shards[best], ids.data() + pos, data.data() + pos * d, take, &got);
        if (s->pending[best]) s->base[best] += got;  // a shard with device-side appends in flight is counted by its base
        stored += got;
        room -= std::min<uint64_t>(room, got);
        if (rc) {
            if (n_inserted) *n_inserted = stored;
            return rc;
        }
        pos += take;
    }
    if (n_inserted) *n_inserted = stored;
    return PB_OK;
}

// synthetic table for benchmarks: rows [0, n) of stream `seed`, contiguous ranges per shard, generated on the devices
int pb_sharded_shard_device(const pb_sharded *s, int shard, int *device) {
    PB_CHECK(s && device, PB_ERR_INVALID, "pb_sharded_shard_device: null argument");
    PB_CHECK(shard >= 0 && shard < s->n, PB_ERR_INVALID, "pb_sharded_shard_device: shard %d of %d", shard, s->n);
    *device = s->devices[shard];
    return PB_OK;
}

int pb_sharded_contains(pb_sharded *s, int64_t image_id, int *found) {
    PB_CHECK(s && found, PB_ERR_INVALID, "pb_sharded_contains: null argument");
    *found = 0;
    std::lock_guard<std::mutex> lock(s->mu);
    for (int g = 0; g < s->n && !*found; ++g) {
        int rc = pb_index_contains(s->shards[g], image_id, found);
        if (rc) return rc;
    }
    return PB_OK;
}

int pb_sharded_append_device(pb_sharded *s, int shard, const int64_t *image_ids, const uint8_t *d_rows, uint64_t n) {
    PB_CHECK(s, PB_ERR_INVALID, "pb_sharded_append_device: null handle");
    PB_CHECK(shard >= 0 && shard < s->n, PB_ERR_INVALID, "pb_sharded_append_device: shard %d of %d", shard, s->n);
    PB_CHECK(n == 0 || (image_ids && d_rows), PB_ERR_INVALID, "pb_sharded_append_device: null ids/rows");
    if (n == 0) return PB_OK;
    {
        // bookkeeping under the table's lock (the device copy below runs under the SHARD's lock only, so the embed threads
        // of different devices insert concurrently): capacity, and the ids must be new to EVERY shard -- fresh
        // last_insert_rowid() values (engine.rs:233,249); an update or a re-insert goes through pb_sharded_append.  Rows and ids
        // of appends still in flight count as stored.
        std::lock_guard<std::mutex> lock(s->mu);
        uint64_t total = 0;
        for (int g = 0; g < s->n; ++g) total += shard_used(s, g);
        PB_CHECK(total + n <= s->capacity, PB_ERR_CAPACITY, "pb_sharded_append_device: %llu rows + %llu > capacity %llu",
                 (unsigned long long)total, (unsigned long long)n, (unsigned long long)s->capacity);
        PB_CHECK(shard_used(s, shard) + n <= s->shard_cap[shard], PB_ERR_CAPACITY,
                 "pb_sharded_append_device: shard %d is full (%llu of %llu rows)", shard, (unsigned long long)shard_used(s, shard),
                 (unsigned long long)s->shard_cap[shard]);
        for (uint64_t i = 0; i < n; ++i) {
            PB_CHECK(!s->inflight.count(image_ids[i]), PB_ERR_INVALID, "pb_sharded_append_device: image_id %lld is being stored by a concurrent call",
                     (long long)image_ids[i]);
            for (int g = 0; g < s->n; ++g) {
                if (g == shard) continue;  // the shard itself checks its own ids (ascending, beyond everything stored)
                int found = 0;
                int rc = pb_index_contains(s->shards[g], image_ids[i], &found);
                if (rc) return rc;
                PB_CHECK(!found, PB_ERR_INVALID, "pb_sharded_append_device: image_id %lld is already stored on shard %d", (long long)image_ids[i], g);
            }
        }
        for (uint64_t i = 1; i < n; ++i)  // (checked again by the shard; here so that nothing is reserved for a call that will fail)
            PB_CHECK(image_ids[i] > image_ids[i - 1], PB_ERR_INVALID, "pb_sharded_append_device: image_ids must be strictly ascending (row %llu)",
                     (unsigned long long)i);
        s->inflight.insert(image_ids, image_ids + n);
        if (s->pending[shard] == 0) s->base[shard] = shard_size(s, shard);  // nothing in flight on this shard: its live size is exact
        s->pending[shard] += n;
    }
    const int rc = pb_index_append_device(s->shards[shard], image_ids, d_rows, n);
    {
        std::lock_guard<std::mutex> lock(s->mu);
        for (uint64_t i = 0; i < n; ++i) s->inflight.erase(image_ids[i]);
        s->pending[shard] -= n;
        if (rc == PB_OK) s->base[shard] += n;  // committed (pb_index_append_device stores all n rows or none)
    }
    return rc;
}

int pb_sharded_fill_synthetic(pb_sharded *s, uint64_t seed, uint64_t n, int64_t first_id) {
    PB_CHECK(s, PB_ERR_INVALID, "pb_sharded_fill_synthetic: null handle");
    PB_CHECK(n <= s->capacity, PB_ERR_CAPACITY, "pb_sharded_fill_synthetic: %llu rows > capacity", (unsigned long long)n);
    std::lock_guard<std::mutex> lock(s->mu);
    const uint64_t per = (n + (uint64_t)s->n - 1) / (uint64_t)s->n;
    return for_each_shard(s, [&](int g) -> int {
        const uint64_t lo = std::min<uint64_t>((uint64_t)g * per, n), hi = std::min<uint64_t>(lo + per, n);
        if (hi == lo) return PB_OK;
        return pb_index_fill_synthetic(s->shards[g], seed, lo, hi - lo, first_id + (int64_t)lo);
    });
}

int pb_sharded_set_option(pb_sharded *s, int option, int64_t value) {
    PB_CHECK(s, PB_ERR_INVALID, "pb_sharded_set_option: null handle");
    PB_CHECK(option != PB_OPT_STREAM, PB_ERR_INVALID, "pb_sharded_set_option: shards keep their own streams");
    std::lock_guard<std::mutex> lock(s->mu);
    for (int g = 0; g < s->n; ++g) {
        int rc = pb_index_set_option(s->shards[g], option, value);
        if (rc) return rc;
    }
    return PB_OK;
}

int pb_sharded_get_stats(pb_sharded *s, pb_scan_stats *out, int reset) {
    PB_CHECK(s && out, PB_ERR_INVALID, "pb_sharded_get_stats: null pointer");
    std::lock_guard<std::mutex> lock(s->mu);
    pb_scan_stats tot{};
    for (int g = 0; g < s->n; ++g) {
        pb_scan_stats st{};
        int rc = pb_index_get_stats(s->shards[g], &st, reset);
        if (rc) return rc;
        tot.queries += st.queries;
        tot.fast_path += st.fast_path;
        tot.fallback += st.fallback;
        tot.second_chance += st.second_chance;
        tot.stamp_timeouts += st.stamp_timeouts;
        tot.profiled_launches += st.profiled_launches;
        tot.profiled_ms += st.profiled_ms;
        tot.profiled_bytes += st.profiled_bytes;
    }
    *out = tot;
    return PB_OK;
}

// Engine::query_by_image_hash_from_image (engine.rs:363-396) over the shards: per-shard top-k, one all-gather, device merge
int pb_sharded_search(pb_sharded *s, const uint8_t *queries, uint32_t nq, uint32_t k, double max_dist, int64_t *out_ids,
                      float *out_dist, uint32_t *out_count) {
    PB_CHECK(s, PB_ERR_INVALID, "pb_sharded_search: null handle");
    PB_CHECK(k >= 1 && k <= PB_MAX_K, PB_ERR_INVALID, "pb_sharded_search: k = %u outside 1..%u", k, PB_MAX_K);
    PB_CHECK(nq == 0 || (queries && out_ids && out_dist && out_count), PB_ERR_INVALID, "pb_sharded_search: null buffer");
    if (nq == 0) return PB_OK;
    std::lock_guard<std::mutex> lock(s->mu);
    const uint32_t batch_max = 1024;  // queries per exchange
    const size_t row = 2 * (size_t)k + 1;
    for (uint32_t q0 = 0; q0 < nq; q0 += batch_max) {
        const uint32_t cq = std::min(batch_max, nq - q0);
        int rc = ensure_buffers(s, cq, k);
        if (rc) return rc;
        const uint8_t *qb = queries + (size_t)q0 * s->dim;
        // every shard answers the batch on its own GPU, concurrently; the messages stay in HBM
        rc = for_each_shard(s, [&](int g) -> int { return pb_index_search_packed(s->shards[g], qb, cq, k, max_dist, s->d_packed[g]); });
        if (rc) return rc;
        rc = exchange(s, (size_t)cq * row);
        if (rc) return rc;
        pb::DeviceGuard guard(s->devices[0]);
        // the merge kernel writes the k results per query straight into pinned host memory (device-visible): no copy
        // commands behind it, one wait
        hipLaunchKernelGGL(pbm::k_merge_packed, dim3(cq), dim3(pbm::MERGE_BLOCK), 0, s->streams[0], s->d_gathered[0], (uint32_t)s->n, cq, k,
                           s->h_out_ids, s->h_out_dist, s->h_out_count);
        PB_HIP(hipGetLastError());
        PB_HIP(hipStreamSynchronize(s->streams[0]));
        memcpy(out_ids + (size_t)q0 * k, s->h_out_ids, (size_t)cq * k * sizeof(int64_t));
        memcpy(out_dist + (size_t)q0 * k, s->h_out_dist, (size_t)cq * k * sizeof(float));
        memcpy(out_count + q0, s->h_out_count, (size_t)cq * sizeof(uint32_t));
    }
    return PB_OK;
}

// The merge step alone, for a host that runs its own exchange (bench.py under torchrun: one process per GPU, the
// all-gather through torch.distributed): d_gathered = int64[n_lists][nq][2k+1] in DEVICE memory of `device`, as
// all-gathered from pb_index_search_packed; results to HOST buffers.
int pb_topk_merge_packed_device(int device, const int64_t *d_gathered, uint32_t n_lists, uint32_t nq, uint32_t k, int64_t *out_ids,
                                float *out_dist, uint32_t *out_count) {
    PB_CHECK(nq == 0 || (d_gathered && out_ids && out_dist && out_count), PB_ERR_INVALID, "pb_topk_merge_packed_device: null buffer");
    PB_CHECK(k >= 1 && k <= PB_MAX_K && n_lists >= 1 && n_lists <= 64, PB_ERR_INVALID, "pb_topk_merge_packed_device: k or n_lists out of range");
    if (nq == 0) return PB_OK;
    pb::DeviceGuard guard(device);
    PB_CHECK(guard.ok, PB_ERR_HIP, "hipSetDevice(%d) failed", device);
    // scratch per device, grown on demand and kept for the life of the process: this entry point sits in a per-batch
    // loop (bench.py under torchrun), where a hipMalloc / hipFree triple per call would cost more than the merge
    struct Scratch {
        int64_t *d_ids = nullptr, *h_ids = nullptr;
        float *d_dist = nullptr, *h_dist = nullptr;
        uint32_t *d_cnt = nullptr, *h_cnt = nullptr;
        size_t cap = 0;  // in (query, k) slots
        size_t cap_q = 0;
    };
    static std::mutex mu;
    static Scratch scratch[64];
    PB_CHECK(device >= 0 && device < 64, PB_ERR_INVALID, "pb_topk_merge_packed_device: device %d", device);
    std::lock_guard<std::mutex> lock(mu);
    Scratch &S = scratch[device];
    const size_t slots = (size_t)nq * k;
    if (slots > S.cap || nq > S.cap_q) {
        (void)hipFree(S.d_ids); (void)hipFree(S.d_dist); (void)hipFree(S.d_cnt);
        if (S.h_ids) (void)hipHostFree(S.h_ids);
        if (S.h_dist) (void)hipHostFree(S.h_dist);
        if (S.h_cnt) (void)hipHostFree(S.h_cnt);
        S = Scratch{};
        const size_t want = std::max<size_t>(slots, (size_t)64 * PB_MAX_K), want_q = std::max<size_t>(nq, 64);
        PB_HIP(hipMalloc(&S.d_ids, want * sizeof(int64_t)));
        PB_HIP(hipMalloc(&S.d_dist, want * sizeof(float)));
        PB_HIP(hipMalloc(&S.d_cnt, want_q * sizeof(uint32_t)));
        PB_HIP(hipHostMalloc(&S.h_ids, want * sizeof(int64_t), hipHostMallocDefault));
        PB_HIP(hipHostMalloc(&S.h_dist, want * sizeof(float), hipHostMallocDefault));
        PB_HIP(hipHostMalloc(&S.h_cnt, want_q * sizeof(uint32_t), hipHostMallocDefault));
        S.cap = want;
        S.cap_q = want_q;
    }
    // the kernel writes straight into pinned host memory: no copy commands, one wait
    hipLaunchKernelGGL(pbm::k_merge_packed, dim3(nq), dim3(pbm::MERGE_BLOCK), 0, nullptr, d_gathered, n_lists, nq, k, S.h_ids, S.h_dist, S.h_cnt);
    PB_HIP(hipGetLastError());
    PB_HIP(hipStreamSynchronize(nullptr));
    memcpy(out_ids, S.h_ids, slots * sizeof(int64_t));
    memcpy(out_dist, S.h_dist, slots * sizeof(float));
    memcpy(out_count, S.h_cnt, (size_t)nq * sizeof(uint32_t));
    return PB_OK;
}

}  // extern "C"
