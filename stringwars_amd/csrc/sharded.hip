// sharded.hip -- several GPUs of one node behind ONE scope of the C ABI (SURVEY 8b `swh_scope_init_gpus`, 8e).
//
// A multi-device scope owns one ordinary scope per device (own stream, scratch, plan buffers) and an RCCL
// communicator per device (`ncclCommInitAll`: single process, one rank per GPU). A pairwise batch is cut into
// contiguous, cells-balanced shards (prefix sum of len(a_i)*len(b_i), the reference's CUPS numerator); shard r is made
// resident and prepared on device r; a call scores every shard on its device with the ordinary engine entry points,
// then gathers the u32 distances to the first device with ONE group of ncclSend / ncclRecv over xGMI -- the single
// collective the north-star names -- and hands them to the caller. No collective inside the DP.
//
// RCCL is bound at run time (dlopen): the library itself stays loadable where no RCCL exists, and a process that already
// carries a copy (PyTorch bundles one) keeps exactly one. Members that share a device (a testing arrangement: N scopes on
// device 0 of a one-GPU box) exchange by device-to-device copies instead, since a communicator cannot hold a GPU twice.
#include <dlfcn.h>

#include <cstdio>
#include <cstring>
#include <memory>
#include <new>

#include "common.hpp"
#include "../../include/stringwars_amd_harness.h"

namespace swh {

// ---- RCCL, bound lazily ------------------------------------------------------------------------------------------
typedef void *ncclComm_t;
struct RcclApi {
    void *library = nullptr;
    int (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    int (*CommDestroy)(ncclComm_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*Send)(const void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    int (*Recv)(void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    bool ready() const { return CommInitAll && CommDestroy && GroupStart && GroupEnd && Send && Recv; }
};
constexpr int kNcclUint32 = 3;   // ncclDataType_t::ncclUint32 (rccl.h)

static RcclApi &rccl() {
    static RcclApi api = [] {
        RcclApi a;
        const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char *name : names)   // a copy the process already carries (e.g. PyTorch's) first
            if ((a.library = dlopen(name, RTLD_NOW | RTLD_NOLOAD))) break;
        for (const char *name : names) {
            if (a.library) break;
            a.library = dlopen(name, RTLD_NOW | RTLD_LOCAL);
        }
        if (!a.library) return a;
        a.CommInitAll = (decltype(a.CommInitAll))dlsym(a.library, "ncclCommInitAll");
        a.CommDestroy = (decltype(a.CommDestroy))dlsym(a.library, "ncclCommDestroy");
        a.GroupStart = (decltype(a.GroupStart))dlsym(a.library, "ncclGroupStart");
        a.GroupEnd = (decltype(a.GroupEnd))dlsym(a.library, "ncclGroupEnd");
        a.Send = (decltype(a.Send))dlsym(a.library, "ncclSend");
        a.Recv = (decltype(a.Recv))dlsym(a.library, "ncclRecv");
        a.GetErrorString = (decltype(a.GetErrorString))dlsym(a.library, "ncclGetErrorString");
        return a;
    }();
    return api;
}

struct MultiScope {
    std::vector<int> devices;
    std::vector<swh_scope_t> members;     // one ordinary scope per entry of `devices`
    std::vector<ncclComm_t> comms;        // empty when the members exchange by copies
    std::vector<hipEvent_t> done;         // per member: its shard has been scored
    std::vector<hipEvent_t> begin;
    hipEvent_t gathered = nullptr, gather_begin = nullptr;
    bool same_device = false;
    swh_shard_timing_t timing{};
};

static thread_local char g_sharded_error[512];
static swh_status_t sharded_fail(const char **error, swh_status_t status, const char *fmt, const char *a = "", const char *b = "") {
    snprintf(g_sharded_error, sizeof g_sharded_error, fmt, a, b);
    if (error) *error = g_sharded_error;
    return status;
}

// A batch made resident for a multi-device scope: shard r of both tapes uploaded to and prepared on device r.
struct ShardedPairs {
    Scope *scope = nullptr;
    std::vector<uint64_t> cuts;                 // members + 1 pair indices
    std::vector<swh_prepared_t> a, b;           // per member
    std::vector<uint32_t *> results;            // per member: device buffer of its shard's distances (member 0: inside `gathered`)
    uint32_t *gathered = nullptr;               // on the first device: all distances in pair order
    uint64_t pairs = 0, cells = 0;
    int utf8 = 0;
};

}  // namespace swh

using namespace swh;

// Cells-balanced contiguous cuts of a pairwise batch held in HOST tapes (pure host code; also what the CPU tests check).
template <typename Off>
static void shard_cuts(const Off *oa, const Off *ob, size_t count, size_t shards, size_t *cuts, uint64_t *cells_out) {
    std::vector<uint64_t> prefix(count + 1, 0);
    for (size_t i = 0; i < count; ++i) prefix[i + 1] = prefix[i] + (uint64_t)(oa[i + 1] - oa[i]) * (uint64_t)(ob[i + 1] - ob[i]);
    const uint64_t total = prefix[count];
    cuts[0] = 0;
    size_t at = 0;
    for (size_t r = 1; r < shards; ++r) {
        // first index whose prefix reaches r/shards of the cells; an all-empty batch falls back to equal counts
        const long double target = (long double)total * r / shards;
        if (total == 0) { cuts[r] = count * r / shards; continue; }
        while (at < count && (long double)prefix[at] < target) ++at;
        cuts[r] = at;
    }
    cuts[shards] = count;
    if (cells_out) *cells_out = total;
}

extern "C" {

swh_status_t swh_device_count(int *count) {
    if (!count) return swh_invalid_argument_k;
    *count = 0;
    if (hipGetDeviceCount(count) != hipSuccess) { (void)hipGetLastError(); *count = 0; }
    return swh_success_k;
}

void swh_shard_cuts_u64tape(const swh_tape_u64_t *a, const swh_tape_u64_t *b, size_t shards, size_t *cuts) {
    shard_cuts(a->offsets, b->offsets, a->count, shards ? shards : 1, cuts, nullptr);
}
void swh_shard_cuts_u32tape(const swh_tape_u32_t *a, const swh_tape_u32_t *b, size_t shards, size_t *cuts) {
    shard_cuts(a->offsets, b->offsets, a->count, shards ? shards : 1, cuts, nullptr);
}

swh_status_t swh_scope_init_gpus(const int *devices, int count, swh_scope_t *out, const char **error) {
    if (!out) return sharded_fail(error, swh_invalid_argument_k, "null scope pointer");
    *out = nullptr;
    if (!devices || count < 1 || count > 64) return sharded_fail(error, swh_invalid_argument_k, "between 1 and 64 devices");
    swh_scope_t parent = nullptr;
    swh_status_t status = swh_scope_init_gpu(devices[0], &parent, error);
    if (status != swh_success_k) return status;
    std::unique_ptr<MultiScope> multi(new MultiScope());
    multi->devices.assign(devices, devices + count);
    bool distinct = true;
    for (int i = 0; i < count; ++i)
        for (int j = 0; j < i; ++j) distinct = distinct && devices[i] != devices[j];
    multi->same_device = !distinct;
    auto cleanup = [&](swh_status_t st) {
        for (swh_scope_t m : multi->members) swh_scope_free(m);
        swh_scope_free(parent);
        return st;
    };
    for (int i = 0; i < count; ++i) {
        swh_scope_t member = nullptr;
        status = swh_scope_init_gpu(devices[i], &member, error);
        if (status != swh_success_k) return cleanup(status);
        swh_scope_set_async(member, 1);
        multi->members.push_back(member);
        hipEvent_t ev = nullptr, ev2 = nullptr;
        (void)hipSetDevice(devices[i]);
        if (hipEventCreate(&ev) != hipSuccess || hipEventCreate(&ev2) != hipSuccess)
            return cleanup(sharded_fail(error, swh_device_error_k, "hipEventCreate failed"));
        multi->done.push_back(ev);
        multi->begin.push_back(ev2);
    }
    (void)hipSetDevice(devices[0]);
    if (hipEventCreate(&multi->gathered) != hipSuccess || hipEventCreate(&multi->gather_begin) != hipSuccess)
        return cleanup(sharded_fail(error, swh_device_error_k, "hipEventCreate failed"));
    if (distinct && count > 1) {
        RcclApi &api = rccl();
        if (!api.ready()) return cleanup(sharded_fail(error, swh_rccl_error_k, "RCCL (librccl.so) could not be loaded: %s", dlerror() ? dlerror() : "symbols missing"));
        multi->comms.resize(count);
        const int rc = api.CommInitAll(multi->comms.data(), count, devices);
        if (rc != 0) {
            multi->comms.clear();
            return cleanup(sharded_fail(error, swh_rccl_error_k, "ncclCommInitAll failed: %s", api.GetErrorString ? api.GetErrorString(rc) : "?"));
        }
        for (int i = 0; i < count; ++i)   // peers read each other's result buffers only through RCCL; enable P2P anyway for fallbacks
            for (int j = 0; j < count; ++j)
                if (i != j) { (void)hipSetDevice(devices[i]); (void)hipDeviceEnablePeerAccess(devices[j], 0); (void)hipGetLastError(); }
        (void)hipSetDevice(devices[0]);
    }
    ((Scope *)parent)->multi = multi.release();
    *out = parent;
    return swh_success_k;
}

swh_status_t swh_scope_device_count(swh_scope_t handle, size_t *devices) {
    if (!handle || !devices) return swh_invalid_argument_k;
    const Scope *scope = (const Scope *)handle;
    *devices = scope->multi ? ((const MultiScope *)scope->multi)->devices.size() : 1;
    return swh_success_k;
}

swh_status_t swh_scope_shard_timing(swh_scope_t handle, swh_shard_timing_t *timing) {
    if (!handle || !timing) return swh_invalid_argument_k;
    const Scope *scope = (const Scope *)handle;
    *timing = scope->multi ? ((const MultiScope *)scope->multi)->timing : swh_shard_timing_t{};
    return swh_success_k;
}

}  // extern "C"

namespace swh {
void free_multi_scope(void *handle) {
    MultiScope *multi = (MultiScope *)handle;
    if (!multi) return;
    RcclApi &api = rccl();
    for (ncclComm_t comm : multi->comms)
        if (comm && api.CommDestroy) api.CommDestroy(comm);
    for (hipEvent_t ev : multi->done) (void)hipEventDestroy(ev);
    for (hipEvent_t ev : multi->begin) (void)hipEventDestroy(ev);
    if (multi->gathered) (void)hipEventDestroy(multi->gathered);
    if (multi->gather_begin) (void)hipEventDestroy(multi->gather_begin);
    for (swh_scope_t member : multi->members) swh_scope_free(member);
    delete multi;
}
}  // namespace swh

static void free_sharded(ShardedPairs *sp) {
    if (!sp) return;
    MultiScope *multi = sp->scope ? (MultiScope *)sp->scope->multi : nullptr;
    for (swh_prepared_t p : sp->a) swh_prepared_free(p);
    for (swh_prepared_t p : sp->b) swh_prepared_free(p);
    for (size_t r = 1; r < sp->results.size(); ++r)
        if (sp->results[r] && multi) { (void)hipSetDevice(multi->devices[r]); (void)hipFree(sp->results[r]); }
    if (sp->gathered && multi) { (void)hipSetDevice(multi->devices[0]); (void)hipFree(sp->gathered); }
    delete sp;
}

template <typename Tape, typename Off>
static swh_status_t sharded_prepare(swh_scope_t handle, const Tape *a, const Tape *b, int utf8, swh_sharded_t *out, const char **error,
                                    swh_status_t (*prepare)(swh_scope_t, const Tape *, int, swh_prepared_t *, const char **)) {
    if (!out) return sharded_fail(error, swh_invalid_argument_k, "null handle pointer");
    *out = nullptr;
    Scope *scope = (Scope *)handle;
    if (!scope || !scope->multi) return sharded_fail(error, swh_invalid_argument_k, "not a multi-device scope (swh_scope_init_gpus)");
    if (!a || !b || a->count != b->count) return sharded_fail(error, swh_invalid_argument_k, "two tapes of equal count");
    hipPointerAttribute_t attr;
    for (const void *p : {(const void *)a->offsets, (const void *)b->offsets, (const void *)a->data, (const void *)b->data}) {
        if (p && hipPointerGetAttributes(&attr, p) == hipSuccess && attr.type == hipMemoryTypeDevice)
            return sharded_fail(error, swh_invalid_argument_k, "sharding reads the tapes on the host: pass host (pageable, pinned or unified) memory");
        (void)hipGetLastError();
    }
    MultiScope *multi = (MultiScope *)scope->multi;
    const size_t members = multi->members.size();
    ShardedPairs *sp = new ShardedPairs();
    sp->scope = scope;
    sp->pairs = a->count;
    sp->utf8 = utf8;
    std::vector<size_t> cuts(members + 1);
    shard_cuts((const Off *)a->offsets, (const Off *)b->offsets, a->count, members, cuts.data(), &sp->cells);
    sp->cuts.assign(cuts.begin(), cuts.end());
    sp->a.assign(members, nullptr);
    sp->b.assign(members, nullptr);
    sp->results.assign(members, nullptr);
    std::vector<Off> rebased;
    auto prepare_shard = [&](const Tape *tape, size_t lo, size_t hi, swh_scope_t member, swh_prepared_t *prepared) {
        // the shard as a tape of its own: offsets rebased to its first byte, so that only its bytes travel to the device
        rebased.resize(hi - lo + 1);
        const Off base = tape->offsets[lo];
        for (size_t i = lo; i <= hi; ++i) rebased[i - lo] = (Off)(tape->offsets[i] - base);
        Tape view{tape->data ? tape->data + base : nullptr, rebased.data(), hi - lo};
        return prepare(member, &view, utf8, prepared, error);
    };
    for (size_t r = 0; r < members; ++r) {
        swh_status_t status = prepare_shard(a, cuts[r], cuts[r + 1], multi->members[r], &sp->a[r]);
        if (status == swh_success_k) status = prepare_shard(b, cuts[r], cuts[r + 1], multi->members[r], &sp->b[r]);
        if (status != swh_success_k) { free_sharded(sp); return status; }
    }
    (void)hipSetDevice(multi->devices[0]);
    if (hipMalloc((void **)&sp->gathered, (sp->pairs + 4) * sizeof(uint32_t)) != hipSuccess) {
        free_sharded(sp);
        return sharded_fail(error, swh_bad_alloc_k, "result vector on the first device");
    }
    sp->results[0] = sp->gathered + cuts[0];
    for (size_t r = 1; r < members; ++r) {
        (void)hipSetDevice(multi->devices[r]);
        if (hipMalloc((void **)&sp->results[r], (cuts[r + 1] - cuts[r] + 4) * sizeof(uint32_t)) != hipSuccess) {
            free_sharded(sp);
            return sharded_fail(error, swh_bad_alloc_k, "result buffer of a shard");
        }
    }
    (void)hipSetDevice(multi->devices[0]);
    *out = (swh_sharded_t)sp;
    return swh_success_k;
}

extern "C" {

swh_status_t swh_sharded_prepare_u64tape(swh_scope_t scope, const swh_tape_u64_t *a, const swh_tape_u64_t *b, int utf8,
                                         swh_sharded_t *sharded, const char **error) {
    return sharded_prepare<swh_tape_u64_t, uint64_t>(scope, a, b, utf8, sharded, error, swh_tape_prepare_u64);
}
swh_status_t swh_sharded_prepare_u32tape(swh_scope_t scope, const swh_tape_u32_t *a, const swh_tape_u32_t *b, int utf8,
                                         swh_sharded_t *sharded, const char **error) {
    return sharded_prepare<swh_tape_u32_t, uint32_t>(scope, a, b, utf8, sharded, error, swh_tape_prepare_u32);
}
swh_status_t swh_sharded_free(swh_sharded_t sharded) {
    free_sharded((ShardedPairs *)sharded);
    return swh_success_k;
}
swh_status_t swh_sharded_cuts(swh_sharded_t sharded, size_t *cuts, size_t capacity) {
    const ShardedPairs *sp = (const ShardedPairs *)sharded;
    if (!sp || !cuts || capacity < sp->cuts.size()) return swh_invalid_argument_k;
    for (size_t i = 0; i < sp->cuts.size(); ++i) cuts[i] = (size_t)sp->cuts[i];
    return swh_success_k;
}

swh_status_t swh_levenshtein_pairs_sharded(swh_levenshtein_t engine, swh_scope_t handle, swh_sharded_t sharded, uint32_t bound,
                                           uint32_t *out, const char **error) {
    Scope *scope = (Scope *)handle;
    ShardedPairs *sp = (ShardedPairs *)sharded;
    if (!scope || !scope->multi || !sp || sp->scope != scope) return sharded_fail(error, swh_invalid_argument_k, "scope and sharded batch do not belong together");
    if (!engine) return sharded_fail(error, swh_invalid_argument_k, "null engine");
    if (((const Engine *)engine)->kind != 0 || ((const Engine *)engine)->matrix_dev)
        return sharded_fail(error, swh_not_implemented_k, "sharded calls take Levenshtein engines with linear gap costs (no per-device tables)");
    if (!out && sp->pairs) return sharded_fail(error, swh_invalid_argument_k, "null output pointer");
    MultiScope *multi = (MultiScope *)scope->multi;
    const size_t members = multi->members.size();
    auto stream_of = [&](size_t r) { return ((Scope *)multi->members[r])->stream; };
    // every shard on its own device, asynchronously
    for (size_t r = 0; r < members; ++r) {
        const size_t n = (size_t)(sp->cuts[r + 1] - sp->cuts[r]);
        (void)hipSetDevice(multi->devices[r]);
        (void)hipEventRecord(multi->begin[r], stream_of(r));
        if (n) {
            swh_prepared_view_t va{sp->a[r], 0, n}, vb{sp->b[r], 0, n};
            swh_status_t status = swh_levenshtein_pairs_prepared(engine, multi->members[r], &va, &vb, bound, sp->results[r], 4, error);
            if (status != swh_success_k) {
                for (size_t q = 0; q <= r; ++q) swh_scope_synchronize(multi->members[q], nullptr);
                return status;
            }
        }
        (void)hipEventRecord(multi->done[r], stream_of(r));
    }
    // the one collective: distances of shards 1.. to the first device
    (void)hipSetDevice(multi->devices[0]);
    (void)hipEventRecord(multi->gather_begin, stream_of(0));
    if (!multi->comms.empty()) {
        RcclApi &api = rccl();
        int rc = api.GroupStart();
        for (size_t r = 1; r < members && rc == 0; ++r) {
            const size_t n = (size_t)(sp->cuts[r + 1] - sp->cuts[r]);
            if (!n) continue;
            rc = api.Recv(sp->gathered + sp->cuts[r], n, kNcclUint32, (int)r, multi->comms[0], stream_of(0));
            if (rc == 0) rc = api.Send(sp->results[r], n, kNcclUint32, 0, multi->comms[r], stream_of(r));
        }
        const int rc_end = api.GroupEnd();
        if (rc == 0) rc = rc_end;
        if (rc != 0) {
            for (size_t q = 0; q < members; ++q) swh_scope_synchronize(multi->members[q], nullptr);
            return sharded_fail(error, swh_rccl_error_k, "RCCL gather failed: %s", api.GetErrorString ? api.GetErrorString(rc) : "?");
        }
    } else {
        // members sharing one device (testing) or a single member: device-to-device copies ordered by events
        for (size_t r = 1; r < members; ++r) {
            const size_t n = (size_t)(sp->cuts[r + 1] - sp->cuts[r]);
            (void)hipStreamWaitEvent(stream_of(0), multi->done[r], 0);
            if (n && hipMemcpyAsync(sp->gathered + sp->cuts[r], sp->results[r], n * sizeof(uint32_t), hipMemcpyDeviceToDevice, stream_of(0)) != hipSuccess)
                return sharded_fail(error, swh_device_error_k, "device-to-device gather failed");
        }
    }
    (void)hipEventRecord(multi->gathered, stream_of(0));
    hipError_t err = hipSuccess;
    if (sp->pairs) err = hipMemcpyAsync(out, sp->gathered, sp->pairs * sizeof(uint32_t), hipMemcpyDefault, stream_of(0));
    for (size_t r = 0; r < members && err == hipSuccess; ++r) { (void)hipSetDevice(multi->devices[r]); err = hipStreamSynchronize(stream_of(r)); }
    (void)hipSetDevice(multi->devices[0]);
    if (err != hipSuccess) return sharded_fail(error, swh_device_error_k, "HIP error '%s' while gathering the shards", hipGetErrorString(err));
    for (size_t r = 0; r < members; ++r) swh_scope_synchronize(multi->members[r], nullptr);   // call summaries of the member scopes
    swh_shard_timing_t timing{};
    for (size_t r = 0; r < members; ++r) {
        float ms = 0;
        (void)hipSetDevice(multi->devices[r]);
        if (hipEventElapsedTime(&ms, multi->begin[r], multi->done[r]) == hipSuccess && ms > timing.compute_ms) timing.compute_ms = ms;
    }
    (void)hipSetDevice(multi->devices[0]);
    float gather_ms = 0;
    if (hipEventElapsedTime(&gather_ms, multi->gather_begin, multi->gathered) == hipSuccess) timing.gather_ms = gather_ms;
    (void)hipGetLastError();
    timing.cells = sp->cells;
    timing.pairs = sp->pairs;
    multi->timing = timing;
    return swh_success_k;
}

// One-shot convenience: shard, upload, score, gather, free. The steady state keeps the swh_sharded_t.
swh_status_t swh_levenshtein_pairs_sharded_u64tape(swh_levenshtein_t engine, swh_scope_t scope, const swh_tape_u64_t *a,
                                                   const swh_tape_u64_t *b, uint32_t bound, uint32_t *out, const char **error) {
    swh_sharded_t sharded = nullptr;
    swh_status_t status = swh_sharded_prepare_u64tape(scope, a, b, 0, &sharded, error);
    if (status != swh_success_k) return status;
    status = swh_levenshtein_pairs_sharded(engine, scope, sharded, bound, out, error);
    swh_sharded_free(sharded);
    return status;
}

}  // extern "C"
