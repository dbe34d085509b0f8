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

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <new>
#include <unordered_map>

#include <mutex>

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
    hipStream_t gather_stream = nullptr;   // first device: the receives run beside that device's own shard
    uint64_t *check_host = nullptr;        // pinned: per member (sum, xor) as computed on the shard's device, then as gathered
    std::vector<uint64_t *> check_dev;     // per member, on its device: 2 words; first device: 2 words per member more
    bool same_device = false;
    bool checked_once = false;             // the first call of a scope verifies its gather (swh_levenshtein_pairs_sharded)
    // alignment engines keep their tables on ONE device: per engine (by its uid), one clone per member, made on first use
    std::unordered_map<uint64_t, std::vector<void *>> engine_clones;
    swh_shard_timing_t timing{};
};

// every live multi-device scope, so that freeing an alignment engine can take its per-device clones along (a harness that builds an
// engine per row on a long-lived scope would otherwise pile up a 64 KB matrix + a class table per device and engine)
static std::mutex g_multi_mutex;
static std::vector<MultiScope *> g_multi_scopes;

// (sum, xor) of n distances: what the self-check of the sharded call compares between a shard's device and the gathered vector
__global__ __launch_bounds__(256) void k_shard_checksum(const uint32_t *values, uint64_t n, unsigned long long *out) {
    unsigned long long sum = 0, x = 0;
    for (uint64_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) { sum += values[i]; x ^= (unsigned long long)values[i] << (i & 31); }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) { sum += __shfl_xor(sum, off); x ^= __shfl_xor(x, off); }
    if ((threadIdx.x & 63) == 0) { atomicAdd(out, sum); atomicXor(out + 1, x); }
}

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

// A dense queries x candidates product made resident for a multi-device scope: the queries are cut into contiguous row
// blocks of equal symbol counts (every query meets every candidate, so cells per row are proportional to its length), block r
// and ALL candidates are prepared on device r, which also holds the block's rows of the result.
struct ShardedCross {
    Scope *scope = nullptr;
    std::vector<uint64_t> cuts;                 // members + 1 query indices
    std::vector<swh_prepared_t> queries, candidates;   // per member
    std::vector<uint64_t *> blocks;             // per member: device matrix of (cuts[r + 1] - cuts[r]) x columns 64-bit results
    uint64_t rows = 0, columns = 0, cells = 0;
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
        free_multi_scope(multi.release());   // members, events, streams, check buffers, communicators
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
        uint64_t *check = nullptr;
        if (hipSetDevice(devices[i]) != hipSuccess) return cleanup(sharded_fail(error, swh_device_error_k, "hipSetDevice failed"));
        if (hipEventCreate(&ev) == hipSuccess) multi->done.push_back(ev);
        if (hipEventCreate(&ev2) == hipSuccess) multi->begin.push_back(ev2);
        if (hipMalloc((void **)&check, (2 + (i == 0 ? 2 * (size_t)count : 0)) * sizeof(uint64_t)) == hipSuccess) multi->check_dev.push_back(check);
        if (!ev || !ev2 || !check) return cleanup(sharded_fail(error, swh_device_error_k, "per-device events / check words could not be created"));
    }
    if (hipSetDevice(devices[0]) != hipSuccess || hipEventCreate(&multi->gathered) != hipSuccess || hipEventCreate(&multi->gather_begin) != hipSuccess ||
        hipStreamCreateWithFlags(&multi->gather_stream, hipStreamNonBlocking) != hipSuccess ||
        hipHostMalloc((void **)&multi->check_host, 4 * (size_t)count * sizeof(uint64_t), hipHostMallocDefault) != hipSuccess)
        return cleanup(sharded_fail(error, swh_device_error_k, "gather stream / events of the first device could not be created"));
    // Test hooks (read per call): STRINGWARS_AMD_RCCL=off makes a scope of several members fail the way a missing RCCL does;
    // =force builds the communicator whatever the device list looks like -- one member (a one-rank communicator: the real
    // ncclCommInitAll / group / destroy calls on a one-GPU box) or members sharing a device (which RCCL refuses).
    const char *rccl_knob = test_hook("STRINGWARS_AMD_RCCL");
    const bool rccl_off = rccl_knob && strcmp(rccl_knob, "off") == 0, rccl_force = rccl_knob && strcmp(rccl_knob, "force") == 0;
    if (rccl_off && count > 1)
        return cleanup(sharded_fail(error, swh_rccl_error_k, "RCCL (librccl.so) could not be loaded: %s", "disabled by STRINGWARS_AMD_RCCL=off"));
    if ((distinct && count > 1) || rccl_force) {
        RcclApi &api = rccl();
        const char *why = api.ready() ? nullptr : dlerror();   // (dlerror() clears the state: read it once)
        if (!api.ready()) return cleanup(sharded_fail(error, swh_rccl_error_k, "RCCL (librccl.so) could not be loaded: %s", why ? why : "symbols missing"));
        multi->comms.resize(count);
        const int rc = api.CommInitAll(multi->comms.data(), count, devices);
        if (rc != 0) {
            multi->comms.clear();
            return cleanup(sharded_fail(error, swh_rccl_error_k, "ncclCommInitAll failed: %s", api.GetErrorString ? api.GetErrorString(rc) : "?"));
        }
        // (peers exchange results only through RCCL, which sets up its own peer mappings: no hipDeviceEnablePeerAccess here)
        if (hipSetDevice(devices[0]) != hipSuccess) return cleanup(sharded_fail(error, swh_device_error_k, "hipSetDevice failed"));
    }
    // rows of a sharded cross-product go straight from device r into a matrix that may live on the first device: peers are mapped
    // where the hardware allows (best effort: without it the runtime stages such copies itself)
    if (distinct && count > 1) {
        for (int r = 1; r < count; ++r) {
            int can = 0;
            if (hipDeviceCanAccessPeer(&can, devices[r], devices[0]) == hipSuccess && can && hipSetDevice(devices[r]) == hipSuccess) (void)hipDeviceEnablePeerAccess(devices[0], 0);
            (void)hipGetLastError();   // (already enabled: fine)
        }
        (void)hipSetDevice(devices[0]);
    }
    {
        std::lock_guard<std::mutex> lock(g_multi_mutex);
        g_multi_scopes.push_back(multi.get());
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
void drop_engine_clones(uint64_t engine_uid) {
    std::vector<void *> clones;
    {
        std::lock_guard<std::mutex> lock(g_multi_mutex);
        for (MultiScope *multi : g_multi_scopes) {
            auto it = multi->engine_clones.find(engine_uid);
            if (it == multi->engine_clones.end()) continue;
            clones.insert(clones.end(), it->second.begin(), it->second.end());
            multi->engine_clones.erase(it);
        }
    }
    for (void *clone : clones) if (clone) swh_nw_free((swh_nw_t)clone);   // (a clone has a uid of its own and no clones: no recursion)
}
void free_multi_scope(void *handle) {
    MultiScope *multi = (MultiScope *)handle;
    if (!multi) return;
    {
        std::lock_guard<std::mutex> lock(g_multi_mutex);
        for (size_t i = 0; i < g_multi_scopes.size(); ++i)
            if (g_multi_scopes[i] == multi) { g_multi_scopes.erase(g_multi_scopes.begin() + i); break; }
    }
    RcclApi &api = rccl();
    for (ncclComm_t comm : multi->comms)
        if (comm && api.CommDestroy) api.CommDestroy(comm);
    for (hipEvent_t ev : multi->done) (void)hipEventDestroy(ev);
    for (hipEvent_t ev : multi->begin) (void)hipEventDestroy(ev);
    if (multi->gathered) (void)hipEventDestroy(multi->gathered);
    if (multi->gather_begin) (void)hipEventDestroy(multi->gather_begin);
    if (multi->gather_stream) (void)hipStreamDestroy(multi->gather_stream);
    if (multi->check_host) (void)hipHostFree(multi->check_host);
    for (size_t i = 0; i < multi->check_dev.size(); ++i) { (void)hipSetDevice(multi->devices[i]); (void)hipFree(multi->check_dev[i]); }
    if (!multi->devices.empty()) (void)hipSetDevice(multi->devices[0]);
    for (auto &entry : multi->engine_clones)
        for (void *clone : entry.second) if (clone) swh_nw_free((swh_nw_t)clone);
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

}  // extern "C"

// One sharded call: every member scores its shard (in pieces) with `score(member index, views, destination)`, the 32-bit
// results are gathered on the first device and handed to the caller. Levenshtein distances and alignment scores alike.
typedef swh_status_t (*ShardScore)(void *engine, swh_scope_t member, const swh_prepared_view_t *a, const swh_prepared_view_t *b,
                                   uint32_t bound, uint32_t *dst, const char **error);
static swh_status_t sharded_call(const std::vector<void *> &engines, ShardScore score, swh_scope_t handle, swh_sharded_t sharded, uint32_t bound,
                                 uint32_t *out, const char **error) {
    Scope *scope = (Scope *)handle;
    ShardedPairs *sp = (ShardedPairs *)sharded;
    if (!out && sp->pairs) return sharded_fail(error, swh_invalid_argument_k, "null output pointer");
    MultiScope *multi = (MultiScope *)scope->multi;
    const size_t members = multi->members.size();
    auto stream_of = [&](size_t r) { return ((Scope *)multi->members[r])->stream; };
    // Every HIP call is checked: a failure drains what was enqueued and comes back as swh_device_error_k with the call's name.
    hipError_t hip_error = hipSuccess;
    const char *hip_what = "";
    auto drain = [&]() {
        for (size_t q = 0; q < members; ++q) { (void)hipSetDevice(multi->devices[q]); (void)hipStreamSynchronize(stream_of(q)); swh_scope_synchronize(multi->members[q], nullptr); }
        (void)hipSetDevice(multi->devices[0]);
        (void)hipStreamSynchronize(multi->gather_stream);
        (void)hipGetLastError();
    };
#define SWH_SHARD_HIP(expr)                                                                                   \
    do {                                                                                                      \
        if (hip_error == hipSuccess) { hip_error = (expr); if (hip_error != hipSuccess) hip_what = #expr; }   \
    } while (0)
#define SWH_SHARD_BAIL()                                                                                                          \
    do {                                                                                                                          \
        if (hip_error != hipSuccess) {                                                                                            \
            drain();                                                                                                              \
            return sharded_fail(error, swh_device_error_k, "HIP error '%s' in the sharded call at %s", hipGetErrorString(hip_error), hip_what); \
        }                                                                                                                         \
    } while (0)
    // The first call of a scope verifies its gather end to end (STRINGWARS_AMD_SHARD_CHECK=1: every call, =0: never): (sum,
    // xor) of every shard's distances computed on the shard's device and again over its range of the gathered vector on the
    // first device. A first real multi-GPU run so validates the RCCL path by itself.
    static const int check_knob = [] { const char *e = getenv("STRINGWARS_AMD_SHARD_CHECK"); return e ? atoi(e) : -1; }();
    const bool self_check = check_knob == 1 || (check_knob != 0 && !multi->checked_once);
    // Shards of a quarter of a million pairs or more are scored in four pieces, the send of piece j enqueued behind its kernel
    // so that it travels while piece j + 1 is scored; the receives run on a stream of their own on the first device.
    constexpr size_t kPieces = 4, kMinPiecePairs = 65536;
    size_t largest = 0;
    for (size_t r = 0; r < members; ++r) largest = std::max<size_t>(largest, (size_t)(sp->cuts[r + 1] - sp->cuts[r]));
    const size_t pieces = largest >= kPieces * kMinPiecePairs ? kPieces : 1;
    auto piece_range = [&](size_t r, size_t j, size_t &lo, size_t &hi) {
        const size_t n = (size_t)(sp->cuts[r + 1] - sp->cuts[r]);
        lo = n * j / pieces; hi = n * (j + 1) / pieces;
    };
    SWH_SHARD_HIP(hipSetDevice(multi->devices[0]));
    SWH_SHARD_HIP(hipEventRecord(multi->gather_begin, stream_of(0)));
    SWH_SHARD_HIP(hipStreamWaitEvent(multi->gather_stream, multi->gather_begin, 0));   // receives overwrite `gathered`: not before the previous call's copy-out
    SWH_SHARD_BAIL();
    for (size_t j = 0; j < pieces; ++j) {
        for (size_t r = 0; r < members; ++r) {
            size_t lo, hi;
            piece_range(r, j, lo, hi);
            SWH_SHARD_HIP(hipSetDevice(multi->devices[r]));
            if (j == 0) SWH_SHARD_HIP(hipEventRecord(multi->begin[r], stream_of(r)));
            SWH_SHARD_BAIL();
            if (hi > lo) {
                swh_prepared_view_t va{sp->a[r], lo, hi - lo}, vb{sp->b[r], lo, hi - lo};
                swh_status_t status = score(engines[r], multi->members[r], &va, &vb, bound, sp->results[r] + lo, error);
                if (status != swh_success_k) { drain(); return status; }
            }
            if (j + 1 == pieces) SWH_SHARD_HIP(hipEventRecord(multi->done[r], stream_of(r)));
        }
        SWH_SHARD_BAIL();
        // piece j of shards 1.. to the first device
        if (!multi->comms.empty()) {
            RcclApi &api = rccl();
            int rc = api.GroupStart();
            for (size_t r = 1; r < members && rc == 0; ++r) {
                size_t lo, hi;
                piece_range(r, j, lo, hi);
                if (hi <= lo) continue;
                rc = api.Recv(sp->gathered + sp->cuts[r] + lo, hi - lo, kNcclUint32, (int)r, multi->comms[0], multi->gather_stream);
                if (rc == 0) rc = api.Send(sp->results[r] + lo, hi - lo, kNcclUint32, 0, multi->comms[r], stream_of(r));
            }
            const int rc_end = api.GroupEnd();
            if (rc == 0) rc = rc_end;
            if (rc != 0) {
                drain();
                return sharded_fail(error, swh_rccl_error_k, "RCCL gather failed: %s", api.GetErrorString ? api.GetErrorString(rc) : "?");
            }
        } else if (j + 1 == pieces) {
            // members sharing one device (testing) or a single member: device-to-device copies ordered by events
            SWH_SHARD_HIP(hipSetDevice(multi->devices[0]));
            for (size_t r = 1; r < members; ++r) {
                const size_t n = (size_t)(sp->cuts[r + 1] - sp->cuts[r]);
                SWH_SHARD_HIP(hipStreamWaitEvent(multi->gather_stream, multi->done[r], 0));
                if (n) SWH_SHARD_HIP(hipMemcpyAsync(sp->gathered + sp->cuts[r], sp->results[r], n * sizeof(uint32_t), hipMemcpyDeviceToDevice, multi->gather_stream));
            }
            SWH_SHARD_BAIL();
        }
    }
    if (self_check) {
        // on the shard's device, behind its last piece ...
        for (size_t r = 0; r < members; ++r) {
            const size_t n = (size_t)(sp->cuts[r + 1] - sp->cuts[r]);
            SWH_SHARD_HIP(hipSetDevice(multi->devices[r]));
            SWH_SHARD_HIP(hipMemsetAsync(multi->check_dev[r], 0, 2 * sizeof(uint64_t), stream_of(r)));
            if (n) hipLaunchKernelGGL(k_shard_checksum, dim3((unsigned)std::min<size_t>((n + 255) / 256, 1024)), dim3(256), 0, stream_of(r), sp->results[r], (uint64_t)n,
                                      (unsigned long long *)multi->check_dev[r]);
            SWH_SHARD_HIP(hipGetLastError());
            SWH_SHARD_HIP(hipMemcpyAsync(multi->check_host + 2 * r, multi->check_dev[r], 2 * sizeof(uint64_t), hipMemcpyDeviceToHost, stream_of(r)));
        }
        SWH_SHARD_BAIL();
    }
    // the first device's stream continues once the receives are in: the caller's copy, the checks over the gathered ranges
    SWH_SHARD_HIP(hipSetDevice(multi->devices[0]));
    SWH_SHARD_HIP(hipEventRecord(multi->gathered, multi->gather_stream));
    SWH_SHARD_HIP(hipStreamWaitEvent(stream_of(0), multi->gathered, 0));
    if (const char *fault = test_hook("STRINGWARS_AMD_SHARD_FAULT")) {   // test hook: damage one gathered distance of the last shard
        const size_t r = members - 1, n = (size_t)(sp->cuts[r + 1] - sp->cuts[r]);
        if (atoi(fault) && n) SWH_SHARD_HIP(hipMemsetAsync(sp->gathered + sp->cuts[r] + n / 2, 0x5A, sizeof(uint32_t), stream_of(0)));
    }
    if (self_check) {
        uint64_t *words = multi->check_dev[0] + 2;
        SWH_SHARD_HIP(hipMemsetAsync(words, 0, 2 * members * sizeof(uint64_t), stream_of(0)));
        for (size_t r = 0; r < members; ++r) {
            const size_t n = (size_t)(sp->cuts[r + 1] - sp->cuts[r]);
            if (n) hipLaunchKernelGGL(k_shard_checksum, dim3((unsigned)std::min<size_t>((n + 255) / 256, 1024)), dim3(256), 0, stream_of(0), sp->gathered + sp->cuts[r], (uint64_t)n,
                                      (unsigned long long *)(words + 2 * r));
        }
        SWH_SHARD_HIP(hipGetLastError());
        SWH_SHARD_HIP(hipMemcpyAsync(multi->check_host + 2 * members, words, 2 * members * sizeof(uint64_t), hipMemcpyDeviceToHost, stream_of(0)));
    }
    if (sp->pairs) SWH_SHARD_HIP(hipMemcpyAsync(out, sp->gathered, sp->pairs * sizeof(uint32_t), hipMemcpyDefault, stream_of(0)));
    for (size_t r = 0; r < members; ++r) { SWH_SHARD_HIP(hipSetDevice(multi->devices[r])); SWH_SHARD_HIP(hipStreamSynchronize(stream_of(r))); }
    SWH_SHARD_HIP(hipSetDevice(multi->devices[0]));
    SWH_SHARD_BAIL();
    for (size_t r = 0; r < members; ++r) {   // call summaries of the member scopes (a prepared tape changed under us: device_error)
        swh_status_t status = swh_scope_synchronize(multi->members[r], error);
        if (status != swh_success_k) { drain(); return status; }
    }
    if (self_check) {
        multi->checked_once = true;
        for (size_t r = 0; r < members; ++r)
            if (multi->check_host[2 * r] != multi->check_host[2 * members + 2 * r] || multi->check_host[2 * r + 1] != multi->check_host[2 * members + 2 * r + 1]) {
                char which[32];
                snprintf(which, sizeof which, "%zu", r);
                return sharded_fail(error, swh_device_error_k, "gathered distances of shard %s differ from what its device computed (checksum mismatch after the %s)",
                                    which, multi->comms.empty() ? "device-to-device copies" : "RCCL gather");
            }
    }
    swh_shard_timing_t timing{};
    for (size_t r = 0; r < members; ++r) {
        float ms = 0;
        SWH_SHARD_HIP(hipSetDevice(multi->devices[r]));
        if (hipEventElapsedTime(&ms, multi->begin[r], multi->done[r]) == hipSuccess && ms > timing.compute_ms) timing.compute_ms = ms;
    }
    SWH_SHARD_HIP(hipSetDevice(multi->devices[0]));
    float gather_ms = 0;
    if (hipEventElapsedTime(&gather_ms, multi->gather_begin, multi->gathered) == hipSuccess) timing.gather_ms = gather_ms;   // first kernel to last receive
    (void)hipGetLastError();
    SWH_SHARD_BAIL();
#undef SWH_SHARD_HIP
#undef SWH_SHARD_BAIL
    timing.cells = sp->cells;
    timing.pairs = sp->pairs;
    multi->timing = timing;
    return swh_success_k;
}

static swh_status_t sharded_arguments(swh_scope_t handle, swh_sharded_t sharded, const void *engine, const char **error) {
    Scope *scope = (Scope *)handle;
    ShardedPairs *sp = (ShardedPairs *)sharded;
    if (!scope || !scope->multi || !sp || sp->scope != scope) return sharded_fail(error, swh_invalid_argument_k, "scope and sharded batch do not belong together");
    if (!engine) return sharded_fail(error, swh_invalid_argument_k, "null engine");
    return swh_success_k;
}

// Alignment engines keep a substitution matrix (and its class table) on one device: a multi-device scope clones the engine
// per member on first use (keyed by the engine's uid; the clones live as long as the scope). The map is shared with drop_engine_clones
// (an engine freed on another thread): looked up and published under g_multi_mutex, the callers work on a COPY of the pointers; the clones
// are built and freed outside the lock (freeing one calls drop_engine_clones itself).
static swh_status_t engine_clones_of(MultiScope *multi, const Engine *source, std::vector<void *> &out, const char **error) {
    {
        std::lock_guard<std::mutex> lock(g_multi_mutex);
        auto it = multi->engine_clones.find(source->uid);
        if (it != multi->engine_clones.end() && !it->second.empty()) { out = it->second; return swh_success_k; }
    }
    std::vector<void *> fresh(multi->members.size(), nullptr);
    for (size_t r = 0; r < multi->members.size(); ++r) {
        const swh_status_t status = clone_alignment_engine(source, multi->members[r], &fresh[r], error);
        if (status != swh_success_k) {
            for (void *clone : fresh) if (clone) swh_nw_free((swh_nw_t)clone);
            return status;
        }
    }
    {
        std::lock_guard<std::mutex> lock(g_multi_mutex);
        std::vector<void *> &slot = multi->engine_clones[source->uid];
        if (slot.empty()) { slot = fresh; out = fresh; fresh.clear(); }
        else out = slot;   // (another thread published its clones first)
    }
    for (void *clone : fresh) if (clone) swh_nw_free((swh_nw_t)clone);
    return swh_success_k;
}

static swh_status_t alignment_sharded(int kind, void *engine, swh_scope_t handle, swh_sharded_t sharded, int32_t *out, const char **error) {
    swh_status_t status = sharded_arguments(handle, sharded, engine, error);
    if (status != swh_success_k) return status;
    const Engine *source = (const Engine *)engine;
    if (source->kind != kind || !source->matrix_host) return sharded_fail(error, swh_invalid_argument_k, "not an engine of this kind");
    if (((ShardedPairs *)sharded)->utf8) return sharded_fail(error, swh_not_implemented_k, "substitution-matrix scoring over UTF-8 code points (the matrix is indexed by bytes)");
    MultiScope *multi = (MultiScope *)((Scope *)handle)->multi;
    std::vector<void *> clones;
    status = engine_clones_of(multi, source, clones, error);
    if (status != swh_success_k) return status;
    ShardScore score = kind == 1
        ? (ShardScore)[](void *e, swh_scope_t m, const swh_prepared_view_t *a, const swh_prepared_view_t *b, uint32_t, uint32_t *dst, const char **err) {
              return swh_nw_pairs_prepared((swh_nw_t)e, m, a, b, (int32_t *)dst, 4, err); }
        : (ShardScore)[](void *e, swh_scope_t m, const swh_prepared_view_t *a, const swh_prepared_view_t *b, uint32_t, uint32_t *dst, const char **err) {
              return swh_sw_pairs_prepared((swh_sw_t)e, m, a, b, (int32_t *)dst, 4, err); };
    return sharded_call(clones, score, handle, sharded, SWH_UNBOUNDED, (uint32_t *)out, error);
}

extern "C" {

swh_status_t swh_levenshtein_pairs_sharded(swh_levenshtein_t engine, swh_scope_t handle, swh_sharded_t sharded, uint32_t bound,
                                           uint32_t *out, const char **error) {
    swh_status_t status = sharded_arguments(handle, sharded, engine, error);
    if (status != swh_success_k) return status;
    if (((const Engine *)engine)->kind != 0) return sharded_fail(error, swh_invalid_argument_k, "not a Levenshtein engine");
    // (Levenshtein engines hold no device tables: one engine serves every member)
    const std::vector<void *> engines(((MultiScope *)((Scope *)handle)->multi)->members.size(), (void *)engine);
    return sharded_call(engines, [](void *e, swh_scope_t m, const swh_prepared_view_t *a, const swh_prepared_view_t *b, uint32_t k, uint32_t *dst, const char **err) {
        return swh_levenshtein_pairs_prepared((swh_levenshtein_t)e, m, a, b, k, dst, 4, err); }, handle, sharded, bound, out, error);
}
swh_status_t swh_nw_pairs_sharded(swh_nw_t engine, swh_scope_t scope, swh_sharded_t sharded, int32_t *out, const char **error) {
    return alignment_sharded(1, engine, scope, sharded, out, error);
}
swh_status_t swh_sw_pairs_sharded(swh_sw_t engine, swh_scope_t scope, swh_sharded_t sharded, int32_t *out, const char **error) {
    return alignment_sharded(2, engine, scope, sharded, out, error);
}

static void free_sharded_cross(ShardedCross *sc) {
    if (!sc) return;
    MultiScope *multi = sc->scope ? (MultiScope *)sc->scope->multi : nullptr;
    for (swh_prepared_t p : sc->queries) swh_prepared_free(p);
    for (swh_prepared_t p : sc->candidates) swh_prepared_free(p);
    for (size_t r = 0; r < sc->blocks.size(); ++r)
        if (sc->blocks[r] && multi) { (void)hipSetDevice(multi->devices[r]); (void)hipFree(sc->blocks[r]); }
    if (multi) (void)hipSetDevice(multi->devices[0]);
    delete sc;
}

swh_status_t swh_sharded_cross_prepare_u64tape(swh_scope_t handle, const swh_tape_u64_t *queries, const swh_tape_u64_t *candidates, int utf8,
                                               swh_sharded_cross_t *out, const char **error) {
    if (!out) return sharded_fail(error, swh_invalid_argument_k, "null handle pointer");
    *out = nullptr;
    Scope *scope = (Scope *)handle;
    if (!scope || !scope->multi) return sharded_fail(error, swh_invalid_argument_k, "not a multi-device scope (swh_scope_init_gpus)");
    if (!queries || !candidates) return sharded_fail(error, swh_invalid_argument_k, "two tapes");
    hipPointerAttribute_t attr;
    for (const void *p : {(const void *)queries->offsets, (const void *)candidates->offsets, (const void *)queries->data, (const void *)candidates->data}) {
        if (p && hipPointerGetAttributes(&attr, p) == hipSuccess && attr.type == hipMemoryTypeDevice)
            return sharded_fail(error, swh_invalid_argument_k, "sharding reads the tapes on the host: pass host (pageable, pinned or unified) memory");
        (void)hipGetLastError();
    }
    MultiScope *multi = (MultiScope *)scope->multi;
    const size_t members = multi->members.size();
    ShardedCross *sc = new ShardedCross();
    sc->scope = scope;
    sc->rows = queries->count; sc->columns = candidates->count; sc->utf8 = utf8;
    // row blocks of equal query bytes (an all-empty query tape: equal counts)
    const uint64_t total = queries->count ? queries->offsets[queries->count] - queries->offsets[0] : 0;
    sc->cuts.assign(members + 1, 0);
    size_t at = 0;
    for (size_t r = 1; r < members; ++r) {
        if (total == 0) { sc->cuts[r] = queries->count * r / members; continue; }
        const long double target = (long double)total * r / members;
        while (at < queries->count && (long double)(queries->offsets[at] - queries->offsets[0]) < target) ++at;
        sc->cuts[r] = at;
    }
    sc->cuts[members] = queries->count;
    sc->cells = total * (candidates->count ? candidates->offsets[candidates->count] - candidates->offsets[0] : 0);
    sc->queries.assign(members, nullptr); sc->candidates.assign(members, nullptr); sc->blocks.assign(members, nullptr);
    std::vector<uint64_t> rebased;
    for (size_t r = 0; r < members; ++r) {
        const size_t lo = (size_t)sc->cuts[r], hi = (size_t)sc->cuts[r + 1];
        rebased.resize(hi - lo + 1);
        const uint64_t first = queries->offsets[lo];
        for (size_t i = lo; i <= hi; ++i) rebased[i - lo] = queries->offsets[i] - first;
        swh_tape_u64_t view{queries->data ? queries->data + first : nullptr, rebased.data(), hi - lo};
        swh_status_t status = swh_tape_prepare_u64(multi->members[r], &view, utf8, &sc->queries[r], error);
        if (status == swh_success_k) status = swh_tape_prepare_u64(multi->members[r], candidates, utf8, &sc->candidates[r], error);
        if (status != swh_success_k) { free_sharded_cross(sc); return status; }
        if (hipSetDevice(multi->devices[r]) != hipSuccess ||
            hipMalloc((void **)&sc->blocks[r], ((hi - lo) * sc->columns + 2) * sizeof(uint64_t)) != hipSuccess) {
            free_sharded_cross(sc);
            return sharded_fail(error, swh_bad_alloc_k, "result rows of a shard");
        }
    }
    (void)hipSetDevice(multi->devices[0]);
    *out = (swh_sharded_cross_t)sc;
    return swh_success_k;
}
swh_status_t swh_sharded_cross_free(swh_sharded_cross_t handle) {
    free_sharded_cross((ShardedCross *)handle);
    return swh_success_k;
}

// Every member fills its rows of the matrix on its own device and copies them straight to where the caller wants them:
// host memory (each device over its own PCIe link) or memory of the first device (peer copies). No collective: the row
// blocks are disjoint.
static swh_status_t cross_sharded(void *engine, int kind, swh_scope_t handle, swh_sharded_cross_t batch, void *matrix, size_t row_stride_bytes,
                                  const char **error) {
    Scope *scope = (Scope *)handle;
    ShardedCross *sc = (ShardedCross *)batch;
    if (!scope || !scope->multi || !sc || sc->scope != scope) return sharded_fail(error, swh_invalid_argument_k, "scope and sharded product do not belong together");
    if (!engine || ((const Engine *)engine)->kind != kind) return sharded_fail(error, swh_invalid_argument_k, "not an engine of this kind");
    if (!matrix && sc->rows && sc->columns) return sharded_fail(error, swh_invalid_argument_k, "null output pointer");
    if (kind != 0 && sc->utf8) return sharded_fail(error, swh_not_implemented_k, "substitution-matrix scoring over UTF-8 code points (the matrix is indexed by bytes)");
    MultiScope *multi = (MultiScope *)scope->multi;
    const size_t members = multi->members.size();
    if (!row_stride_bytes) row_stride_bytes = sc->columns * sizeof(uint64_t);
    std::vector<void *> engines(members, engine);
    if (kind != 0) {   // alignment engines: one clone per member (see alignment_sharded)
        const Engine *source = (const Engine *)engine;
        std::vector<void *> clones;
        const swh_status_t status = engine_clones_of(multi, source, clones, error);
        if (status != swh_success_k) return status;
        engines = clones;
    }
    auto stream_of = [&](size_t r) { return ((Scope *)multi->members[r])->stream; };
    hipError_t hip_error = hipSuccess;
    auto drain = [&]() {
        for (size_t q = 0; q < members; ++q) { (void)hipSetDevice(multi->devices[q]); (void)hipStreamSynchronize(stream_of(q)); swh_scope_synchronize(multi->members[q], nullptr); }
        (void)hipSetDevice(multi->devices[0]);
        (void)hipGetLastError();
    };
    for (size_t r = 0; r < members && hip_error == hipSuccess; ++r) {
        const size_t n = (size_t)(sc->cuts[r + 1] - sc->cuts[r]);
        hip_error = hipSetDevice(multi->devices[r]);
        if (hip_error == hipSuccess) hip_error = hipEventRecord(multi->begin[r], stream_of(r));
        if (hip_error != hipSuccess || !n || !sc->columns) { if (hip_error == hipSuccess) hip_error = hipEventRecord(multi->done[r], stream_of(r)); continue; }
        swh_prepared_view_t vq{sc->queries[r], 0, n}, vc{sc->candidates[r], 0, (size_t)sc->columns};
        swh_status_t status;
        if (kind == 0) status = swh_levenshtein_cross_prepared((swh_levenshtein_t)engines[r], multi->members[r], &vq, &vc, (size_t *)sc->blocks[r], sc->columns * 8, error);
        else if (kind == 1) status = swh_nw_cross_prepared((swh_nw_t)engines[r], multi->members[r], &vq, &vc, (ptrdiff_t *)sc->blocks[r], sc->columns * 8, error);
        else status = swh_sw_cross_prepared((swh_sw_t)engines[r], multi->members[r], &vq, &vc, (ptrdiff_t *)sc->blocks[r], sc->columns * 8, error);
        if (status != swh_success_k) { drain(); return status; }
        hip_error = hipEventRecord(multi->done[r], stream_of(r));
        if (hip_error == hipSuccess)
            hip_error = hipMemcpy2DAsync((char *)matrix + (size_t)sc->cuts[r] * row_stride_bytes, row_stride_bytes, sc->blocks[r], sc->columns * 8,
                                         sc->columns * 8, n, hipMemcpyDefault, stream_of(r));
    }
    for (size_t r = 0; r < members && hip_error == hipSuccess; ++r) {
        hip_error = hipSetDevice(multi->devices[r]);
        if (hip_error == hipSuccess) hip_error = hipStreamSynchronize(stream_of(r));
    }
    (void)hipSetDevice(multi->devices[0]);
    if (hip_error != hipSuccess) {
        drain();
        return sharded_fail(error, swh_device_error_k, "HIP error '%s' in the sharded cross-product", hipGetErrorString(hip_error));
    }
    for (size_t r = 0; r < members; ++r) {
        swh_status_t status = swh_scope_synchronize(multi->members[r], error);
        if (status != swh_success_k) { drain(); return status; }
    }
    swh_shard_timing_t timing{};
    for (size_t r = 0; r < members; ++r) {
        float ms = 0;
        (void)hipSetDevice(multi->devices[r]);
        if (hipEventElapsedTime(&ms, multi->begin[r], multi->done[r]) == hipSuccess && ms > timing.compute_ms) timing.compute_ms = ms;
    }
    (void)hipSetDevice(multi->devices[0]);
    (void)hipGetLastError();
    timing.cells = sc->cells;
    timing.pairs = sc->rows * sc->columns;
    multi->timing = timing;
    return swh_success_k;
}
swh_status_t swh_levenshtein_cross_sharded(swh_levenshtein_t engine, swh_scope_t scope, swh_sharded_cross_t batch, size_t *matrix,
                                           size_t row_stride_bytes, const char **error) {
    return cross_sharded(engine, 0, scope, batch, matrix, row_stride_bytes, error);
}
swh_status_t swh_nw_cross_sharded(swh_nw_t engine, swh_scope_t scope, swh_sharded_cross_t batch, ptrdiff_t *matrix, size_t row_stride_bytes,
                                  const char **error) {
    return cross_sharded(engine, 1, scope, batch, matrix, row_stride_bytes, error);
}
swh_status_t swh_sw_cross_sharded(swh_sw_t engine, swh_scope_t scope, swh_sharded_cross_t batch, ptrdiff_t *matrix, size_t row_stride_bytes,
                                  const char **error) {
    return cross_sharded(engine, 2, scope, batch, matrix, row_stride_bytes, error);
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
