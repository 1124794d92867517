// viamd_amd/csrc/vmd_reduce.cpp — the multi-GPU merge behind the C ABI (include/vmd_eval.h "multi-GPU merge").
//
// VIAMD evaluates a script from C++ (/root/reference/src/main.cpp:993-1008: pool threads call md_script_eval_frame_range
// on disjoint frame ranges of ONE eval).  Across GPUs the same thing happens one process per GPU: every rank runs
// vmd_eval_frame_range on its block of frames, then ONE call - vmd_eval_reduce - sums the integer accumulators of all
// ranks in place on the device (RCCL all-reduce over xGMI), merges what lives on the host (fp64 normalisation weights,
// temporal rows, the frame mask) in one packed fp64 all-reduce, and re-derives the float views.  Integer sums make the
// merged result independent of the rank count.
//
// The collective is an interface (vmd_collective_i): vmd_comm_* is its RCCL implementation - librccl is loaded at run
// time (dlopen of the soname, so a host that already carries an RCCL, e.g. through PyTorch, shares that copy; a plain C++
// host gets /opt/rocm/lib/librccl.so.1) - and tests plug in a gloo-backed one on the CPU emulator build.
// This file only uses the public C ABI of the evaluator.
#include <hip/hip_runtime.h>

#include <dlfcn.h>

#include <cstdio>
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "vmd_eval.h"

extern "C" void vmd_set_last_error(const char* msg);

static bool red_fail(const std::string& msg) {
    vmd_set_last_error(msg.c_str());
    fprintf(stderr, "[viamd_amd] error: %s\n", msg.c_str());
    return false;
}

// ------------------------------------------------------------------------------------------------ RCCL, loaded at run time
// The handful of RCCL entry points and constants the merge needs (rccl.h: ncclDataType_t ncclUint64 = 5, ncclFloat64 = 8;
// ncclRedOp_t ncclSum = 0; ncclResult_t ncclSuccess = 0; ncclUniqueId = 128 opaque bytes passed by value).
namespace {
struct NcclUniqueId { char internal[VMD_COMM_ID_BYTES]; };
typedef void* NcclComm;
struct Rccl {
    void* handle = nullptr;
    int (*GetUniqueId)(NcclUniqueId*) = nullptr;
    int (*CommInitRank)(NcclComm*, int, NcclUniqueId, int) = nullptr;
    int (*CommDestroy)(NcclComm) = nullptr;
    int (*CommCount)(const NcclComm, int*) = nullptr;
    int (*CommUserRank)(const NcclComm, int*) = nullptr;
    int (*AllReduce)(const void*, void*, size_t, int, int, NcclComm, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    std::string error;
};
Rccl* rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        // a copy the process already carries first (PyTorch bundles its own RCCL: two different RCCL builds in one process each
        // want the devices for themselves), only then the system's
        for (const char* n : names) {
            r.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD);
            if (r.handle) break;
        }
        for (const char* n : names) {
            if (r.handle) break;
            r.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL);
        }
        if (!r.handle) {
            const char* why = dlerror();            // one call: dlerror() clears the message it returns
            r.error = std::string("cannot load librccl: ") + (why ? why : "?");
            return;
        }
        auto sym = [&](const char* s) { void* p = dlsym(r.handle, s); if (!p && r.error.empty()) r.error = std::string("librccl lacks ") + s; return p; };
        r.GetUniqueId = (int (*)(NcclUniqueId*))sym("ncclGetUniqueId");
        r.CommInitRank = (int (*)(NcclComm*, int, NcclUniqueId, int))sym("ncclCommInitRank");
        r.CommDestroy = (int (*)(NcclComm))sym("ncclCommDestroy");
        r.CommCount = (int (*)(const NcclComm, int*))sym("ncclCommCount");
        r.CommUserRank = (int (*)(const NcclComm, int*))sym("ncclCommUserRank");
        r.AllReduce = (int (*)(const void*, void*, size_t, int, int, NcclComm, hipStream_t))sym("ncclAllReduce");
        r.GroupStart = (int (*)())sym("ncclGroupStart");
        r.GroupEnd = (int (*)())sym("ncclGroupEnd");
        r.GetErrorString = (const char* (*)(int))sym("ncclGetErrorString");
    });
    return &r;
}
bool rccl_ok(Rccl* r, int rc, const char* what) {
    if (rc == 0) return true;
    return red_fail(std::string(what) + " failed: " + (r->GetErrorString ? r->GetErrorString(rc) : "RCCL error"));
}
}  // namespace

struct vmd_comm_t {
    NcclComm comm = nullptr;
    bool owned = false;
    int rank = 0, size = 1;
    vmd_collective_i iface;
};

static int comm_rank(void* inst) { return ((vmd_comm_t*)inst)->rank; }
static int comm_size(void* inst) { return ((vmd_comm_t*)inst)->size; }
static bool comm_allreduce(void* inst, void* buf, size_t n, int dtype, void* stream) {
    vmd_comm_t* c = (vmd_comm_t*)inst;
    Rccl* r = rccl();
    if (n == 0) return true;
    return rccl_ok(r, r->AllReduce(buf, buf, n, dtype, /*ncclSum*/ 0, c->comm, (hipStream_t)stream), "ncclAllReduce");
}
static bool comm_allreduce_u64(void* inst, uint64_t* buf, size_t n, void* stream) { return comm_allreduce(inst, buf, n, /*ncclUint64*/ 5, stream); }
static bool comm_allreduce_f64(void* inst, double* buf, size_t n, void* stream) { return comm_allreduce(inst, buf, n, /*ncclFloat64*/ 8, stream); }
static bool comm_allreduce_u32(void* inst, uint32_t* buf, size_t n, void* stream) { return comm_allreduce(inst, buf, n, /*ncclUint32*/ 3, stream); }
static bool comm_group_begin(void*) { Rccl* r = rccl(); return rccl_ok(r, r->GroupStart(), "ncclGroupStart"); }
static bool comm_group_end(void*) { Rccl* r = rccl(); return rccl_ok(r, r->GroupEnd(), "ncclGroupEnd"); }

static vmd_comm_t* comm_wrap(NcclComm comm, bool owned) {
    Rccl* r = rccl();
    vmd_comm_t* c = new vmd_comm_t();
    c->comm = comm; c->owned = owned;
    if (!rccl_ok(r, r->CommCount(comm, &c->size), "ncclCommCount") || !rccl_ok(r, r->CommUserRank(comm, &c->rank), "ncclCommUserRank")) {
        delete c;
        return nullptr;
    }
    c->iface.inst = c;
    c->iface.rank = comm_rank; c->iface.size = comm_size;
    c->iface.allreduce_sum_u64 = comm_allreduce_u64; c->iface.allreduce_sum_f64 = comm_allreduce_f64;
    c->iface.group_begin = comm_group_begin; c->iface.group_end = comm_group_end;
    c->iface.allreduce_sum_u32 = comm_allreduce_u32;
    return c;
}

extern "C" bool vmd_comm_unique_id(uint8_t id[VMD_COMM_ID_BYTES]) {
    Rccl* r = rccl();
    if (!r->error.empty()) return red_fail(r->error);
    NcclUniqueId u;
    if (!rccl_ok(r, r->GetUniqueId(&u), "ncclGetUniqueId")) return false;
    memcpy(id, u.internal, VMD_COMM_ID_BYTES);
    return true;
}

extern "C" vmd_comm_t* vmd_comm_create(int nranks, int rank, const uint8_t id[VMD_COMM_ID_BYTES]) {
    Rccl* r = rccl();
    if (!r->error.empty()) { red_fail(r->error); return nullptr; }
    if (nranks < 1 || rank < 0 || rank >= nranks || !id) { red_fail("vmd_comm_create: bad rank / size / id"); return nullptr; }
    NcclUniqueId u;
    memcpy(u.internal, id, VMD_COMM_ID_BYTES);
    NcclComm comm = nullptr;
    if (!rccl_ok(r, r->CommInitRank(&comm, nranks, u, rank), "ncclCommInitRank")) return nullptr;
    vmd_comm_t* c = comm_wrap(comm, true);
    if (!c) r->CommDestroy(comm);
    return c;
}

extern "C" vmd_comm_t* vmd_comm_from_nccl(void* nccl_comm) {
    Rccl* r = rccl();
    if (!r->error.empty()) { red_fail(r->error); return nullptr; }
    if (!nccl_comm) { red_fail("vmd_comm_from_nccl: NULL communicator"); return nullptr; }
    return comm_wrap((NcclComm)nccl_comm, false);
}

extern "C" void vmd_comm_destroy(vmd_comm_t* c) {
    if (!c) return;
    if (c->owned && c->comm) rccl()->CommDestroy(c->comm);
    delete c;
}

extern "C" const vmd_collective_i* vmd_comm_collective(vmd_comm_t* c) { return c ? &c->iface : nullptr; }
extern "C" int vmd_comm_rank(const vmd_comm_t* c) { return c ? c->rank : 0; }
extern "C" int vmd_comm_size(const vmd_comm_t* c) { return c ? c->size : 1; }

// ------------------------------------------------------------------------------------------------ the merge

extern "C" int vmd_eval_internal_device(const vmd_script_eval_t* eval);
extern "C" void vmd_eval_internal_lock(vmd_script_eval_t* eval, int lock);
extern "C" vmd_reduce_stats_t* vmd_eval_internal_reduce_stats(vmd_script_eval_t* eval);

namespace {
// u64 voxels <-> u32 for the trip over the links (only when the merged counts provably fit: vmd_accum_view_t::count_bound)
__global__ void k_narrow_u64(const uint64_t* __restrict__ src, uint32_t* __restrict__ dst, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = (uint32_t)src[i];
}
__global__ void k_widen_u32(const uint32_t* __restrict__ src, uint64_t* __restrict__ dst, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = (uint64_t)src[i];
}
// the largest count of a volume on THIS rank (block maximum through LDS, one atomicMax per block): what the measured bound of the narrowing is made of
__global__ void k_max_u64(const uint64_t* __restrict__ src, size_t n, unsigned long long* __restrict__ out) {
    __shared__ unsigned long long part[256];
    unsigned long long m = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) m = src[i] > m ? src[i] : m;
    part[threadIdx.x] = m;
    __syncthreads();
    for (unsigned o = blockDim.x / 2; o > 0; o >>= 1) {
        if (threadIdx.x < o && part[threadIdx.x + o] > part[threadIdx.x]) part[threadIdx.x] = part[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0 && part[0]) atomicMax(out, part[0]);
}
// staging of what travels: a small pool of buffers per device for the whole process (a merge per evaluation: no allocation in the steady
// state).  A merge LEASES a buffer for its duration and never waits for another merge: a host that drives several ranks from threads
// of one process - one GPU each, or several ranks on one GPU - would otherwise have rank A waiting in the collective for a rank B that
// waits for A's buffer (tests/native/reduce_threads.cpp runs exactly that).  (the buffer used to be thread_local -
// leaked per thread - and lived on whatever device the caller had current.)
struct Scratch { void* p = nullptr; size_t cap = 0; };
struct ScratchPool { std::mutex mtx; std::vector<Scratch> idle; };
ScratchPool g_scratch_pool[64];
struct ScratchLease {                            // declared after DeviceScope: returned (or freed) while the eval's device is still current
    int dev;
    Scratch sc;
    explicit ScratchLease(int d) : dev(d) {}
    bool take(size_t need) {
        {
            std::lock_guard<std::mutex> l(g_scratch_pool[dev].mtx);
            std::vector<Scratch>& idle = g_scratch_pool[dev].idle;
            // the smallest idle buffer that is large enough, else the largest one (it is regrown below)
            auto better = [need](const Scratch& a, const Scratch& b) {
                const bool fa = a.cap >= need, fb = b.cap >= need;
                if (fa != fb) return fa;
                return fa ? a.cap < b.cap : a.cap > b.cap;
            };
            size_t best = idle.size();
            for (size_t i = 0; i < idle.size(); ++i) if (best == idle.size() || better(idle[i], idle[best])) best = i;
            if (best < idle.size()) { sc = idle[best]; idle.erase(idle.begin() + (long)best); }
        }
        if (sc.cap >= need) return true;
        if (sc.p) (void)hipFree(sc.p);
        sc.p = nullptr; sc.cap = 0;
        if (hipMalloc(&sc.p, need) != hipSuccess) { (void)hipGetLastError(); sc.p = nullptr; return false; }
        sc.cap = need;
        return true;
    }
    ~ScratchLease() {
        if (!sc.p) return;
        {
            std::lock_guard<std::mutex> l(g_scratch_pool[dev].mtx);
            if (g_scratch_pool[dev].idle.size() < 4) { g_scratch_pool[dev].idle.push_back(sc); return; }
        }
        (void)hipFree(sc.p);
    }
};
// the caller's current device comes back on every exit path; the timing events are destroyed on every exit path
struct DeviceScope {
    int prev = -1;
    hipEvent_t t0 = nullptr, t1 = nullptr;
    DeviceScope() { if (hipGetDevice(&prev) != hipSuccess) { (void)hipGetLastError(); prev = -1; } }
    ~DeviceScope() {
        if (t0) (void)hipEventDestroy(t0);
        if (t1) (void)hipEventDestroy(t1);
        if (prev >= 0) (void)hipSetDevice(prev);
    }
};
struct EvalLock {
    vmd_script_eval_t* e;
    explicit EvalLock(vmd_script_eval_t* ev) : e(ev) { vmd_eval_internal_lock(e, 1); }
    ~EvalLock() { vmd_eval_internal_lock(e, 0); }
};
}  // namespace

extern "C" bool vmd_eval_reduce(vmd_script_eval_t* eval, const vmd_collective_i* coll, void* stream) {
    if (!eval) return red_fail("vmd_eval_reduce: eval is NULL");
    if (!coll || !coll->allreduce_sum_u64 || !coll->allreduce_sum_f64) return red_fail("vmd_eval_reduce: incomplete collective interface");
    if (!vmd_eval_wait_settled(eval)) return false;       // a deferred-settle eval (option readahead_lone): this rank's totals first
    DeviceScope scope;
    hipEvent_t& t0 = scope.t0;
    hipEvent_t& t1 = scope.t1;
    const int dev = vmd_eval_internal_device(eval);
    if (dev < 0 || dev >= 64 || hipSetDevice(dev) != hipSuccess) return red_fail("vmd_eval_reduce: cannot select the evaluator's device");
    hipStream_t s = (hipStream_t)stream;
    (void)hipEventCreate(&t0); (void)hipEventCreate(&t1);
    const int world = coll->size ? coll->size(coll->inst) : 1;
    if (t0) (void)hipEventRecord(t0, s);
    const size_t nviews = vmd_eval_accum_views(eval, nullptr, 0);
    std::vector<vmd_accum_view_t> views(nviews);
    vmd_eval_accum_views(eval, views.data(), nviews);
    const size_t F = vmd_eval_num_frames(eval);
    // what travels besides the u64 accumulators:
    //  * u32 copies of the volumes whose merged counts fit 32 bits (half the bytes of the one payload that has a size: 8.4 instead of
    //    16.8 MB per 128^3 volume);
    //  * everything that lives on the host as ONE fp64 buffer: normalisation weights, temporal rows (zero on the ranks that did not
    //    evaluate the frame; a float survives the trip through fp64 unchanged) and the frame mask (a frame is evaluated by one rank,
    //    so the sum is 0/1; > 0 also covers ranks that evaluated the same frame)
    // VIAMD_AMD_REDUCE_MEASURE=1 (tests): never trust the static bound, always measure - the path config 4 takes, on systems of test size
    const char* fm = getenv("VIAMD_AMD_REDUCE_MEASURE");
    const bool force_measure = fm && fm[0] == '1';
    size_t n_f64 = F, n_u32 = 0;
    std::vector<char> narrow(nviews, 0);          // 1 = the bound proves it, 2 = to be decided from the measured maxima (below)
    size_t n_measured = 0;
    for (size_t i = 0; i < nviews; ++i) {
        const vmd_accum_view_t& v = views[i];
        n_f64 += (v.weights64 ? v.num_weights : 0) + (v.temporal ? v.num_temporal : 0);
        if (!(coll->allreduce_sum_u32 && v.counts_dev && v.num_counts >= 65536 && world > 0)) continue;
        // count_bound assumes every frame index is merged once; ranks that evaluated the same indices (the bench's weak scaling does: every
        // rank its own trajectory) multiply a voxel by up to the rank count, so the STATIC proof needs bound x ranks < 2^32.  Every rank
        // computes the same bound from the same script and frame count, so all ranks decide alike.
        if (!force_measure && v.count_bound != 0 && v.count_bound <= 0xffffffffull / (uint64_t)world) narrow[i] = 1;
        // Where that fails (config 4: 10 000 frames x 7 structures x 33 000 targets = 2.3e9 - while no voxel of it ever holds more than a
        // few hundred hits) the ranks MEASURE: the sum over the ranks of each rank's largest voxel bounds every merged voxel whatever
        // the ranks evaluated.  The maxima ride in the packed fp64 all-reduce (a u64 below 2^53 survives it), which then has to leave
        // BEFORE the volumes: two collective launches instead of one, 8.4 instead of 16.8 MB per volume on the links.
        else { narrow[i] = 2; n_measured += 1; }
        n_u32 += (v.num_counts + 1) & ~(size_t)1;               // keep the fp64 part behind it 8-byte aligned (reserved also for undecided volumes)
    }
    n_f64 += n_measured;
    std::vector<double> packed(n_f64);
    {
        EvalLock lock(eval);                                         // the host-side arrays belong to the evaluator
        size_t off = 0;
        for (const vmd_accum_view_t& v : views) {
            if (v.weights64) { memcpy(&packed[off], v.weights64, v.num_weights * sizeof(double)); off += v.num_weights; }
            if (v.temporal) { for (size_t i = 0; i < v.num_temporal; ++i) packed[off + i] = (double)v.temporal[i]; off += v.num_temporal; }
        }
        const uint8_t* mask = vmd_eval_frame_mask(eval);
        for (size_t f = 0; f < F; ++f) packed[off + f] = mask[f] ? 1.0 : 0.0;
    }
    const size_t off_measured = n_f64 - n_measured;      // the last n_measured slots: this rank's largest count per undecided volume
    ScratchLease lease(dev);
    const size_t need = n_u32 * sizeof(uint32_t) + n_f64 * sizeof(double);
    if (!lease.take(need)) return red_fail("vmd_eval_reduce: hipMalloc failed");
    Scratch& sc = lease.sc;
    uint32_t* d_u32 = (uint32_t*)sc.p;
    double* d_packed = (double*)((char*)sc.p + n_u32 * sizeof(uint32_t));
    vmd_reduce_stats_t st;
    memset(&st, 0, sizeof(st));
    const bool grouped = coll->group_begin && coll->group_end;
    bool ok = true;
    bool packed_done = false;                    // the packed fp64 part has already travelled (measured bounds)
    if (n_measured) {
        // ---- first launch: this rank's largest count per undecided volume (block maxima, one u64 per volume, behind the u32 scratch is
        // not needed: they go through 8 bytes each at the end of the packed buffer), then the packed all-reduce, then the decision
        unsigned long long* d_max = (unsigned long long*)(d_packed + off_measured);         // the slots are overwritten with doubles below
        ok = hipMemsetAsync(d_max, 0, n_measured * sizeof(unsigned long long), s) == hipSuccess;
        size_t k = 0;
        for (size_t i = 0; i < nviews && ok; ++i) {
            if (narrow[i] != 2) continue;
            const size_t n = views[i].num_counts;
            hipLaunchKernelGGL(k_max_u64, dim3((unsigned)std::min<size_t>((n + 255) / 256, 2048)), dim3(256), 0, s, (const uint64_t*)views[i].counts_dev, n, d_max + k);
            ok = hipGetLastError() == hipSuccess;
            ++k;
        }
        std::vector<unsigned long long> h_max(n_measured, 0);
        ok = ok && hipMemcpyAsync(h_max.data(), d_max, n_measured * sizeof(unsigned long long), hipMemcpyDeviceToHost, s) == hipSuccess;
        ok = ok && hipStreamSynchronize(s) == hipSuccess;
        if (!ok) return red_fail("vmd_eval_reduce: measuring the volumes failed");
        for (size_t j = 0; j < n_measured; ++j) packed[off_measured + j] = h_max[j] < (1ull << 53) ? (double)h_max[j] : 1.0e19;      // beyond 2^53: never narrow
        ok = hipMemcpyAsync(d_packed, packed.data(), n_f64 * sizeof(double), hipMemcpyHostToDevice, s) == hipSuccess;
        ok = ok && coll->allreduce_sum_f64(coll->inst, d_packed, n_f64, s);
        st.bytes += n_f64 * sizeof(double); st.calls += 1;
        ok = ok && hipMemcpyAsync(packed.data(), d_packed, n_f64 * sizeof(double), hipMemcpyDeviceToHost, s) == hipSuccess;
        ok = ok && hipStreamSynchronize(s) == hipSuccess;
        if (!ok) return red_fail("vmd_eval_reduce: the first all-reduce failed");
        packed_done = true;
        k = 0;
        for (size_t i = 0; i < nviews; ++i) {
            if (narrow[i] != 2) continue;
            narrow[i] = packed[off_measured + k] <= 4294967295.0 ? 1 : 0;       // the same sum on every rank: the same decision on every rank
            ++k;
        }
    } else {
        ok = hipMemcpyAsync(d_packed, packed.data(), n_f64 * sizeof(double), hipMemcpyHostToDevice, s) == hipSuccess;
    }
    {
        size_t o = 0;
        for (size_t i = 0; i < nviews && ok; ++i) {
            if (!narrow[i]) { if (coll->allreduce_sum_u32 && views[i].counts_dev && views[i].num_counts >= 65536 && world > 0) o += (views[i].num_counts + 1) & ~(size_t)1; continue; }
            const size_t n = views[i].num_counts;
            hipLaunchKernelGGL(k_narrow_u64, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, (const uint64_t*)views[i].counts_dev, d_u32 + o, n);
            ok = hipGetLastError() == hipSuccess;
            o += (n + 1) & ~(size_t)1;
        }
    }
    if (!ok) return red_fail("vmd_eval_reduce: staging the merge failed");
    // ---- ONE collective launch for the accumulators: every all-reduce inside one group (north_star: "a single RCCL reduce"; the packed
    // fp64 part rides in it unless it had to go first with the measured bounds)
    if (grouped && !coll->group_begin(coll->inst)) return false;
    {
        size_t o = 0;
        for (size_t i = 0; i < nviews && ok; ++i) {
            const vmd_accum_view_t& v = views[i];
            if (!v.counts_dev || !v.num_counts) continue;
            const bool slot = coll->allreduce_sum_u32 && v.num_counts >= 65536 && world > 0;       // a stretch of the u32 scratch is reserved for it
            if (narrow[i]) {
                ok = coll->allreduce_sum_u32(coll->inst, d_u32 + o, v.num_counts, s);
                st.bytes += v.num_counts * sizeof(uint32_t); st.volumes_as_u32 += 1;
            } else {
                ok = coll->allreduce_sum_u64(coll->inst, v.counts_dev, v.num_counts, s);      // in place on the evaluator's accumulators
                st.bytes += v.num_counts * sizeof(uint64_t);
            }
            if (slot) o += (v.num_counts + 1) & ~(size_t)1;
            st.calls += 1;
        }
        if (!packed_done) {
            ok = ok && coll->allreduce_sum_f64(coll->inst, d_packed, n_f64, s);
            st.bytes += n_f64 * sizeof(double); st.calls += 1;
        }
    }
    if (grouped && !coll->group_end(coll->inst)) return false;
    st.grouped = grouped ? 1 : 0;
    if (!ok) return red_fail("vmd_eval_reduce: an all-reduce failed");
    {
        size_t o = 0;
        for (size_t i = 0; i < nviews && ok; ++i) {
            if (!narrow[i]) { if (coll->allreduce_sum_u32 && views[i].counts_dev && views[i].num_counts >= 65536 && world > 0) o += (views[i].num_counts + 1) & ~(size_t)1; continue; }
            const size_t n = views[i].num_counts;
            hipLaunchKernelGGL(k_widen_u32, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, (const uint32_t*)(d_u32 + o), views[i].counts_dev, n);
            ok = hipGetLastError() == hipSuccess;
            o += (n + 1) & ~(size_t)1;
        }
    }
    if (!packed_done) ok = ok && hipMemcpyAsync(packed.data(), d_packed, n_f64 * sizeof(double), hipMemcpyDeviceToHost, s) == hipSuccess;
    if (t1) (void)hipEventRecord(t1, s);
    ok = ok && hipStreamSynchronize(s) == hipSuccess;       // also: the in-place counts are final before finalize reads them
    if (!ok) return red_fail("vmd_eval_reduce: the merge failed on the device");
    std::vector<uint8_t> merged(F);
    {
        EvalLock lock(eval);
        size_t off = 0;
        for (const vmd_accum_view_t& v : views) {
            if (v.weights64) { memcpy(v.weights64, &packed[off], v.num_weights * sizeof(double)); off += v.num_weights; }
            if (v.temporal) { for (size_t i = 0; i < v.num_temporal; ++i) v.temporal[i] = (float)packed[off + i]; off += v.num_temporal; }
        }
        for (size_t f = 0; f < F; ++f) merged[f] = packed[off + f] > 0.5 ? 1 : 0;
    }
    vmd_eval_set_frame_mask(eval, merged.data(), F);
    float ms = 0.0f;
    if (t0 && t1 && hipEventElapsedTime(&ms, t0, t1) == hipSuccess) st.ms = ms;
    if (vmd_reduce_stats_t* dst = vmd_eval_internal_reduce_stats(eval)) *dst = st;
    return vmd_eval_finalize(eval);
}
