// comm.hip -- the ONE exchange step of the multi-GPU path: a final gather of the result blocks (u0, status) over
// RCCL / xGMI, behind the C ABI and without PyTorch.  Instances are independent, so there is no data-path collective; every GPU
// owns a contiguous block of the global batch (shard bounds are the caller's: ihm2_amd/dist.py::shard_bounds).
//   * single process, one handle per device:  ihm2mpc_group_create (ncclCommInitAll) / _allgather_results / _free
//   * one process per GPU (torch.distributed.run launches bench.py that way): ihm2mpc_comm_unique_id on rank 0, the 128 bytes travel
//     by any side channel (ihm2_amd/dist.py: a TCP socket on MASTER_ADDR), ihm2mpc_comm_init on every rank, then
//     ihm2mpc_comm_allgather_results / ihm2mpc_comm_allreduce_max (the bench's max-over-ranks timing) / ihm2mpc_comm_free.
// Blocks may differ in size by construction of the block split: they are padded to the largest block for ncclAllGather (payload:
// 20 bytes per instance) and trimmed on the host.
// RCCL is loaded on first use (dlopen): a single-GPU user of libihm2mpc.so neither loads librccl.so nor depends on its presence.
#include <rccl/rccl.h>      // types and prototypes only: the entry points are resolved at run time

#include <dlfcn.h>

#include <algorithm>
#include <cstring>
#include <mutex>

#include <string>
#include <vector>

#include "ihm2mpc_internal.h"

extern int ihm2_fail(const char *fmt, ...);

namespace {
struct RcclApi {
    decltype(&ncclGetUniqueId) GetUniqueId;
    decltype(&ncclCommInitRank) CommInitRank;
    decltype(&ncclCommInitAll) CommInitAll;
    decltype(&ncclCommDestroy) CommDestroy;
    decltype(&ncclCommCount) CommCount;
    decltype(&ncclAllGather) AllGather;
    decltype(&ncclAllReduce) AllReduce;
    decltype(&ncclGroupStart) GroupStart;
    decltype(&ncclGroupEnd) GroupEnd;
    decltype(&ncclGetErrorString) GetErrorString;
    bool ok = false;
};
RcclApi rccl;
std::once_flag rccl_once;
std::string rccl_err;

// Resolves the entry points once per process (handles may be created from several threads).  A librccl.so the process has already mapped --
// torch ships its own -- is reused (RTLD_NOLOAD) instead of mapping a second, different RCCL beside it; whatever is opened here keeps its
// symbols to itself (RTLD_LOCAL).
void rccl_resolve()
{
    void *lib = nullptr;
    for (const char *name : {"librccl.so", "librccl.so.1"}) {
        lib = dlopen(name, RTLD_NOW | RTLD_NOLOAD | RTLD_LOCAL);
        if (lib) break;
    }
    if (!lib)
        for (const char *name : {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"}) {
            lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (lib) break;
        }
    if (!lib) { const char *e = dlerror(); rccl_err = std::string("RCCL is not available: ") + (e ? e : "dlopen failed"); return; }
#define RCCL_SYM(field, sym)                                                             \
    rccl.field = reinterpret_cast<decltype(rccl.field)>(dlsym(lib, #sym));                \
    if (!rccl.field) { rccl_err = "librccl.so has no symbol " #sym; return; }
    RCCL_SYM(GetUniqueId, ncclGetUniqueId) RCCL_SYM(CommInitRank, ncclCommInitRank) RCCL_SYM(CommInitAll, ncclCommInitAll)
    RCCL_SYM(CommDestroy, ncclCommDestroy) RCCL_SYM(CommCount, ncclCommCount) RCCL_SYM(AllGather, ncclAllGather) RCCL_SYM(AllReduce, ncclAllReduce)
    RCCL_SYM(GroupStart, ncclGroupStart) RCCL_SYM(GroupEnd, ncclGroupEnd) RCCL_SYM(GetErrorString, ncclGetErrorString)
#undef RCCL_SYM
    rccl.ok = true;
}

// 0 = the entry points are there; -1 = librccl.so or one of its symbols is missing (reported through ihm2mpc_last_error)
int rccl_load()
{
    std::call_once(rccl_once, rccl_resolve);
    return rccl.ok ? 0 : ihm2_fail("%s", rccl_err.c_str());
}
}  // namespace

#define NCCL_TRY(call)                                                                                            \
    do {                                                                                                          \
        ncclResult_t r_ = (call);                                                                                 \
        if (r_ != ncclSuccess) return ihm2_fail("%s failed: %s (%s:%d)", #call, rccl.GetErrorString(r_), __FILE__, __LINE__); \
    } while (0)
#define HIPC_TRY(call)                                                                                            \
    do {                                                                                                          \
        hipError_t e_ = (call);                                                                                   \
        if (e_ != hipSuccess) return ihm2_fail("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

struct ihm2mpc_group {
    int n;
    std::vector<int> dev;                       // device of every handle (the handles may be freed before the group)
    std::vector<ihm2mpc_handle *> h;
    std::vector<ncclComm_t> comm;
    std::vector<double *> send_u, recv_u;       // per device: (Bmax, 2), (n, Bmax, 2)
    std::vector<int32_t *> send_s, recv_s;
    int Bmax;
};

struct ihm2mpc_comm {
    ncclComm_t comm;
    int world, rank, Bmax;
    double *send_u, *recv_u, *red;
    int32_t *send_s, *recv_s;
    long long *ids;                             // (1 + world) device identities of ihm2mpc_comm_info, allocated on first use
    std::vector<int> sizes;
};

static int alloc_bufs(int n, int Bmax, double **su, double **ru, int32_t **ss, int32_t **rs)
{
    HIPC_TRY(hipMalloc((void **)su, (size_t)Bmax * 2 * sizeof(double)));
    HIPC_TRY(hipMalloc((void **)ru, (size_t)n * Bmax * 2 * sizeof(double)));
    HIPC_TRY(hipMalloc((void **)ss, (size_t)Bmax * sizeof(int32_t)));
    HIPC_TRY(hipMalloc((void **)rs, (size_t)n * Bmax * sizeof(int32_t)));
    HIPC_TRY(hipMemset(*su, 0, (size_t)Bmax * 2 * sizeof(double)));
    HIPC_TRY(hipMemset(*ss, 0, (size_t)Bmax * sizeof(int32_t)));
    return 0;
}

extern "C" {

int ihm2mpc_group_free(ihm2mpc_group *g);

int ihm2mpc_group_create(ihm2mpc_handle *const *handles, int32_t n, ihm2mpc_group **out)
{
    if (!handles || !out || n < 1) return ihm2_fail("null argument or n < 1");
    if (rccl_load()) return -1;
    std::vector<int> devs(n);
    int Bmax = 0;
    for (int i = 0; i < n; i++) {
        if (!handles[i]) return ihm2_fail("null handle %d", i);
        devs[i] = handles[i]->cfg.device;
        for (int j = 0; j < i; j++) if (devs[j] == devs[i]) return ihm2_fail("handles %d and %d live on the same device %d: one handle per device", j, i, devs[i]);
        Bmax = std::max(Bmax, handles[i]->B);
    }
    ihm2mpc_group *g = new ihm2mpc_group();
    g->n = n; g->Bmax = Bmax; g->dev = devs;
    g->h.assign(handles, handles + n);
    g->comm.assign(n, nullptr); g->send_u.assign(n, nullptr); g->recv_u.assign(n, nullptr); g->send_s.assign(n, nullptr); g->recv_s.assign(n, nullptr);
    // every failure from here on releases what the group holds so far (ihm2mpc_group_free skips what is still null)
    ncclResult_t r = rccl.CommInitAll(g->comm.data(), n, devs.data());
    if (r != ncclSuccess) {
        g->comm.assign(n, nullptr);
        (void)ihm2mpc_group_free(g);
        return ihm2_fail("ncclCommInitAll failed: %s", rccl.GetErrorString(r));
    }
    for (int i = 0; i < n; i++) {
        if (hipSetDevice(devs[i]) != hipSuccess || alloc_bufs(n, Bmax, &g->send_u[i], &g->recv_u[i], &g->send_s[i], &g->recv_s[i])) {
            (void)ihm2mpc_group_free(g);
            return ihm2_fail("device buffers of the gather could not be allocated on device %d", devs[i]);
        }
    }
    *out = g;
    return 0;
}

// u0_all (sum of the B's, 2) and status_all (sum of the B's), host, in handle order; taken from device 0's receive buffers
int ihm2mpc_group_allgather_results(ihm2mpc_group *g, double *u0_all, int32_t *status_all)
{
    if (!g || !u0_all || !status_all) return ihm2_fail("null argument");
    for (int i = 0; i < g->n; i++) {
        ihm2mpc_handle *h = g->h[i];
        HIPC_TRY(hipSetDevice(g->dev[i]));
        HIPC_TRY(hipMemcpyAsync(g->send_u[i], h->u0, (size_t)h->B * 2 * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
        HIPC_TRY(hipMemcpyAsync(g->send_s[i], h->status, (size_t)h->B * sizeof(int32_t), hipMemcpyDeviceToDevice, h->stream));
    }
    NCCL_TRY(rccl.GroupStart());
    for (int i = 0; i < g->n; i++) {
        NCCL_TRY(rccl.AllGather(g->send_u[i], g->recv_u[i], (size_t)g->Bmax * 2, ncclDouble, g->comm[i], g->h[i]->stream));
        NCCL_TRY(rccl.AllGather(g->send_s[i], g->recv_s[i], (size_t)g->Bmax, ncclInt32, g->comm[i], g->h[i]->stream));
    }
    NCCL_TRY(rccl.GroupEnd());
    for (int i = 0; i < g->n; i++) {
        HIPC_TRY(hipSetDevice(g->dev[i]));
        HIPC_TRY(hipStreamSynchronize(g->h[i]->stream));
    }
    HIPC_TRY(hipSetDevice(g->dev[0]));
    size_t off = 0;
    for (int i = 0; i < g->n; i++) {
        const size_t B = g->h[i]->B;
        HIPC_TRY(hipMemcpy(u0_all + off * 2, g->recv_u[0] + (size_t)i * g->Bmax * 2, B * 2 * sizeof(double), hipMemcpyDeviceToHost));
        HIPC_TRY(hipMemcpy(status_all + off, g->recv_s[0] + (size_t)i * g->Bmax, B * sizeof(int32_t), hipMemcpyDeviceToHost));
        off += B;
    }
    return 0;
}

int ihm2mpc_group_free(ihm2mpc_group *g)
{
    if (!g) return 0;
    for (int i = 0; i < g->n; i++) {
        (void)hipSetDevice(g->dev[i]);
        for (void *p : {(void *)g->send_u[i], (void *)g->recv_u[i], (void *)g->send_s[i], (void *)g->recv_s[i]}) if (p) (void)hipFree(p);
        if (g->comm[i]) (void)rccl.CommDestroy(g->comm[i]);
    }
    delete g;
    return 0;
}

// ---- one process per GPU ----
int ihm2mpc_comm_unique_id(uint8_t *id128)
{
    if (!id128) return ihm2_fail("null argument");
    if (rccl_load()) return -1;
    static_assert(sizeof(ncclUniqueId) == 128, "the id travels as 128 bytes");
    ncclUniqueId id;
    NCCL_TRY(rccl.GetUniqueId(&id));
    std::memcpy(id128, &id, sizeof id);
    return 0;
}

int ihm2mpc_comm_free(ihm2mpc_handle *h);

// sizes (world): instances of every rank's block (this rank's must equal the handle's batch)
int ihm2mpc_comm_init(ihm2mpc_handle *h, int32_t world, int32_t rank, const uint8_t *id128, const int32_t *sizes)
{
    if (!h || !id128 || !sizes) return ihm2_fail("null argument");
    if (world < 1 || rank < 0 || rank >= world) return ihm2_fail("rank %d out of range for world size %d", rank, world);
    if (h->comm) return ihm2_fail("the handle already has a communicator");
    if (sizes[rank] != h->B) return ihm2_fail("sizes[%d] = %d does not match the handle's batch %d", rank, sizes[rank], h->B);
    if (rccl_load()) return -1;
    HIPC_TRY(hipSetDevice(h->cfg.device));
    ihm2mpc_comm *c = new ihm2mpc_comm();
    c->world = world; c->rank = rank; c->Bmax = 0; c->comm = nullptr;
    c->send_u = c->recv_u = c->red = nullptr; c->send_s = c->recv_s = nullptr; c->ids = nullptr;
    c->sizes.assign(sizes, sizes + world);
    for (int r = 0; r < world; r++) c->Bmax = std::max(c->Bmax, (int)sizes[r]);
    ncclUniqueId id;
    std::memcpy(&id, id128, sizeof id);
    h->comm = c;        // from here on ihm2mpc_comm_free releases whatever has been set up
    ncclResult_t r = rccl.CommInitRank(&c->comm, world, id, rank);
    if (r != ncclSuccess) {
        c->comm = nullptr;
        (void)ihm2mpc_comm_free(h);
        return ihm2_fail("ncclCommInitRank failed: %s", rccl.GetErrorString(r));
    }
    if (alloc_bufs(world, c->Bmax, &c->send_u, &c->recv_u, &c->send_s, &c->recv_s) || hipMalloc((void **)&c->red, 2 * sizeof(double)) != hipSuccess) {
        (void)ihm2mpc_comm_free(h);
        return ihm2_fail("device buffers of the gather could not be allocated");
    }
    return 0;
}

// u0_all (sum of sizes, 2), status_all (sum of sizes): host, in rank order, on every rank
int ihm2mpc_comm_allgather_results(ihm2mpc_handle *h, double *u0_all, int32_t *status_all)
{
    if (!h || !h->comm || !u0_all || !status_all) return ihm2_fail("null argument or no communicator (ihm2mpc_comm_init)");
    ihm2mpc_comm *c = h->comm;
    HIPC_TRY(hipSetDevice(h->cfg.device));
    HIPC_TRY(hipMemcpyAsync(c->send_u, h->u0, (size_t)h->B * 2 * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
    HIPC_TRY(hipMemcpyAsync(c->send_s, h->status, (size_t)h->B * sizeof(int32_t), hipMemcpyDeviceToDevice, h->stream));
    NCCL_TRY(rccl.GroupStart());
    NCCL_TRY(rccl.AllGather(c->send_u, c->recv_u, (size_t)c->Bmax * 2, ncclDouble, c->comm, h->stream));
    NCCL_TRY(rccl.AllGather(c->send_s, c->recv_s, (size_t)c->Bmax, ncclInt32, c->comm, h->stream));
    NCCL_TRY(rccl.GroupEnd());
    HIPC_TRY(hipStreamSynchronize(h->stream));
    size_t off = 0;
    for (int r = 0; r < c->world; r++) {
        const size_t B = c->sizes[r];
        HIPC_TRY(hipMemcpy(u0_all + off * 2, c->recv_u + (size_t)r * c->Bmax * 2, B * 2 * sizeof(double), hipMemcpyDeviceToHost));
        HIPC_TRY(hipMemcpy(status_all + off, c->recv_s + (size_t)r * c->Bmax, B * sizeof(int32_t), hipMemcpyDeviceToHost));
        off += B;
    }
    return 0;
}

// *value <- max over the ranks of *value; doubles as the barrier of the timed region
int ihm2mpc_comm_allreduce_max(ihm2mpc_handle *h, double *value)
{
    if (!h || !h->comm || !value) return ihm2_fail("null argument or no communicator (ihm2mpc_comm_init)");
    ihm2mpc_comm *c = h->comm;
    HIPC_TRY(hipSetDevice(h->cfg.device));
    HIPC_TRY(hipMemcpyAsync(c->red, value, sizeof(double), hipMemcpyHostToDevice, h->stream));
    NCCL_TRY(rccl.AllReduce(c->red, c->red + 1, 1, ncclDouble, ncclMax, c->comm, h->stream));
    HIPC_TRY(hipMemcpyAsync(value, c->red + 1, sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIPC_TRY(hipStreamSynchronize(h->stream));
    return 0;
}

// What the communicator actually spans, for the record of a multi-GPU run: *count <- ncclCommCount, device_ids (world) <- the PCI identity
// (domain << 24 | bus << 8 | device, a tag of the device's UUID above bit 48) of the device every rank computes on, gathered over the communicator itself -- N ranks on fewer
// than N devices show up as repeated identities.
int ihm2mpc_comm_info(ihm2mpc_handle *h, int32_t *count, int64_t *device_ids)
{
    if (!h || !h->comm || !count || !device_ids) return ihm2_fail("null argument or no communicator (ihm2mpc_comm_init)");
    ihm2mpc_comm *c = h->comm;
    HIPC_TRY(hipSetDevice(h->cfg.device));
    int n = 0;
    NCCL_TRY(rccl.CommCount(c->comm, &n));
    *count = n;
    int dom = 0, bus = 0, dev = 0;
    HIPC_TRY(hipDeviceGetAttribute(&dom, hipDeviceAttributePciDomainID, h->cfg.device));
    HIPC_TRY(hipDeviceGetAttribute(&bus, hipDeviceAttributePciBusId, h->cfg.device));
    HIPC_TRY(hipDeviceGetAttribute(&dev, hipDeviceAttributePciDeviceId, h->cfg.device));
    long long mine = ((long long)dom << 24) | ((long long)bus << 8) | (long long)dev;
    // partitions of one package share (domain, bus, device): fifteen bits of a hash of the device's UUID ride in bits 48..62, so that two logical
    // devices are told apart and two ranks on ONE logical device still are not (no UUID: tag 0, the PCI identity alone decides)
    hipUUID uu;
    if (hipDeviceGetUuid(&uu, h->cfg.device) == hipSuccess) {
        unsigned long long f = 1469598103934665603ull;
        for (unsigned char b : uu.bytes) f = (f ^ b) * 1099511628211ull;
        mine |= (long long)((f ^ (f >> 15) ^ (f >> 30) ^ (f >> 45)) & 0x7FFFull) << 48;
    } else (void)hipGetLastError();
    if (!c->ids) HIPC_TRY(hipMalloc((void **)&c->ids, (size_t)(1 + c->world) * sizeof(long long)));
    HIPC_TRY(hipMemcpyAsync(c->ids, &mine, sizeof mine, hipMemcpyHostToDevice, h->stream));
    NCCL_TRY(rccl.AllGather(c->ids, c->ids + 1, 1, ncclInt64, c->comm, h->stream));
    std::vector<long long> all(c->world);
    HIPC_TRY(hipMemcpyAsync(all.data(), c->ids + 1, (size_t)c->world * sizeof(long long), hipMemcpyDeviceToHost, h->stream));
    HIPC_TRY(hipStreamSynchronize(h->stream));
    for (int r = 0; r < c->world; r++) device_ids[r] = all[r];
    return 0;
}

int ihm2mpc_comm_free(ihm2mpc_handle *h)
{
    if (!h || !h->comm) return 0;
    ihm2mpc_comm *c = h->comm;
    (void)hipSetDevice(h->cfg.device);
    (void)hipStreamSynchronize(h->stream);
    for (void *p : {(void *)c->send_u, (void *)c->recv_u, (void *)c->send_s, (void *)c->recv_s, (void *)c->red, (void *)c->ids}) if (p) (void)hipFree(p);
    if (c->comm) (void)rccl.CommDestroy(c->comm);
    delete c;
    h->comm = nullptr;
    return 0;
}

}  // extern "C"
