// In-library histogram exchange for C callers that drive several GPUs from ONE process: a thin wrapper over RCCL's
// single-process communicator (ncclCommInitAll) with one int64 sum all-reduce.  The Python host path does the same
// exchange with torch.distributed (one process per GPU, backend "nccl" = RCCL); this file is for callers without it.
// RCCL is bound at run time (dlopen) so that the library loads, and every other entry point works, on hosts
// without librccl; with one device the all-reduce is the identity and needs no RCCL at all.
#include "../../include/isingdiff.h"
#include "idm_internal.h"

#include <dlfcn.h>

#include <string>
#include <vector>

namespace {

typedef void *nccl_comm_t;
typedef int (*fn_comm_init_all)(nccl_comm_t *, int, const int *);
typedef int (*fn_comm_destroy)(nccl_comm_t);
typedef int (*fn_all_reduce)(const void *, void *, size_t, int, int, nccl_comm_t, hipStream_t);
typedef int (*fn_group)(void);
typedef const char *(*fn_err)(int);
constexpr int kNcclInt64 = 4, kNcclSum = 0;   // ncclDataType_t / ncclRedOp_t values of rccl.h

struct Rccl {
    void *handle = nullptr;
    fn_comm_init_all init_all = nullptr;
    fn_comm_destroy destroy = nullptr;
    fn_all_reduce all_reduce = nullptr;
    fn_group group_start = nullptr, group_end = nullptr;
    fn_err err = nullptr;
};

Rccl *load_rccl(std::string *why)
{
    static Rccl r;
    static bool tried = false;
    if (!tried) {
        tried = true;
        for (const char *name : {"librccl.so.1", "librccl.so"}) {
            r.handle = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (r.handle) break;
        }
        if (r.handle) {
            r.init_all = (fn_comm_init_all)dlsym(r.handle, "ncclCommInitAll");
            r.destroy = (fn_comm_destroy)dlsym(r.handle, "ncclCommDestroy");
            r.all_reduce = (fn_all_reduce)dlsym(r.handle, "ncclAllReduce");
            r.group_start = (fn_group)dlsym(r.handle, "ncclGroupStart");
            r.group_end = (fn_group)dlsym(r.handle, "ncclGroupEnd");
            r.err = (fn_err)dlsym(r.handle, "ncclGetErrorString");
        }
    }
    if (!r.handle || !r.init_all || !r.destroy || !r.all_reduce || !r.group_start || !r.group_end) {
        if (why) *why = "librccl.so could not be loaded (needed for more than one device)";
        return nullptr;
    }
    return &r;
}

}  // namespace

struct idm_comm {
    std::vector<int> devices;
    std::vector<nccl_comm_t> comms;   // empty for a single device
};

namespace {

// Both multi-device calls walk the devices with hipSetDevice (ncclCommInitAll does so itself); the caller's current
// device is put back on every return path, so that its following idm_* and HIP calls land where they did before.
struct DeviceGuard {
    int dev = -1;
    DeviceGuard() { if (hipGetDevice(&dev) != hipSuccess) dev = -1; }
    ~DeviceGuard() { if (dev >= 0) (void)hipSetDevice(dev); }
};

}  // namespace

extern "C" {

int idm_comm_create(idm_comm **out, const int *devices, int n)
{
    if (!out || !devices || n <= 0) return idm_set_error(IDM_ERR_INVALID, "bad argument");
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess) return idm_set_error(IDM_ERR_HIP, "hipGetDeviceCount failed");
    for (int i = 0; i < n; ++i) {
        if (devices[i] < 0 || devices[i] >= ndev) return idm_set_error(IDM_ERR_INVALID, "no such device");
        for (int j = 0; j < i; ++j)
            if (devices[j] == devices[i]) return idm_set_error(IDM_ERR_INVALID, "a device is listed twice");
    }
    idm_comm *c = new idm_comm();
    c->devices.assign(devices, devices + n);
    if (n > 1) {
        DeviceGuard guard;
        std::string why;
        Rccl *r = load_rccl(&why);
        if (!r) { delete c; return idm_set_error(IDM_ERR_UNSUPPORTED, why.c_str()); }
        c->comms.resize(n);
        const int rc = r->init_all(c->comms.data(), n, devices);
        if (rc != 0) {
            const std::string msg = std::string("ncclCommInitAll: ") + (r->err ? r->err(rc) : "error");
            delete c;
            return idm_set_error(IDM_ERR_HIP, msg.c_str());
        }
    }
    *out = c;
    return IDM_OK;
}

int idm_allreduce_i64(idm_comm *c, int64_t *const *per_device_bufs, size_t count, void *const *streams)
{
    if (!c || !per_device_bufs || count == 0) return idm_set_error(IDM_ERR_INVALID, "bad argument");
    const int n = (int)c->devices.size();
    for (int i = 0; i < n; ++i)
        if (!per_device_bufs[i]) return idm_set_error(IDM_ERR_INVALID, "null buffer");
    if (n == 1) return IDM_OK;   // the sum over one rank
    Rccl *r = load_rccl(nullptr);
    if (!r) return idm_set_error(IDM_ERR_UNSUPPORTED, "librccl.so is not loaded");
    DeviceGuard guard;
    int rc = r->group_start();
    for (int i = 0; i < n && rc == 0; ++i) {
        if (hipSetDevice(c->devices[i]) != hipSuccess) { rc = -1; break; }
        rc = r->all_reduce(per_device_bufs[i], per_device_bufs[i], count, kNcclInt64, kNcclSum, c->comms[i],
                           streams ? (hipStream_t)streams[i] : nullptr);
    }
    const int rc_end = r->group_end();
    if (rc == 0) rc = rc_end;
    if (rc != 0) return idm_set_error(IDM_ERR_HIP, (std::string("ncclAllReduce: ") + (rc > 0 && r->err ? r->err(rc) : "error")).c_str());
    return IDM_OK;
}

int idm_comm_destroy(idm_comm *c)
{
    if (!c) return IDM_OK;
    if (!c->comms.empty()) {
        Rccl *r = load_rccl(nullptr);
        if (r)
            for (nccl_comm_t k : c->comms) (void)r->destroy(k);
    }
    delete c;
    return IDM_OK;
}

}  // extern "C"
