// comm.hip.h -- the hit-list collective behind the C ABI (vs_comm_*): RCCL called directly.
//
// north_star: "a batch of thousands of independent region queries shards trivially across the 8 GPUs of one node with an
// RCCL all-gatherv of hit lists over xGMI".  The reference has no counterpart (its loop over the regions is serial,
// src/commands.cc:145); the seam is the same one as the queries': C++ host code (the reference's query_main) calls plain
// C entry points.  Rounds 2-3 had the collective above the ABI, in Python (torch.distributed); a maintainer binding
// query_main got no multi-GPU path without it (missing #2).
//
// RCCL is loaded at run time (dlopen), not linked: a process that already carries an RCCL (torch's) shares that copy,
// one without it (the CLI) takes /opt/rocm's.  Records are padded to `max_count` per rank and travel in ONE ncclAllGather
// on the communicator's own stream behind an event -- beside the caller's next kernels when async.
#pragma once
#include <dlfcn.h>
#include <rccl/rccl.h>

struct RcclApi {
  void* lib = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  std::string path, error;
};
static std::string& rccl_error() { static std::string e; return e; }
static RcclApi* rccl_api() {
  static RcclApi api;
  static bool tried = false;
  if (tried) return api.lib ? &api : nullptr;
  tried = true;
  std::vector<std::string> names;
  const char* forced = getenv("VS_RCCL_LIB");   // an explicit library wins over everything, and nothing else is tried (tests: tests/native/fake_rccl.cpp)
  if (forced && *forced) {
    if (void* h = dlopen(forced, RTLD_NOW | RTLD_LOCAL)) { api.lib = h; api.path = forced; }
    else {
      const char* de = dlerror();
      api.error = std::string("cannot load VS_RCCL_LIB=") + forced + ": " + (de ? de : "not found");
      rccl_error() = api.error;
      return nullptr;
    }
  }
  if (!api.lib)
    for (const char* n : {"librccl.so", "librccl.so.1"}) {   // a copy the process already carries (torch's) first
      if (void* h = dlopen(n, RTLD_NOW | RTLD_NOLOAD)) { api.lib = h; api.path = n; break; }
    }
  if (!api.lib) {
    names.insert(names.end(), {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"});
    for (auto& n : names)
      if (void* h = dlopen(n.c_str(), RTLD_NOW | RTLD_LOCAL)) { api.lib = h; api.path = n; break; }
  }
  if (!api.lib) {
    const char* de = dlerror();
    api.error = std::string("cannot load RCCL (librccl.so): ") + (de ? de : "not found");
    rccl_error() = api.error;
    return nullptr;
  }
  api.GetUniqueId = (decltype(api.GetUniqueId))dlsym(api.lib, "ncclGetUniqueId");
  api.CommInitRank = (decltype(api.CommInitRank))dlsym(api.lib, "ncclCommInitRank");
  api.AllGather = (decltype(api.AllGather))dlsym(api.lib, "ncclAllGather");
  api.CommDestroy = (decltype(api.CommDestroy))dlsym(api.lib, "ncclCommDestroy");
  api.CommCount = (decltype(api.CommCount))dlsym(api.lib, "ncclCommCount");
  api.GetErrorString = (decltype(api.GetErrorString))dlsym(api.lib, "ncclGetErrorString");
  if (!api.GetUniqueId || !api.CommInitRank || !api.AllGather || !api.CommDestroy || !api.CommCount || !api.GetErrorString) {
    api.error = "RCCL at " + api.path + " lacks a symbol the collective needs";
    rccl_error() = api.error;
    api.lib = nullptr;
    return nullptr;
  }
  return &api;
}
static int rccl_missing() {   // (rccl_api() has been tried: its error says why)
  return fail(VS_ERR_UNSUPPORTED, "RCCL is not available in this process: %s", rccl_error().c_str());
}
#define RCCL_TRY(api, expr)                                                                                   \
  do {                                                                                                        \
    ncclResult_t r_ = (expr);                                                                                 \
    if (r_ != ncclSuccess) return fail(VS_ERR_HIP, "%s failed: %s", #expr, (api)->GetErrorString(r_));      \
  } while (0)

struct vs_comm {
  vs_index* idx = nullptr;
  int rank = 0, world = 1;
  ncclComm_t comm = nullptr;
  hipStream_t stream = nullptr;     // the collective's own stream: an async all-gather runs beside the index handle's kernels
  hipEvent_t packed = nullptr, done = nullptr;
  void* send = nullptr;             // this rank's padded records
  size_t send_cap = 0;
  void* recv = nullptr;             // receive buffer of the host-destination form
  size_t recv_cap = 0;
  bool pending = false;
  bool counted = false;             // in vs_index::live_comms (the handle's close is deferred until the communicator is gone)
};

extern "C" {

int vs_comm_unique_id(void* id_out) {
  if (!id_out) return fail(VS_ERR_ARG, "null argument");
  static_assert(sizeof(ncclUniqueId) == VS_COMM_ID_BYTES, "unique id size of the ABI");
  RcclApi* api = rccl_api();
  if (!api) return rccl_missing();
  ncclUniqueId id;
  RCCL_TRY(api, api->GetUniqueId(&id));
  memcpy(id_out, &id, sizeof(id));
  return VS_OK;
}

int vs_comm_init(vs_index* idx, int rank, int world, const void* id, vs_comm** out) {
  if (!idx || !id || !out) return fail(VS_ERR_ARG, "null argument");
  if (world < 1 || rank < 0 || rank >= world) return fail(VS_ERR_ARG, "rank %d of %d", rank, world);
  if (idx->device < 0) return fail(VS_ERR_NO_DEVICE, "index handle was opened without a device");
  RcclApi* api = rccl_api();
  if (!api) return rccl_missing();
  HIP_TRY(hipSetDevice(idx->device));
  vs_comm* c = new vs_comm();
  c->idx = idx; c->rank = rank; c->world = world;
  ncclUniqueId nid;
  memcpy(&nid, id, sizeof(nid));
  ncclResult_t rc = api->CommInitRank(&c->comm, world, nid, rank);
  if (rc != ncclSuccess) { delete c; return fail(VS_ERR_HIP, "ncclCommInitRank failed: %s", api->GetErrorString(rc)); }
  if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&c->packed, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&c->done, hipEventDisableTiming) != hipSuccess) {
    vs_comm_destroy(c);
    return fail(VS_ERR_HIP, "stream / event creation for the collective failed");
  }
  idx->live_comms++;
  c->counted = true;
  *out = c;
  return VS_OK;
}

int vs_comm_wait(vs_comm* c) {
  if (!c) return fail(VS_ERR_ARG, "null argument");
  if (!c->pending) return VS_OK;
  c->pending = false;
  HIP_TRY(hipSetDevice(c->idx->device));
  HIP_TRY(hipEventSynchronize(c->done));
  return VS_OK;
}

int vs_comm_info(vs_comm* c, int* rank, int* world, int* rccl_ranks) {
  if (!c) return fail(VS_ERR_ARG, "null argument");
  if (rank) *rank = c->rank;
  if (world) *world = c->world;
  if (rccl_ranks) {
    RcclApi* api = rccl_api();
    if (!api) return rccl_missing();
    int n = 0;
    RCCL_TRY(api, api->CommCount(c->comm, &n));
    *rccl_ranks = n;
  }
  return VS_OK;
}

int vs_comm_allgather_regions(vs_comm* c, vs_result* r, uint64_t region_base, uint64_t max_count, void* device_dst, int async_op) {
  if (!c || !r || !device_dst) return fail(VS_ERR_ARG, "null argument");
  if (r->idx != c->idx) return fail(VS_ERR_ARG, "the result belongs to another index handle than the communicator");
  if (r->d.Q > max_count) return fail(VS_ERR_ARG, "this rank holds %llu regions, max_count is %llu", (unsigned long long)r->d.Q, (unsigned long long)max_count);
  RcclApi* api = rccl_api();
  if (!api) return rccl_missing();
  vs_index* idx = c->idx;
  HIP_TRY(hipSetDevice(idx->device));
  VS_TRY(vs_comm_wait(c));   // (the send buffer is reused)
  const size_t bytes = std::max<uint64_t>(max_count, 1) * 32;
  if (c->send_cap < bytes) {
    if (c->send) (void)hipFree(c->send);
    c->send = nullptr; c->send_cap = 0;
    HIP_TRY(hipMalloc(&c->send, bytes));
    c->send_cap = bytes;
  }
  // this rank's records (rows beyond its count are never read: the counts delimit them), on the handle's stream ...
  if (r->d.Q) {
    VS_TRY(launch_pack_regions(r, (uint64_t*)c->send, region_base));
    // the pack kernel reads the result's per-region arrays: the result's completion event moves behind it, so that a
    // vs_result_free before vs_comm_wait does not hand those arrays back to the pool under the kernel
    VS_TRY(pooled_event(idx, &r->ev_done));
    HIP_TRY(hipEventRecord(r->ev_done, idx->stream));
    r->pending = true;
  }
  HIP_TRY(hipEventRecord(c->packed, idx->stream));
  // ... and ONE all-gather of the padded records on the communicator's stream behind them
  HIP_TRY(hipStreamWaitEvent(c->stream, c->packed, 0));
  RCCL_TRY(api, api->AllGather(c->send, device_dst, (size_t)max_count * 4, ncclUint64, c->comm, c->stream));
  HIP_TRY(hipEventRecord(c->done, c->stream));
  c->pending = true;
  if (!async_op) VS_TRY(vs_comm_wait(c));
  return VS_OK;
}

int vs_comm_allgather_regions_host(vs_comm* c, vs_result* r, uint64_t region_base, uint64_t max_count, void* host_dst) {
  if (!c || !host_dst) return fail(VS_ERR_ARG, "null argument");
  HIP_TRY(hipSetDevice(c->idx->device));
  VS_TRY(vs_comm_wait(c));
  const size_t bytes = (size_t)c->world * std::max<uint64_t>(max_count, 1) * 32;
  if (c->recv_cap < bytes) {
    if (c->recv) (void)hipFree(c->recv);
    c->recv = nullptr; c->recv_cap = 0;
    HIP_TRY(hipMalloc(&c->recv, bytes));
    c->recv_cap = bytes;
  }
  VS_TRY(vs_comm_allgather_regions(c, r, region_base, max_count, c->recv, 0));
  HIP_TRY(hipMemcpy(host_dst, c->recv, (size_t)c->world * max_count * 32, hipMemcpyDeviceToHost));
  return VS_OK;
}

void vs_comm_destroy(vs_comm* c) {
  if (!c) return;
  if (c->idx && c->idx->device >= 0) (void)hipSetDevice(c->idx->device);
  if (c->pending && c->done) (void)hipEventSynchronize(c->done);
  RcclApi* api = rccl_api();
  if (c->comm && api) (void)api->CommDestroy(c->comm);
  if (c->send) (void)hipFree(c->send);
  if (c->recv) (void)hipFree(c->recv);
  if (c->packed) (void)hipEventDestroy(c->packed);
  if (c->done) (void)hipEventDestroy(c->done);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  vs_index* idx = c->counted ? c->idx : nullptr;
  delete c;
  if (idx) {
    idx->live_comms--;
    if (idx->close_pending && idx->live_results == 0 && idx->live_comms == 0) vs_index_close(idx);
  }
}

}  // extern "C"
