// Data-parallel gradient exchange behind the C ABI: ug_comm_{unique_id, init, allreduce_bucket, wait, destroy} over RCCL.
//
// replaces: the DistributedDataParallel reducer that accelerator.prepare / accelerator.backward put behind the reference's
// step (training/train.py:492,775) for hosts that do not run torch.distributed.  One ug_comm per rank
// (one process per GPU).  It owns ONE side HIP stream and the bf16 staging buffers; a bucket of the flat fp32 gradient buffer is
// handed over with the event after which it is final, and everything -- pack, collective(s), unpack -- runs on the side
// stream while the caller's stream continues with backward.  ug_comm_wait makes a consumer stream wait for every bucket issued
// so far.  Nothing synchronises the host.
//
// RCCL is resolved at run time (dlopen of the copy already mapped into the process -- PyTorch ships its own librccl.so.1 -- or
// of UNIGEN_RCCL_LIB / librccl.so.1 on the loader path), so the library itself has no link-time dependency on it and loads on
// hosts without RCCL; only ug_comm_init needs it.
//
// Wire formats (mode):
//   UG_COMM_FP32          ncclAllReduce(fp32, AVG) on the bucket itself: DDP's arithmetic, bit for bit
//   UG_COMM_BF16_FP32ACC  bf16 on the links, fp32 arithmetic: pack bf16(g); all-to-all of the world slices (grouped
//                         ncclSend / ncclRecv); rank-ordered fp32 sum of the world copies of this rank's slice, x 1 / world,
//                         one bf16 rounding (ug_grad_sum_shards_bf16); ncclAllGather; unpack
//   UG_COMM_BF16          pack bf16(g / world); ncclAllReduce(bf16, SUM); unpack  (world - 1 extra roundings on a ring)
//   UG_COMM_FP32_RSAG     fp32 like UG_COMM_FP32 but as ncclReduceScatter(AVG) + ncclAllGather in place on the bucket (every rank
//                         owns one slice of the sum; both halves are single collectives RCCL can spread over all seven xGMI links
//                         of a rank at once), a short ncclAllReduce for the tail that does not divide by the world size
// ug_comm_allgather moves opaque bytes (the per-token embedding-lookup gradient rows + their ids, unigen_hip/ddp.py).
#include "common.h"
#include "unigen_hip.h"
#include <dlfcn.h>
#include <stdlib.h>
#include <string.h>
#include <new>

// The handful of RCCL types and enumerators this file needs, declared here with the values of rccl.h (NCCL's stable ABI) so that the
// library builds on hosts without RCCL headers; every function is resolved with dlsym at ug_comm_init time.
extern "C" {
typedef struct ncclComm* ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef int ncclResult_t;
typedef int ncclDataType_t;
typedef int ncclRedOp_t;
}
enum : int { ncclSuccess = 0 };
enum : int { ncclUint8 = 1, ncclFloat32 = 7, ncclBfloat16 = 9 };
enum : int { ncclSum = 0, ncclAvg = 4 };

namespace {

struct RcclApi {
  void* dl = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*ReduceScatter)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
};
RcclApi g_rccl;

int load_rccl() {
  if (g_rccl.dl) return UG_OK;
  const char* env = getenv("UNIGEN_RCCL_LIB");
  void* h = nullptr;
  if (env && *env) h = dlopen(env, RTLD_NOW | RTLD_LOCAL);
  if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD);          // the copy the host framework already mapped
  if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_NOLOAD);
  if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
  if (!h) h = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_LOCAL);
  if (!h) {
    ug_set_error("ug_comm: librccl.so.1 not found (set UNIGEN_RCCL_LIB): %s", dlerror());
    return UG_ERR_STATE;
  }
#define UG_SYM(field, name)                                                              \
  *reinterpret_cast<void**>(&g_rccl.field) = dlsym(h, name);                             \
  if (!g_rccl.field) { ug_set_error("ug_comm: %s missing from librccl", name); return UG_ERR_STATE; }
  UG_SYM(GetUniqueId, "ncclGetUniqueId")
  UG_SYM(CommInitRank, "ncclCommInitRank")
  UG_SYM(CommDestroy, "ncclCommDestroy")
  UG_SYM(AllReduce, "ncclAllReduce")
  UG_SYM(AllGather, "ncclAllGather")
  UG_SYM(ReduceScatter, "ncclReduceScatter")
  UG_SYM(Send, "ncclSend")
  UG_SYM(Recv, "ncclRecv")
  UG_SYM(GroupStart, "ncclGroupStart")
  UG_SYM(GroupEnd, "ncclGroupEnd")
  UG_SYM(GetErrorString, "ncclGetErrorString")
#undef UG_SYM
  g_rccl.dl = h;
  return UG_OK;
}

#define UG_NCCL(call)                                                                    \
  do {                                                                                   \
    ncclResult_t r__ = (call);                                                           \
    if (r__ != ncclSuccess) {                                                            \
      ug_set_error("%s failed: %s", #call, g_rccl.GetErrorString(r__));                  \
      return UG_ERR_LAUNCH;                                                              \
    }                                                                                    \
  } while (0)

}  // namespace

struct ug_comm {
  ncclComm_t comm;
  int world, rank;
  hipStream_t stream;          // the side stream every bucket runs on
  hipEvent_t ready, done;      // re-recorded per bucket / per wait
  bf16_t* stage;               // [2][cap] bf16: send | recv
  int64_t cap;                 // elements per half
  int64_t bytes_on_wire;       // payload handed to the collectives since init
};

extern "C" int ug_comm_unique_id(void* id128) {
  UG_REQUIRE(id128 != nullptr, "ug_comm_unique_id: null output");
  if (int rc = load_rccl()) return rc;
  ncclUniqueId id;
  UG_NCCL(g_rccl.GetUniqueId(&id));
  static_assert(sizeof(id) == UG_COMM_ID_BYTES, "ncclUniqueId size");
  memcpy(id128, &id, sizeof(id));
  return UG_OK;
}

extern "C" int ug_comm_init(ug_comm** out, int world, int rank, const void* id128, int64_t max_bucket_elems) {
  UG_REQUIRE(out && id128 && world >= 1 && rank >= 0 && rank < world && max_bucket_elems > 0,
             "ug_comm_init: need an output pointer, the 128-byte id of rank 0, 0 <= rank < world and a positive bucket size");
  if (int rc = load_rccl()) return rc;
  ug_comm* c = new (std::nothrow) ug_comm;
  UG_REQUIRE(c != nullptr, "ug_comm_init: out of host memory");
  memset(c, 0, sizeof(*c));
  c->world = world; c->rank = rank;
  ncclUniqueId id;
  memcpy(&id, id128, sizeof(id));
  ncclResult_t r = g_rccl.CommInitRank(&c->comm, world, id, rank);
  if (r != ncclSuccess) {
    ug_set_error("ncclCommInitRank(world=%d, rank=%d) failed: %s", world, rank, g_rccl.GetErrorString(r));
    delete c;
    return UG_ERR_LAUNCH;
  }
  // slices of 8-element granularity per rank, padded bucket
  c->cap = ((max_bucket_elems + 8 * world - 1) / (8 * world)) * (8 * world);
  if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess ||
      hipEventCreateWithFlags(&c->ready, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&c->done, hipEventDisableTiming) != hipSuccess ||
      hipMalloc(&c->stage, (size_t)c->cap * 2 * sizeof(bf16_t)) != hipSuccess) {
    (void)hipGetLastError();
    ug_set_error("ug_comm_init: stream / event / %zu-byte staging allocation failed", (size_t)c->cap * 4);
    (void)ug_comm_destroy(c);
    return UG_ERR_STATE;
  }
  *out = c;
  return UG_OK;
}

extern "C" int ug_comm_destroy(ug_comm* c) {
  if (!c) return UG_OK;
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  if (c->stage) (void)hipFree(c->stage);
  if (c->ready) (void)hipEventDestroy(c->ready);
  if (c->done) (void)hipEventDestroy(c->done);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  if (c->comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(c->comm);
  delete c;
  return UG_OK;
}

extern "C" int ug_comm_allreduce_bucket(ug_comm* c, float* grad, int64_t n, int mode, hipStream_t producer) {
  UG_REQUIRE(c && grad && n > 0 && ug_aligned16(grad), "ug_comm_allreduce_bucket: need a communicator and a 16-byte aligned bucket");
  UG_REQUIRE(mode == UG_COMM_FP32 || mode == UG_COMM_BF16_FP32ACC || mode == UG_COMM_BF16 || mode == UG_COMM_FP32_RSAG,
             "ug_comm_allreduce_bucket: unknown mode %d", mode);
  const int W = c->world;
  const int64_t chunk = ((n + 8 * W - 1) / (8 * W)) * 8, n_pad = chunk * W;
  UG_REQUIRE(mode == UG_COMM_FP32 || mode == UG_COMM_FP32_RSAG || n_pad <= c->cap, "ug_comm_allreduce_bucket: bucket of %ld elements exceeds the %ld the communicator was "
             "created for", (long)n, (long)c->cap);
  // the side stream starts when the producer stream has retired what is queued on it now (the kernels that wrote the bucket)
  UG_HIP(hipEventRecord(c->ready, producer));
  UG_HIP(hipStreamWaitEvent(c->stream, c->ready, 0));
  hipStream_t s = c->stream;
  if (mode == UG_COMM_FP32) {
    UG_NCCL(g_rccl.AllReduce(grad, grad, (size_t)n, ncclFloat32, ncclAvg, c->comm, s));
    c->bytes_on_wire += n * 4;
    return UG_OK;
  }
  if (mode == UG_COMM_FP32_RSAG) {
    // in place: rank r's slice of the mean lands at grad + r * part (NCCL's in-place reduce-scatter convention), then every
    // rank's slice is gathered back around it; part is a multiple of 4 elements so slices stay 16-byte aligned
    const int64_t part = (n / ((int64_t)W * 4)) * 4, body = part * W;
    if (part > 0) {
      UG_NCCL(g_rccl.ReduceScatter(grad, grad + (int64_t)c->rank * part, (size_t)part, ncclFloat32, ncclAvg, c->comm, s));
      UG_NCCL(g_rccl.AllGather(grad + (int64_t)c->rank * part, grad, (size_t)part, ncclFloat32, c->comm, s));
    }
    if (n > body) UG_NCCL(g_rccl.AllReduce(grad + body, grad + body, (size_t)(n - body), ncclFloat32, ncclAvg, c->comm, s));
    c->bytes_on_wire += n * 4;
    return UG_OK;
  }
  bf16_t* send = c->stage;
  bf16_t* recv = c->stage + c->cap;
  if (mode == UG_COMM_BF16) {
    if (int rc = ug_grad_pack_bf16(grad, send, n, 1.0f / (float)W, s)) return rc;
    UG_NCCL(g_rccl.AllReduce(send, send, (size_t)n, ncclBfloat16, ncclSum, c->comm, s));
    c->bytes_on_wire += n * 2;
    return ug_grad_unpack_bf16(send, grad, n, s);
  }
  if (n_pad > n) UG_HIP(hipMemsetAsync(send + n, 0, (size_t)(n_pad - n) * sizeof(bf16_t), s));
  if (int rc = ug_grad_pack_bf16(grad, send, n, 1.0f, s)) return rc;
  UG_NCCL(g_rccl.GroupStart());                          // all-to-all: recv[j] = rank j's copy of this rank's slice
  ncclResult_t first = ncclSuccess;                      // (an error inside the bracket must not leave the group open: every
  const char* what = "";                                 //  later collective of this thread would queue into it and hang)
  for (int j = 0; j < W && first == ncclSuccess; ++j) {
    first = g_rccl.Send(send + (int64_t)j * chunk, (size_t)chunk, ncclBfloat16, j, c->comm, s);
    what = "ncclSend";
    if (first == ncclSuccess) {
      first = g_rccl.Recv(recv + (int64_t)j * chunk, (size_t)chunk, ncclBfloat16, j, c->comm, s);
      what = "ncclRecv";
    }
  }
  const ncclResult_t closed = g_rccl.GroupEnd();
  if (first != ncclSuccess || closed != ncclSuccess) {
    ug_set_error("ug_comm all-to-all: %s failed: %s", first != ncclSuccess ? what : "ncclGroupEnd",
                 g_rccl.GetErrorString(first != ncclSuccess ? first : closed));
    return UG_ERR_LAUNCH;
  }
  bf16_t* mine = send + (int64_t)c->rank * chunk;        // (this rank's own packed slice has been sent: reuse it)
  if (int rc = ug_grad_sum_shards_bf16(recv, W, chunk, mine, chunk, 1.0f / (float)W, s)) return rc;
  UG_NCCL(g_rccl.AllGather(mine, recv, (size_t)chunk, ncclBfloat16, c->comm, s));
  c->bytes_on_wire += 2 * n_pad * 2;
  return ug_grad_unpack_bf16(recv, grad, n, s);
}

extern "C" int ug_comm_allgather(ug_comm* c, const void* send, void* recv, int64_t bytes_per_rank, hipStream_t producer) {
  UG_REQUIRE(c && send && recv && bytes_per_rank > 0, "ug_comm_allgather: need a communicator, both buffers and a positive size");
  UG_HIP(hipEventRecord(c->ready, producer));
  UG_HIP(hipStreamWaitEvent(c->stream, c->ready, 0));
  UG_NCCL(g_rccl.AllGather(send, recv, (size_t)bytes_per_rank, ncclUint8, c->comm, c->stream));
  return UG_OK;                                  // (bytes_on_wire counts the gradient buckets only; the host accounts for what it gathers)
}

extern "C" int ug_comm_wait(ug_comm* c, hipStream_t consumer) {
  UG_REQUIRE(c != nullptr, "ug_comm_wait: null communicator");
  UG_HIP(hipEventRecord(c->done, c->stream));
  UG_HIP(hipStreamWaitEvent(consumer, c->done, 0));
  return UG_OK;
}

extern "C" int64_t ug_comm_bytes_on_wire(const ug_comm* c) { return c ? c->bytes_on_wire : 0; }
