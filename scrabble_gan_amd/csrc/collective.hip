// The collective of the data-parallel train step behind the C-ABI: SUM all-reduce of a flat
// device buffer over RCCL (xGMI inside one 8 x MI355X node).  The reference has no distributed code; the contract is
// BASELINE.json's north_star ("RCCL all-reduce of G/D/R gradients") and SURVEY fact 5 (gradients are SUMS over the batch,
// /root/reference/src/bigacgan/data_utils.py:450,454,458,467, so ranks add, never average).
//
// librccl is resolved at first use (dlsym on the process image first -- PyTorch-ROCm ships and loads its own librccl, and
// two RCCL copies in one process must not both own the devices -- then dlopen), so libscrabble_hip.so itself carries no
// link-time dependency on it and single-GPU users never load it through this path.
#include "sg_common.h"
#include <dlfcn.h>
#include <string.h>

namespace {
struct UniqueId { char internal[128]; };
typedef void* Comm;
typedef int (*fn_all_reduce)(const void*, void*, size_t, int, int, Comm, hipStream_t);
typedef int (*fn_get_id)(UniqueId*);
typedef int (*fn_init_rank)(Comm*, int, UniqueId, int);
typedef int (*fn_destroy)(Comm);

struct Rccl {
  fn_all_reduce all_reduce = nullptr;
  fn_get_id get_id = nullptr;
  fn_init_rank init_rank = nullptr;
  fn_destroy destroy = nullptr;
  bool ok = false;
};

Rccl& rccl() {
  static Rccl r = [] {
    Rccl x;
    void* h = RTLD_DEFAULT;
    if (!dlsym(h, "ncclAllReduce")) {
      h = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
      if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
      if (!h) return x;
    }
    x.all_reduce = (fn_all_reduce)dlsym(h, "ncclAllReduce");
    x.get_id = (fn_get_id)dlsym(h, "ncclGetUniqueId");
    x.init_rank = (fn_init_rank)dlsym(h, "ncclCommInitRank");
    x.destroy = (fn_destroy)dlsym(h, "ncclCommDestroy");
    x.ok = x.all_reduce && x.get_id && x.init_rank && x.destroy;
    return x;
  }();
  return r;
}
}  // namespace

// dtype: SG_DTYPE_F32 0, SG_DTYPE_BF16 1, SG_DTYPE_F64 3 (SG_DTYPE_FP8_E4M3 2 is a storage type, not reducible)
extern "C" int sg_allreduce_sum(void* buf, long n, int dtype, void* comm, void* stream) {
  if (!buf || !comm || n < 0) return SG_ERR_ARG;
  if (n == 0) return SG_OK;
  int nccl_type;
  switch (dtype) {
    case 0: nccl_type = 7; break;   // ncclFloat32
    case 1: nccl_type = 9; break;   // ncclBfloat16
    case 3: nccl_type = 8; break;   // ncclFloat64
    default: return SG_ERR_UNSUPPORTED;
  }
  Rccl& r = rccl();
  if (!r.ok) return SG_ERR_UNSUPPORTED;
  return r.all_reduce(buf, buf, (size_t)n, nccl_type, /*ncclSum*/ 0, (Comm)comm, (hipStream_t)stream) == 0 ? SG_OK : SG_ERR_LAUNCH;
}

// id128: 128 bytes written by the rank that creates the job's id (every rank then passes the same bytes to init_rank)
extern "C" int sg_rccl_unique_id(void* id128) {
  if (!id128) return SG_ERR_ARG;
  Rccl& r = rccl();
  if (!r.ok) return SG_ERR_UNSUPPORTED;
  UniqueId id;
  if (r.get_id(&id) != 0) return SG_ERR_LAUNCH;
  memcpy(id128, id.internal, sizeof(id.internal));
  return SG_OK;
}

extern "C" int sg_rccl_comm_init_rank(void** comm, int nranks, const void* id128, int rank) {
  if (!comm || !id128 || nranks < 1 || rank < 0 || rank >= nranks) return SG_ERR_ARG;
  Rccl& r = rccl();
  if (!r.ok) return SG_ERR_UNSUPPORTED;
  UniqueId id;
  memcpy(id.internal, id128, sizeof(id.internal));
  Comm c = nullptr;
  if (r.init_rank(&c, nranks, id, rank) != 0) return SG_ERR_LAUNCH;
  *comm = c;
  return SG_OK;
}

extern "C" int sg_rccl_comm_destroy(void* comm) {
  if (!comm) return SG_OK;
  Rccl& r = rccl();
  if (!r.ok) return SG_ERR_UNSUPPORTED;
  return r.destroy((Comm)comm) == 0 ? SG_OK : SG_ERR_LAUNCH;
}
