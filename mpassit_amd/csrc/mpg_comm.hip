// Multi-GPU verbs of the C-ABI, on librccl directly (no torch, no MPI): one process per GPU, the images find each other
// through a file that carries the RCCL unique id.
//
// What they replace: ESMF's per-Regrid source exchange (hidden in the route handle) and ESMF_FieldGather
// (write_data.F90:1006-1453) -- with target rows sharded over the GPUs (regDecomp = (/1, npets/), model_grid.F90:693) and
// source cells owned in contiguous id blocks (model_grid.F90:423-438), every rank needs the cells its rows reference that
// another rank owns: one grouped ncclSend / ncclRecv exchange per field batch (point-to-point over xGMI; with banded cell
// numbering only the two row-block neighbours have anything to send).  The schedule is the one of mpassit_amd/dist.py
// (HaloSchedule.build), restated here so that a C or Fortran host has it without Python; tests/test_comm_plan.py checks
// the two against each other on the CPU, tests/test_comm_gpu.py runs the RCCL path with a world of one on the GPU box.
//
// RCCL is loaded with dlopen at mpg_comm_init: the library itself keeps no link-time dependency on it (single-GPU hosts
// never touch it), and in a process that already holds a RCCL (PyTorch's) the same image is used.
#include <dlfcn.h>
#include <fcntl.h>
#include <rccl/rccl.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <memory>
#include <mutex>
#include <thread>

#include "mpg_internal.h"

namespace {
struct Rccl {
  void *lib = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  const char *(*GetErrorString)(ncclResult_t) = nullptr;
} R;

int rccl_load() {
  if (R.lib) return MPG_SUCCESS;
  const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  for (const char *n : names)
    if ((R.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
  if (!R.lib) {
    mpg_set_error("mpg_comm_init: cannot load librccl (%s)", dlerror());
    return MPG_ERR_UNSUPPORTED;
  }
#define SYM(field, name)                                                      \
  if (!(*(void **)(&R.field) = dlsym(R.lib, name))) {                         \
    mpg_set_error("mpg_comm_init: librccl lacks %s", name);                   \
    R.lib = nullptr;                                                          \
    return MPG_ERR_UNSUPPORTED;                                               \
  }
  SYM(GetUniqueId, "ncclGetUniqueId")
  SYM(CommInitRank, "ncclCommInitRank")
  SYM(CommDestroy, "ncclCommDestroy")
  SYM(Send, "ncclSend")
  SYM(Recv, "ncclRecv")
  SYM(GroupStart, "ncclGroupStart")
  SYM(GroupEnd, "ncclGroupEnd")
  SYM(AllGather, "ncclAllGather")
  SYM(GetErrorString, "ncclGetErrorString")
#undef SYM
  return MPG_SUCCESS;
}

#define MPG_NCCL(call)                                                                             \
  do {                                                                                             \
    ncclResult_t r_ = (call);                                                                      \
    if (r_ != ncclSuccess) {                                                                       \
      mpg_set_error("%s failed: %s (%s:%d)", #call, R.GetErrorString(r_), __FILE__, __LINE__);     \
      return MPG_ERR_HIP;                                                                          \
    }                                                                                              \
  } while (0)

// an open ncclGroupStart is closed on every path out of the scope (a send that fails must not leave the communicator inside a group)
struct GroupScope {
  bool open = false;
  ncclResult_t start() {
    ncclResult_t r = R.GroupStart();
    open = r == ncclSuccess;
    return r;
  }
  ncclResult_t end() {
    open = false;
    return R.GroupEnd();
  }
  ~GroupScope() {
    if (open) (void)R.GroupEnd();
  }
};
struct EventScope {
  hipEvent_t e = nullptr;
  ~EventScope() {
    if (e) (void)hipEventDestroy(e);
  }
};
}  // namespace

// One transfer of a grouped exchange as a rank states it: `ns` bytes from sbuf to `peer`, `nr` bytes from `peer` into rbuf
// (either may be 0).  A real communicator turns the list into ONE ncclGroupStart / ncclSend / ncclRecv / ncclGroupEnd.
struct P2P {
  int peer;
  const void *sbuf;
  size_t ns;
  void *rbuf;
  size_t nr;
};

// Rehearsal on ONE GPU (mpg_comm_virtual; tests and tools only): V virtual ranks, one host thread each, share the ONE-rank
// RCCL communicator of the process.  RCCL refuses two ranks on one card, but it accepts ncclSend / ncclRecv with peer == self
// inside one group, matched in issue order.  So every collective step of the verbs below is a rendezvous of the V threads;
// the last one to arrive issues, for all of them, the very calls a real rank would issue -- the same dlsym'd entry points,
// the same device pointers and byte counts out of each rank's own schedule, each transfer on its rank's stream -- with every
// peer mapped to rank 0 of the real communicator: sends in (source rank, list order), each followed by the receive the
// DESTINATION rank listed for that source.  A send whose size differs from the receive it meets is a schedule bug and is
// reported as one.  What this cannot show is bytes crossing xGMI.
struct VGroup {
  int V = 0;
  ncclComm_t comm = nullptr;
  std::mutex mu;                 // rendezvous state
  std::condition_variable cv;
  int arrived = 0;
  uint64_t gen = 0;
  int rc = MPG_SUCCESS;
  std::string err;
  std::mutex work_mu;            // device work of the virtual ranks' threads outside the rendezvous runs one at a time
  std::vector<const void *> ag_send;
  int64_t ag_nbytes = 0;
  std::vector<char> ag_all;
  std::vector<std::vector<P2P>> ops;
  std::vector<hipStream_t> streams;
  int refs = 0;
  int64_t n_send_calls = 0, n_recv_calls = 0, n_groups = 0, n_allgathers = 0;   // what really went through RCCL
};

struct mpg_comm_s {
  int rank = 0, nranks = 1;
  ncclComm_t comm = nullptr;
  VGroup *vg = nullptr;          // set on a virtual rank (and on the real communicator that owns the group)
  bool is_virtual = false;
};

// ---- the schedule (pure host logic; mirrors dist.HaloSchedule.build) ------------------------------------------------------
struct HaloPlan {
  int rank = 0, nranks = 1, mode = 0;       // mode 0: range, 1: compact, 2: owned (the caller's own partition of the cells)
  int64_t n_local = 0, own0 = 0, own1 = 0, base = 0, own_pos0 = 0, own_pos1 = 0;
  std::vector<int64_t> send_a, send_b;      // range: per peer, offsets [a, b) inside the own block
  std::vector<std::vector<int32_t>> send_ids;   // compact / owned: per peer (self included), offsets inside the own block / owned list
  std::vector<int64_t> recv_a, recv_b;      // range / compact: per peer, destination range in the local index space
  std::vector<std::vector<int32_t>> recv_ids;   // owned: per peer (self included), destination positions in the local index space
  int64_t send_count(int q) const { return mode == 0 ? send_b[q] - send_a[q] : (int64_t)send_ids[q].size(); }
  int64_t recv_count(int q) const { return mode == 2 ? (int64_t)recv_ids[q].size() : recv_b[q] - recv_a[q]; }
};

static void para_block(int64_t n, int world, int r, int64_t *a, int64_t *b) {   // model_grid.F90:2428-2441, 0-based half-open
  const int64_t w1 = n / world, w2 = n % world;
  *a = r * w1 + std::min<int64_t>(r, w2);
  *b = *a + w1 + (w2 > r ? 1 : 0);
}

struct Vote { int64_t want_range, nlo, nhi, empty; };

// votes: what every rank's needed ids look like; -> mode and the ownership blocks
static void plan_blocks(int world, int64_t n_cells, int ownership, const std::vector<Vote> &votes, int *mode,
                        std::vector<std::pair<int64_t, int64_t>> &blocks) {
  bool all_range = true;
  for (auto &v : votes) all_range = all_range && v.want_range;
  *mode = all_range ? 0 : 1;
  blocks.resize(world);
  bool aligned = *mode == 0 && ownership == 0;
  std::vector<int64_t> los(world), his(world);
  if (aligned) {
    int64_t prev = 0;
    bool any = false;
    for (auto &v : votes)
      if (!v.empty) {
        prev = any ? std::min(prev, v.nlo) : v.nlo;
        any = true;
      }
    for (int q = 0; q < world; ++q) {
      los[q] = votes[q].nlo;
      his[q] = votes[q].nhi;
      if (votes[q].empty) los[q] = his[q] = prev;   // a rank that needs nothing owns an empty block right after its predecessor
      prev = his[q];
    }
    for (int q = 1; q < world; ++q)
      if (los[q] < los[q - 1] || his[q] < his[q - 1]) aligned = false;   // row blocks do not map to increasing id ranges
  }
  if (aligned) {
    std::vector<int64_t> bnd(1, los[0]);
    for (int q = 1; q < world; ++q) {
      int64_t b = los[q] < his[q - 1] ? (los[q] + his[q - 1]) / 2 : los[q];
      bnd.push_back(std::max(b, bnd.back()));
    }
    bnd.push_back(std::max(his[world - 1], bnd.back()));
    for (int q = 0; q < world; ++q) blocks[q] = {bnd[q], bnd[q + 1]};
  } else {
    for (int q = 0; q < world; ++q) para_block(n_cells, world, q, &blocks[q].first, &blocks[q].second);
  }
}

static Vote make_vote(const int32_t *needed, int64_t n, int64_t n_cells, int world, int rank) {
  int64_t a, b;
  para_block(n_cells, world, rank, &a, &b);
  Vote v;
  v.empty = n == 0;
  v.nlo = n ? needed[0] : a;
  v.nhi = n ? (int64_t)needed[n - 1] + 1 : a;
  v.want_range = v.empty || (double)(v.nhi - v.nlo) <= 1.25 * (double)n;   // banded numbering: covering range ~ count
  return v;
}

static void span_of(const Vote &v, const std::pair<int64_t, int64_t> &blk, int64_t *lo, int64_t *hi) {
  *lo = std::min(v.nlo, blk.first);
  *hi = std::max(v.nhi, blk.second);
  if (v.empty) {
    *lo = blk.first;
    *hi = blk.second;
  }
}

// range mode: everything follows from the votes (a rank's span is a function of its vote and block)
static void plan_range(HaloPlan &p, const std::vector<Vote> &votes, const std::vector<std::pair<int64_t, int64_t>> &blocks) {
  const int world = p.nranks, rank = p.rank;
  int64_t lo, hi;
  span_of(votes[rank], blocks[rank], &lo, &hi);
  const int64_t c0 = blocks[rank].first, c1 = blocks[rank].second;
  p.own0 = c0; p.own1 = c1; p.base = lo; p.n_local = hi - lo; p.own_pos0 = c0 - lo; p.own_pos1 = c1 - lo;
  p.send_a.assign(world, 0); p.send_b.assign(world, 0); p.recv_a.assign(world, 0); p.recv_b.assign(world, 0);
  for (int q = 0; q < world; ++q) {
    if (q == rank) continue;
    int64_t qlo, qhi;
    span_of(votes[q], blocks[q], &qlo, &qhi);
    int64_t a = std::max(qlo, c0), b = std::min(qhi, c1);          // what q wants from my block
    if (b > a) { p.send_a[q] = a - c0; p.send_b[q] = b - c0; }
    a = std::max(lo, blocks[q].first); b = std::min(hi, blocks[q].second);   // what I want from q's block
    if (b > a) { p.recv_a[q] = a - lo; p.recv_b[q] = b - lo; }
  }
}

// compact mode: needs every rank's id list
static void plan_compact(HaloPlan &p, const std::vector<std::pair<int64_t, int64_t>> &blocks, const int64_t *n_needed,
                         const int32_t *const *needed) {
  const int world = p.nranks, rank = p.rank;
  const int64_t c0 = blocks[rank].first, c1 = blocks[rank].second;
  p.own0 = c0; p.own1 = c1; p.base = 0; p.n_local = n_needed[rank]; p.own_pos0 = p.own_pos1 = 0;
  p.send_ids.assign(world, {});
  p.recv_a.assign(world, 0); p.recv_b.assign(world, 0);
  const int32_t *mine = needed[rank];
  for (int q = 0; q < world; ++q) {
    const int32_t *th = needed[q], *e = th + n_needed[q];
    const int32_t *a = std::lower_bound(th, e, (int32_t)std::min<int64_t>(c0, 0x7fffffff));
    const int32_t *b = std::lower_bound(th, e, (int32_t)std::min<int64_t>(c1, 0x7fffffff));
    for (const int32_t *it = a; it < b; ++it) p.send_ids[q].push_back((int32_t)(*it - c0));
    p.recv_a[q] = std::lower_bound(mine, mine + n_needed[rank], (int32_t)std::min<int64_t>(blocks[q].first, 0x7fffffff)) - mine;
    p.recv_b[q] = std::lower_bound(mine, mine + n_needed[rank], (int32_t)std::min<int64_t>(blocks[q].second, 0x7fffffff)) - mine;
  }
}

// owned mode (round 5): the cells are partitioned by the CALLER -- owned[q] = rank q's sorted ids, disjoint, any shape: the model's own
// decomposition of a coupled run, or a partition that follows the target rows of a mesh without banded numbering (bench.py gives every
// cell to the lowest rank that references it: only the overlap of neighbouring row blocks then travels, where equal id blocks of a
// Morton-numbered mesh send 7/8 of everything at 8 ranks).  Local space = the rank's sorted needed ids (as in compact mode); what a
// peer sends are its owned cells among them, in id order on both sides.  -> the first needed id that nobody owns, or -1
static int64_t plan_owned(HaloPlan &p, const int64_t *n_needed, const int32_t *const *needed, const int64_t *n_owned, const int32_t *const *owned) {
  const int world = p.nranks, rank = p.rank;
  p.mode = 2;
  p.own0 = 0; p.own1 = n_owned[rank]; p.base = 0; p.n_local = n_needed[rank]; p.own_pos0 = p.own_pos1 = 0;
  p.send_ids.assign(world, {});
  p.recv_ids.assign(world, {});
  const int32_t *mine = needed[rank], *mown = owned[rank];
  std::vector<char> found((size_t)n_needed[rank], 0);
  for (int q = 0; q < world; ++q) {
    // what q needs of my cells: needed[q] ^ owned[rank], as offsets into my owned list
    const int32_t *a = needed[q], *ae = a + n_needed[q], *b = mown, *be = mown + n_owned[rank];
    while (a < ae && b < be) {
      if (*a < *b) ++a;
      else if (*b < *a) ++b;
      else { p.send_ids[q].push_back((int32_t)(b - mown)); ++a; ++b; }
    }
    // what I need of q's cells: needed[rank] ^ owned[q], as positions in my local space
    a = mine; ae = mine + n_needed[rank]; b = owned[q]; be = b + n_owned[q];
    while (a < ae && b < be) {
      if (*a < *b) ++a;
      else if (*b < *a) ++b;
      else { p.recv_ids[q].push_back((int32_t)(a - mine)); found[(size_t)(a - mine)] = 1; ++a; ++b; }
    }
  }
  for (int64_t i = 0; i < n_needed[rank]; ++i)
    if (!found[(size_t)i]) return mine[i];
  return -1;
}

struct mpg_halo_s {
  mpg_comm_s *comm = nullptr;
  HaloPlan plan;
  DevBuf<int32_t> recv_ids_dev;             // owned: all peers' destination positions back to back
  std::vector<int64_t> rids_off;
  std::vector<int64_t> soff, roff;          // per peer element offsets (per row) inside the packed buffers
  int64_t stot = 0, rtot = 0;
  DevBuf<int32_t> send_ids_dev;             // compact: all peers' offset lists back to back
  std::vector<int64_t> ids_off;
  DevBuf<char> sendbuf, recvbuf;            // sized for buf_rows rows of buf_es bytes per element
  int64_t buf_bytes = 0;
};

// dst[k][i] = src[k * ld + ids[i]] for elements of 4 or 8 bytes
template <typename T>
__global__ __launch_bounds__(256) void k_pack_ids(const T *__restrict__ src, int64_t ld, int nrows, const int32_t *__restrict__ ids, int64_t n,
                                                  T *__restrict__ dst) {
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int32_t c = ids[i];
  for (int k = 0; k < nrows; ++k) dst[(int64_t)k * n + i] = src[(int64_t)k * ld + c];
}

// the same for elements of `ew` 4-byte words (a file-order source row [nlev] of float32 / float64 is one element: the strip a
// peer wants is then whole rows of the [cell][nlev] slab); one thread per word, consecutive lanes on consecutive words
__global__ __launch_bounds__(256) void k_pack_ids_w(const uint32_t *__restrict__ src, int64_t ld, int nrows, const int32_t *__restrict__ ids, int64_t n, int ew,
                                                    uint32_t *__restrict__ dst) {
  const int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (t >= n * ew) return;
  const int64_t i = t / ew;
  const int w = (int)(t - i * ew);
  const int64_t c = ids[i];
  for (int k = 0; k < nrows; ++k) dst[((int64_t)k * n + i) * ew + w] = src[((int64_t)k * ld + c) * ew + w];
}

static int pack_ids(const void *src, int64_t ld, int nrows, int elem_bytes, const int32_t *ids, int64_t n, void *dst, hipStream_t s) {
  if (n <= 0 || nrows <= 0) return MPG_SUCCESS;
  if (elem_bytes == 8) {
    k_pack_ids<unsigned long long><<<(unsigned)((n + 255) / 256), 256, 0, s>>>((const unsigned long long *)src, ld, nrows, ids, n, (unsigned long long *)dst);
  } else if (elem_bytes == 4) {
    k_pack_ids<uint32_t><<<(unsigned)((n + 255) / 256), 256, 0, s>>>((const uint32_t *)src, ld, nrows, ids, n, (uint32_t *)dst);
  } else {
    const int ew = elem_bytes / 4;
    k_pack_ids_w<<<(unsigned)((n * ew + 255) / 256), 256, 0, s>>>((const uint32_t *)src, ld, nrows, ids, n, ew, (uint32_t *)dst);
  }
  MPG_HIP(hipGetLastError());
  return MPG_SUCCESS;
}

// the inverse of the pack: dst[k * ld + pos[i]] = src[k][i], elements of `ew` 4-byte words (owned mode's unpack: a peer's cells sit
// anywhere among the rank's sorted needed ids)
__global__ __launch_bounds__(256) void k_unpack_ids_w(const uint32_t *__restrict__ src, int64_t n, int nrows, const int32_t *__restrict__ pos, int64_t ld, int ew,
                                                      uint32_t *__restrict__ dst) {
  const int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (t >= n * ew) return;
  const int64_t i = t / ew;
  const int w = (int)(t - i * ew);
  const int64_t c = pos[i];
  for (int k = 0; k < nrows; ++k) dst[((int64_t)k * ld + c) * ew + w] = src[((int64_t)k * n + i) * ew + w];
}
static int unpack_ids(const void *src, int64_t n, int nrows, int elem_bytes, const int32_t *pos, int64_t ld, void *dst, hipStream_t s) {
  if (n <= 0 || nrows <= 0) return MPG_SUCCESS;
  const int ew = elem_bytes / 4;
  k_unpack_ids_w<<<(unsigned)((n * ew + 255) / 256), 256, 0, s>>>((const uint32_t *)src, n, nrows, pos, ld, ew, (uint32_t *)dst);
  MPG_HIP(hipGetLastError());
  return MPG_SUCCESS;
}

// The id file: rank 0 writes {magic, launch tag, wall-clock time, nranks, RCCL unique id} under a temporary name and renames
// it; the others poll for it.  What can go wrong is a file LEFT BEHIND by an earlier launch that died before rank 0 could
// remove it: a rank reading that id would sit in ncclCommInitRank for ever.  Hence
//   * rank 0 removes whatever is at `id_file` before it writes, and removes its own file once ncclCommInitRank has returned
//     (every rank has read it by then: the call is collective);
//   * a reader only accepts a file with the right magic and nranks, with ITS launch tag (a hash of MPASSIT_RUN_ID when
//     the launcher sets one -- tools/mpassit_ranks.py, bench.py do), and written no longer than MPG_COMM_STALE_S (300 s)
//     before this process loaded the library; anything else is ignored and the wait goes on;
//   * a launch WITHOUT a tag cannot tell its own file from one a crashed launch left behind minutes ago (a reader that starts
//     before its rank 0 would take the old id and sit in ncclCommInitRank until the deadline): several ranks without
//     MPASSIT_RUN_ID are refused.  MPG_COMM_ALLOW_UNTAGGED=1 overrides that for hosts that guarantee a fresh path per launch;
//     a reader then accepts only a file written at most MPG_COMM_STALE_S (then 30 s by default) before it loaded the library
//     whose bytes are still the same one second later (rank 0 replaces a stale file as soon as it starts);
//   * every wait has a deadline (MPG_COMM_TIMEOUT_S, default 120 s) -- for the file, and for ncclCommInitRank itself, which
//     runs on a helper thread: a peer that never arrives gives MPG_ERR_TIMEOUT, not a hang.  After a timeout the process
//     must exit (the helper thread is still inside RCCL); never restart a process that has touched the GPU.
namespace {
struct IdFile {
  char magic[8];
  uint64_t tag;
  int64_t written_ns;
  int32_t nranks, pad;
  ncclUniqueId id;
};
const char ID_MAGIC[8] = {'M', 'P', 'G', 'R', 'C', 'C', 'L', '2'};
int64_t wall_ns() {
  struct timespec ts;
  clock_gettime(CLOCK_REALTIME, &ts);
  return (int64_t)ts.tv_sec * 1000000000ll + ts.tv_nsec;
}
const int64_t g_loaded_ns = wall_ns();   // when this process loaded the library
uint64_t launch_tag() {                  // FNV-1a of MPASSIT_RUN_ID; 0 without one
  const char *e = getenv("MPASSIT_RUN_ID");
  if (!e || !*e) return 0;
  uint64_t h = 1469598103934665603ull;
  for (; *e; ++e) h = (h ^ (unsigned char)*e) * 1099511628211ull;
  return h ? h : 1;
}
double env_seconds(const char *name, double dflt) {
  const char *e = getenv(name);
  if (!e || !*e) return dflt;
  const double v = atof(e);
  return v > 0.0 ? v : dflt;
}
}  // namespace
double mpg_comm_timeout_s() { return env_seconds("MPG_COMM_TIMEOUT_S", 120.0); }

// Why a candidate id file is not this launch's (nullptr: it is).  Pure host logic, exported for the CPU tests.
extern "C" const char *mpg_comm_idfile_verdict(const void *bytes, int64_t nbytes, int nranks, uint64_t tag, int64_t reader_loaded_ns, double stale_s) {
  if (nbytes != (int64_t)sizeof(IdFile)) return "wrong size";
  const IdFile *f = (const IdFile *)bytes;
  if (memcmp(f->magic, ID_MAGIC, 8)) return "wrong magic";
  if (f->nranks != nranks) return "written for another number of ranks";
  if (f->tag != tag) return "written by another launch (MPASSIT_RUN_ID differs)";
  if ((double)(reader_loaded_ns - f->written_ns) * 1e-9 > stale_s) return "older than this process: left behind by an earlier launch";
  return nullptr;
}

extern "C" {

int mpg_comm_init(int rank, int nranks, const char *id_file, mpg_comm *out) {
  MPG_CHECK_INIT();
  MPG_ARG(out && nranks >= 1 && rank >= 0 && rank < nranks, "mpg_comm_init: bad rank / nranks");
  MPG_ARG(nranks == 1 || (id_file && *id_file), "mpg_comm_init: nranks > 1 needs the path of the id file");
  int rc = rccl_load();
  if (rc) return rc;
  const uint64_t tag = launch_tag();
  const bool untagged = nranks > 1 && tag == 0;
  if (untagged) {
    const char *ok = getenv("MPG_COMM_ALLOW_UNTAGGED");
    if (!ok || !*ok || *ok == '0') {
      mpg_set_error("mpg_comm_init: %d ranks need MPASSIT_RUN_ID in the environment, unique per launch (it tags the id file: without it a file left by a "
                    "crashed launch cannot be told from this launch's); MPG_COMM_ALLOW_UNTAGGED=1 overrides", nranks);
      return MPG_ERR_INVALID_ARG;
    }
  }
  const double timeout_s = mpg_comm_timeout_s(), stale_s = env_seconds("MPG_COMM_STALE_S", untagged ? 30.0 : 300.0);
  const int64_t t_start = wall_ns();
  IdFile f;
  memset(&f, 0, sizeof(f));
  if (rank == 0) {
    MPG_NCCL(R.GetUniqueId(&f.id));
    if (nranks > 1) {   // under a temporary name, then renamed: a reader never sees half a file
      memcpy(f.magic, ID_MAGIC, 8);
      f.tag = tag;
      f.written_ns = wall_ns();
      f.nranks = nranks;
      (void)unlink(id_file);   // whatever an earlier launch left there
      std::string tmp = std::string(id_file) + ".tmp";
      FILE *fp = fopen(tmp.c_str(), "wb");
      if (!fp || fwrite(&f, 1, sizeof(f), fp) != sizeof(f) || fclose(fp) || rename(tmp.c_str(), id_file)) {
        mpg_set_error("mpg_comm_init: cannot write the id file %s", id_file);
        return MPG_ERR_INVALID_ARG;
      }
    }
  } else {
    bool got = false;
    const char *why = "no file";
    while (!got && (double)(wall_ns() - t_start) * 1e-9 < timeout_s) {
      IdFile cand;
      FILE *fp = fopen(id_file, "rb");
      if (fp) {
        const size_t n = fread(&cand, 1, sizeof(cand), fp);
        fclose(fp);
        why = mpg_comm_idfile_verdict(&cand, (int64_t)n, nranks, tag, g_loaded_ns, stale_s);
        if (!why && untagged) {   // no tag to tell launches apart: the file must survive a second look (rank 0 replaces what it finds)
          usleep(1000000);
          IdFile again;
          FILE *f2 = fopen(id_file, "rb");
          const size_t n2 = f2 ? fread(&again, 1, sizeof(again), f2) : 0;
          if (f2) fclose(f2);
          if (n2 != sizeof(again) || memcmp(&again, &cand, sizeof(cand))) why = "replaced while it was being read (an earlier launch's file)";
        }
        if (!why) {
          f = cand;
          got = true;
        }
      }
      if (!got) usleep(5000);
    }
    if (!got) {
      mpg_set_error("mpg_comm_init: rank %d gave up after %.0f s waiting for this launch's id file %s (last candidate: %s)", rank, timeout_s, id_file, why);
      return MPG_ERR_TIMEOUT;
    }
  }
  // ncclCommInitRank blocks until every rank has joined: on a helper thread, so that a missing peer is an error after
  // timeout_s instead of a hang
  struct Join {
    std::mutex mu;
    std::condition_variable cv;
    bool done = false;
    ncclResult_t res = ncclSuccess;
    ncclComm_t comm = nullptr;
  };
  auto join = std::make_shared<Join>();
  int device = 0;
  MPG_HIP(hipGetDevice(&device));
  const ncclUniqueId id = f.id;
  std::thread([join, device, nranks, id, rank]() {
    ncclComm_t c = nullptr;
    ncclResult_t r = hipSetDevice(device) == hipSuccess ? R.CommInitRank(&c, nranks, id, rank) : ncclUnhandledCudaError;
    std::lock_guard<std::mutex> lk(join->mu);
    join->res = r;
    join->comm = c;
    join->done = true;
    join->cv.notify_all();
  }).detach();
  bool joined;
  {
    std::unique_lock<std::mutex> lk(join->mu);
    const double left = timeout_s - (double)(wall_ns() - t_start) * 1e-9;
    joined = join->cv.wait_for(lk, std::chrono::duration<double>(left > 1.0 ? left : 1.0), [&] { return join->done; });
  }
  if (rank == 0 && nranks > 1) (void)unlink(id_file);   // read by everyone who joined; never left for a later launch
  if (!joined) {
    mpg_set_error("mpg_comm_init: rank %d of %d: ncclCommInitRank did not return within %.0f s (a peer is missing or read another launch's id); "
                  "exit this process", rank, nranks, timeout_s);
    return MPG_ERR_TIMEOUT;
  }
  if (join->res != ncclSuccess) {
    mpg_set_error("ncclCommInitRank failed: %s", R.GetErrorString(join->res));
    return MPG_ERR_HIP;
  }
  mpg_comm_s *c = new mpg_comm_s();
  c->rank = rank;
  c->nranks = nranks;
  c->comm = join->comm;
  *out = c;
  return MPG_SUCCESS;
}

int mpg_comm_destroy(mpg_comm c) {
  if (!c) return MPG_SUCCESS;
  if (c->is_virtual) {   // a virtual rank borrows the real communicator
    std::lock_guard<std::mutex> lk(c->vg->mu);
    --c->vg->refs;
    delete c;
    return MPG_SUCCESS;
  }
  if (c->vg) {
    MPG_ARG(c->vg->refs == 0, "mpg_comm_destroy: virtual ranks of this communicator are still alive (destroy them first)");
    delete c->vg;
  }
  if (c->comm) (void)R.CommDestroy(c->comm);
  delete c;
  return MPG_SUCCESS;
}

// Rehearsal only (struct VGroup above): virtual rank v_rank of v_nranks on top of `real`, the ONE-rank communicator of this
// process.  Every virtual rank is driven by its own host thread; the collective verbs (mpg_comm_allgather, mpg_halo_build,
// mpg_halo_exchange_dev, mpg_gather_rows) must be called by all of them, as on real ranks.
int mpg_comm_virtual(mpg_comm real, int v_rank, int v_nranks, mpg_comm *out) {
  MPG_CHECK_INIT();
  MPG_ARG(real && out && !real->is_virtual && real->nranks == 1, "mpg_comm_virtual: needs the one-rank communicator of this process (mpg_comm_init(0, 1, ...))");
  MPG_ARG(v_nranks >= 1 && v_nranks <= 64 && v_rank >= 0 && v_rank < v_nranks, "mpg_comm_virtual: bad rank / number of virtual ranks");
  static std::mutex mk;
  std::lock_guard<std::mutex> lk(mk);
  if (!real->vg) {
    VGroup *g = new VGroup();
    g->V = v_nranks;
    g->comm = real->comm;
    g->ag_send.assign(v_nranks, nullptr);
    g->ops.resize(v_nranks);
    g->streams.assign(v_nranks, nullptr);
    real->vg = g;
  }
  MPG_ARG(real->vg->V == v_nranks, "mpg_comm_virtual: this communicator already carries a virtual group of another size");
  mpg_comm_s *c = new mpg_comm_s();
  c->rank = v_rank;
  c->nranks = v_nranks;
  c->comm = real->comm;
  c->vg = real->vg;
  c->is_virtual = true;
  {
    std::lock_guard<std::mutex> l2(c->vg->mu);
    ++c->vg->refs;
  }
  *out = c;
  return MPG_SUCCESS;
}

// what the virtual ranks of `c`'s group have really put through RCCL so far (any pointer may be NULL)
int mpg_comm_virtual_stats(mpg_comm c, int64_t *groups, int64_t *sends, int64_t *recvs, int64_t *allgathers) {
  MPG_ARG(c && c->vg, "mpg_comm_virtual_stats: not a virtual rank (nor the communicator under one)");
  std::lock_guard<std::mutex> lk(c->vg->mu);
  if (groups) *groups = c->vg->n_groups;
  if (sends) *sends = c->vg->n_send_calls;
  if (recvs) *recvs = c->vg->n_recv_calls;
  if (allgathers) *allgathers = c->vg->n_allgathers;
  return MPG_SUCCESS;
}

int mpg_comm_info(mpg_comm c, int *rank, int *nranks) {
  MPG_ARG(c, "mpg_comm_info: NULL communicator");
  if (rank) *rank = c->rank;
  if (nranks) *nranks = c->nranks;
  return MPG_SUCCESS;
}

// hipStreamSynchronize with a deadline: a collective whose peer died never completes
static int stream_wait_deadline(hipStream_t s, const char *what) {
  const double timeout_s = mpg_comm_timeout_s();
  const int64_t t0 = wall_ns();
  for (;;) {
    hipError_t e = hipStreamQuery(s);
    if (e == hipSuccess) return MPG_SUCCESS;
    if (e != hipErrorNotReady) {
      mpg_set_error("%s: %s", what, hipGetErrorString(e));
      return MPG_ERR_HIP;
    }
    if ((double)(wall_ns() - t0) * 1e-9 > timeout_s) {
      mpg_set_error("%s: the exchange did not complete within %.0f s (a peer is gone); exit this process", what, timeout_s);
      return MPG_ERR_TIMEOUT;
    }
    usleep(50);
  }
}

}  // extern "C"

// Rendezvous of the virtual ranks' threads: `deposit` runs under the group's lock as a thread arrives, `leader_work` once,
// in the last thread to arrive, while the others wait; every thread returns the leader's rc (and error text).
template <typename FD, typename FL>
static int vg_rendezvous(VGroup *g, FD &&deposit, FL &&leader_work) {
  std::unique_lock<std::mutex> lk(g->mu);
  const uint64_t my_gen = g->gen;
  deposit();
  if (++g->arrived == g->V) {
    g->rc = leader_work();
    g->err = g->rc ? mpg_last_error() : "";
    g->arrived = 0;
    ++g->gen;
    g->cv.notify_all();
    return g->rc;
  }
  const double t = mpg_comm_timeout_s();
  if (!g->cv.wait_for(lk, std::chrono::duration<double>(t), [&] { return g->gen != my_gen; })) {
    mpg_set_error("virtual ranks: %d of %d arrived at a collective step within %.0f s (every virtual rank needs its own thread, and all of them "
                  "must make the same calls)", g->arrived, g->V, t);
    --g->arrived;
    return MPG_ERR_TIMEOUT;
  }
  if (g->rc) mpg_set_error("%s", g->err.c_str());
  return g->rc;
}

// the metadata all-gather of V virtual ranks: their contributions, in rank order, travel through ONE ncclAllGather of the
// one-rank communicator on the set-up stream (the call a real rank makes, on the stream it makes it on)
static int vg_allgather(mpg_comm_s *c, const void *send_host, int64_t nbytes, void *recv_host) {
  VGroup *g = c->vg;
  const int V = g->V;
  int rc = vg_rendezvous(
      g,
      [&] {
        g->ag_send[c->rank] = send_host;
        if (g->arrived == 0) g->ag_nbytes = nbytes;
        else if (g->ag_nbytes != nbytes) g->ag_nbytes = -1;
      },
      [&]() -> int {
        MPG_ARG(g->ag_nbytes == nbytes, "mpg_comm_allgather: the virtual ranks contribute different byte counts");
        hipStream_t s = mpg_setup_stream();
        const size_t tot = (size_t)nbytes * V;
        std::vector<char> cat(tot);
        for (int q = 0; q < V; ++q) memcpy(cat.data() + (size_t)q * nbytes, g->ag_send[q], (size_t)nbytes);
        TmpBuf<char> sb, rb;
        int r;
        if ((r = sb.alloc(tot)) || (r = rb.alloc(tot))) return r;
        g->ag_all.resize(tot);
        MPG_HIP(hipMemcpyAsync(sb.p, cat.data(), tot, hipMemcpyHostToDevice, s));
        MPG_NCCL(R.AllGather(sb.p, rb.p, tot, ncclChar, g->comm, s));
        ++g->n_allgathers;
        MPG_HIP(hipMemcpyAsync(g->ag_all.data(), rb.p, tot, hipMemcpyDeviceToHost, s));
        return stream_wait_deadline(s, "mpg_comm_allgather (virtual ranks)");
      });
  if (rc) return rc;
  memcpy(recv_host, g->ag_all.data(), (size_t)nbytes * V);   // stays valid until every rank has arrived at the NEXT step
  return MPG_SUCCESS;
}

// One grouped exchange.  A real rank: ncclGroupStart, its sends and receives, ncclGroupEnd, on its stream.  Virtual ranks:
// the lists of all V threads are matched (a send of rank r to rank q with the next receive rank q lists for r; equal sizes
// required -- the two ranks' schedules must agree) and go out as ONE group of sends and receives to self on rank 0's stream,
// with event edges from and to the other ranks' streams.
static int comm_exchange(mpg_comm_s *c, const std::vector<P2P> &ops, hipStream_t s) {
  if (!c->is_virtual) {
    if (c->nranks == 1) return MPG_SUCCESS;
    GroupScope grp;
    MPG_NCCL(grp.start());
    for (const P2P &o : ops) {
      if (o.ns) MPG_NCCL(R.Send(o.sbuf, o.ns, ncclChar, o.peer, c->comm, s));
      if (o.nr) MPG_NCCL(R.Recv(o.rbuf, o.nr, ncclChar, o.peer, c->comm, s));
    }
    MPG_NCCL(grp.end());
    return MPG_SUCCESS;
  }
  VGroup *g = c->vg;
  const int V = g->V;
  return vg_rendezvous(
      g,
      [&] {
        g->ops[c->rank] = ops;
        g->streams[c->rank] = s;
      },
      [&]() -> int {
        struct Pair { const P2P *snd, *rcv; int src, dst; };
        std::vector<Pair> pairs;
        std::vector<size_t> cur((size_t)V * V, 0);
        auto next_recv = [&](int dst, int src) -> const P2P * {
          const std::vector<P2P> &L = g->ops[dst];
          size_t &k = cur[(size_t)dst * V + src];
          while (k < L.size() && !(L[k].peer == src && L[k].nr)) ++k;
          return k < L.size() ? &L[k++] : nullptr;
        };
        for (int src = 0; src < V; ++src)
          for (const P2P &o : g->ops[src]) {
            if (!o.ns) continue;
            if (o.peer < 0 || o.peer >= V || o.peer == src) {
              mpg_set_error("virtual ranks: rank %d lists a send to rank %d", src, o.peer);
              return MPG_ERR_INVALID_ARG;
            }
            const P2P *r = next_recv(o.peer, src);
            if (!r || r->nr != o.ns) {
              mpg_set_error("virtual ranks: rank %d sends %zu bytes to rank %d, which expects %zu from it: the two schedules disagree", src, o.ns, o.peer,
                            r ? r->nr : (size_t)0);
              return MPG_ERR_INVALID_ARG;
            }
            pairs.push_back({&o, r, src, o.peer});
          }
        for (int dst = 0; dst < V; ++dst)
          for (int src = 0; src < V; ++src)
            if (const P2P *r = next_recv(dst, src)) {
              mpg_set_error("virtual ranks: rank %d expects %zu bytes from rank %d, which sends nothing more", dst, r->nr, src);
              return MPG_ERR_INVALID_ARG;
            }
        if (pairs.empty()) return MPG_SUCCESS;
        // One communicator launches a group on ONE stream.  The group goes on rank 0's stream, with event edges from every
        // other rank's stream (its pack is complete before its bytes are sent) and back (its unpack waits for its receives).
        // (Measured, round 5: handing RCCL each transfer on its own rank's stream inside one group left the receiving ranks'
        // streams unordered against the copy -- their unpack read the receive buffer before the data had landed.)
        hipStream_t lead = g->streams[0];
        EventScope evs;
        MPG_HIP(hipEventCreateWithFlags(&evs.e, hipEventDisableTiming));
        hipEvent_t ev = evs.e;
        for (int q = 1; q < V; ++q)
          if (g->streams[q] != lead) {
            MPG_HIP(hipEventRecord(ev, g->streams[q]));
            MPG_HIP(hipStreamWaitEvent(lead, ev, 0));
          }
        GroupScope grp;
        MPG_NCCL(grp.start());
        for (const Pair &p : pairs) {
          MPG_NCCL(R.Send(p.snd->sbuf, p.snd->ns, ncclChar, 0, g->comm, lead));
          MPG_NCCL(R.Recv(p.rcv->rbuf, p.rcv->nr, ncclChar, 0, g->comm, lead));
        }
        MPG_NCCL(grp.end());
        ++g->n_groups;
        g->n_send_calls += (int64_t)pairs.size();
        g->n_recv_calls += (int64_t)pairs.size();
        MPG_HIP(hipEventRecord(ev, lead));
        for (int q = 1; q < V; ++q)
          if (g->streams[q] != lead) MPG_HIP(hipStreamWaitEvent(g->streams[q], ev, 0));
        return MPG_SUCCESS;
      });
}

extern "C" {

// every rank contributes nbytes from send_host; recv_host gets nranks * nbytes in rank order (small host-side metadata)
int mpg_comm_allgather(mpg_comm c, const void *send_host, int64_t nbytes, void *recv_host) {
  MPG_CHECK_INIT();
  MPG_ARG(c && send_host && recv_host && nbytes > 0, "mpg_comm_allgather: bad argument");
  if (c->is_virtual) return vg_allgather(c, send_host, nbytes, recv_host);
  hipStream_t s = mpg_setup_stream();
  TmpBuf<char> sb, rb;
  int rc;
  if ((rc = sb.alloc((size_t)nbytes)) || (rc = rb.alloc((size_t)nbytes * c->nranks))) return rc;
  MPG_HIP(hipMemcpyAsync(sb.p, send_host, (size_t)nbytes, hipMemcpyHostToDevice, s));
  MPG_NCCL(R.AllGather(sb.p, rb.p, (size_t)nbytes, ncclChar, c->comm, s));
  MPG_HIP(hipMemcpyAsync(recv_host, rb.p, (size_t)nbytes * c->nranks, hipMemcpyDeviceToHost, s));
  return stream_wait_deadline(s, "mpg_comm_allgather");
}

// The schedule as a pure function of every rank's needed ids (diagnostics / tests; mpg_halo_build obtains the same inputs
// with two all-gathers).  send_ids_flat / send_ids_off: compact mode only (offsets inside the own block, per peer).
int mpg_halo_plan_host(int rank, int nranks, int64_t n_cells, int ownership, const int64_t *n_needed, const int32_t *const *needed, int *mode,
                       int64_t *n_local, int64_t *own, int64_t *base, int64_t *own_pos, int64_t *send_count, int64_t *send_a, int64_t *recv_a,
                       int64_t *recv_b, int32_t *send_ids_flat, int64_t send_ids_cap, int64_t *send_ids_off) {
  MPG_ARG(nranks >= 1 && rank >= 0 && rank < nranks && n_cells >= 0 && n_needed && needed, "mpg_halo_plan_host: bad argument");
  std::vector<Vote> votes(nranks);
  for (int q = 0; q < nranks; ++q) votes[q] = make_vote(needed[q], n_needed[q], n_cells, nranks, q);
  HaloPlan p;
  p.rank = rank;
  p.nranks = nranks;
  std::vector<std::pair<int64_t, int64_t>> blocks;
  plan_blocks(nranks, n_cells, ownership, votes, &p.mode, blocks);
  if (p.mode == 0) plan_range(p, votes, blocks);
  else plan_compact(p, blocks, n_needed, needed);
  if (mode) *mode = p.mode;
  if (n_local) *n_local = p.n_local;
  if (own) { own[0] = p.own0; own[1] = p.own1; }
  if (base) *base = p.base;
  if (own_pos) { own_pos[0] = p.own_pos0; own_pos[1] = p.own_pos1; }
  int64_t off = 0;
  for (int q = 0; q < nranks; ++q) {
    if (send_count) send_count[q] = p.send_count(q);
    if (send_a) send_a[q] = p.mode == 0 ? p.send_a[q] : -1;
    if (recv_a) recv_a[q] = p.recv_a[q];
    if (recv_b) recv_b[q] = p.recv_b[q];
    if (send_ids_off) send_ids_off[q] = off;
    if (p.mode == 1) {
      if (send_ids_flat) {
        MPG_ARG(off + (int64_t)p.send_ids[q].size() <= send_ids_cap, "mpg_halo_plan_host: send_ids_flat too small");
        std::copy(p.send_ids[q].begin(), p.send_ids[q].end(), send_ids_flat + off);
      }
      off += (int64_t)p.send_ids[q].size();
    }
  }
  if (send_ids_off) send_ids_off[nranks] = off;
  return MPG_SUCCESS;
}

// The owned-mode schedule as a pure function (tests, diagnostics): needed[q] / owned[q] = rank q's sorted unique ids.  Outputs for
// `rank`: n_local; per peer (self included) send_off[q] .. send_off[q + 1] into send_flat = offsets into the rank's owned list, in the
// order sent; recv_off / recv_flat = destination positions in its local space, in the order received.  Returns MPG_ERR_INVALID_ARG
// when a needed cell is owned by nobody.
int mpg_halo_plan_owned_host(int rank, int nranks, const int64_t *n_needed, const int32_t *const *needed, const int64_t *n_owned,
                             const int32_t *const *owned, int64_t *n_local, int32_t *send_flat, int64_t send_cap, int64_t *send_off, int32_t *recv_flat,
                             int64_t recv_cap, int64_t *recv_off) {
  MPG_ARG(nranks >= 1 && rank >= 0 && rank < nranks && n_needed && needed && n_owned && owned && send_off && recv_off, "mpg_halo_plan_owned_host: bad argument");
  HaloPlan p;
  p.rank = rank;
  p.nranks = nranks;
  const int64_t orphan = plan_owned(p, n_needed, needed, n_owned, owned);
  if (orphan >= 0) {
    mpg_set_error("mpg_halo_plan_owned_host: rank %d needs cell %lld, which no rank owns", rank, (long long)orphan);
    return MPG_ERR_INVALID_ARG;
  }
  if (n_local) *n_local = p.n_local;
  int64_t so = 0, ro = 0;
  for (int q = 0; q < nranks; ++q) {
    send_off[q] = so;
    recv_off[q] = ro;
    MPG_ARG(so + (int64_t)p.send_ids[q].size() <= send_cap && ro + (int64_t)p.recv_ids[q].size() <= recv_cap, "mpg_halo_plan_owned_host: output too small");
    if (send_flat) std::copy(p.send_ids[q].begin(), p.send_ids[q].end(), send_flat + so);
    if (recv_flat) std::copy(p.recv_ids[q].begin(), p.recv_ids[q].end(), recv_flat + ro);
    so += (int64_t)p.send_ids[q].size();
    ro += (int64_t)p.recv_ids[q].size();
  }
  send_off[nranks] = so;
  recv_off[nranks] = ro;
  return MPG_SUCCESS;
}

// the common end of the two builds: the handle leaves the Store cache and is re-indexed in place (as mpg_handle_rebase /
// mpg_handle_localize do), the packed buffers' offsets are laid out, the index lists go to the device.  Frees H on failure.
static int halo_finish(mpg_halo_s *H, mpg_handle_s *h, std::vector<int32_t> &ids, hipStream_t s) {
  HaloPlan &p = H->plan;
  const int world = p.nranks;
  int rc;
  mpg_cache_detach(h);
  h->free_tile_lists();
  h->lf_choice = h->cf_choice = 0;
  if (p.mode == 0) rc = mpg_k_rebase(h, p.base, p.n_local, s);
  else rc = mpg_k_unique_sources(h, ids, true, s);
  if (rc) { delete H; return rc; }
  H->soff.assign(world + 1, 0);
  H->roff.assign(world + 1, 0);
  H->ids_off.assign(world + 1, 0);
  H->rids_off.assign(world + 1, 0);
  for (int q = 0; q < world; ++q) {
    H->soff[q + 1] = H->soff[q] + p.send_count(q);
    H->roff[q + 1] = H->roff[q] + p.recv_count(q);
    H->ids_off[q + 1] = H->ids_off[q] + (p.mode >= 1 ? (int64_t)p.send_ids[q].size() : 0);
    H->rids_off[q + 1] = H->rids_off[q] + (p.mode == 2 ? (int64_t)p.recv_ids[q].size() : 0);
  }
  H->stot = H->soff[world];
  H->rtot = H->roff[world];
  auto upload = [&](DevBuf<int32_t> &dev, const std::vector<std::vector<int32_t>> &lists, const std::vector<int64_t> &off) -> int {
    if (off[world] == 0) return MPG_SUCCESS;
    int r = dev.alloc((size_t)off[world]);
    if (r) return r;
    for (int q = 0; q < world; ++q)
      if (!lists[q].empty())
        if (hipMemcpy(dev.p + off[q], lists[q].data(), sizeof(int32_t) * lists[q].size(), hipMemcpyHostToDevice) != hipSuccess) {
          mpg_set_error("mpg_halo_build: uploading the index lists failed");
          return MPG_ERR_HIP;
        }
    return MPG_SUCCESS;
  };
  if (p.mode >= 1 && (rc = upload(H->send_ids_dev, p.send_ids, H->ids_off))) { mpg_halo_destroy(H); return rc; }
  if (p.mode == 2 && (rc = upload(H->recv_ids_dev, p.recv_ids, H->rids_off))) { mpg_halo_destroy(H); return rc; }
  return MPG_SUCCESS;
}

int mpg_halo_build(mpg_comm c, mpg_handle h, int64_t n_cells, int ownership, mpg_halo *out) {
  MPG_CHECK_INIT();
  MPG_ARG(c && h && out && n_cells > 0 && n_cells < 0x7fffffff, "mpg_halo_build: bad argument");
  MPG_ARG(!h->localized && h->n_pole == 0, "mpg_halo_build: the handle was re-indexed already, or carries pole terms");
  MPG_ARG(h->refcount <= 1, "mpg_halo_build: the handle is shared; re-indexing it in place would corrupt the other holder's indices");
  MPG_ARG(!mpg_handle_is_windowed(h), "mpg_halo_build: the handle's mesh has a source window (mpg_mesh_set_source_window): its indices are "
                                      "window-relative; a windowed mesh and a halo exchange are alternatives -- reset the window first");
  hipStream_t s = mpg_setup_stream();
  // virtual ranks (rehearsal): the threads' device work on the shared set-up stream and the Store cache runs one at a time;
  // the lock is dropped around every collective step
  std::unique_lock<std::mutex> work;
  if (c->is_virtual) work = std::unique_lock<std::mutex>(c->vg->work_mu);
  auto allgather = [&](const void *snd, int64_t nb, void *rcv) {
    if (work.owns_lock()) work.unlock();
    const int r = mpg_comm_allgather(c, snd, nb, rcv);
    if (c->is_virtual) work.lock();
    return r;
  };
  std::vector<int32_t> ids;
  int rc = mpg_k_unique_sources(h, ids, false, s);
  if (rc) return rc;
  const int world = c->nranks, rank = c->rank;
  Vote mine = make_vote(ids.data(), (int64_t)ids.size(), n_cells, world, rank);
  std::vector<Vote> votes(world);
  int64_t cnt = (int64_t)ids.size();
  std::vector<int64_t> counts(world);
  if (world > 1) {
    if ((rc = allgather(&mine, sizeof(Vote), votes.data()))) return rc;
    if ((rc = allgather(&cnt, sizeof(int64_t), counts.data()))) return rc;
  } else {
    votes[0] = mine;
    counts[0] = cnt;
  }
  mpg_halo_s *H = new mpg_halo_s();
  H->comm = c;
  HaloPlan &p = H->plan;
  p.rank = rank;
  p.nranks = world;
  std::vector<std::pair<int64_t, int64_t>> blocks;
  plan_blocks(world, n_cells, ownership, votes, &p.mode, blocks);
  if (p.mode == 0) {
    plan_range(p, votes, blocks);
  } else {   // arbitrary numbering: every rank's id list travels once
    const int64_t mx = *std::max_element(counts.begin(), counts.end());
    std::vector<int32_t> padded((size_t)std::max<int64_t>(mx, 1), 0x7fffffff), all((size_t)std::max<int64_t>(mx, 1) * world);
    std::copy(ids.begin(), ids.end(), padded.begin());
    if (world > 1) {
      if ((rc = allgather(padded.data(), (int64_t)padded.size() * 4, all.data()))) { delete H; return rc; }
    } else {
      all = padded;
    }
    std::vector<const int32_t *> lists(world);
    for (int q = 0; q < world; ++q) lists[q] = all.data() + (size_t)q * padded.size();
    plan_compact(p, blocks, counts.data(), lists.data());
  }
  rc = halo_finish(H, h, ids, s);
  if (rc) return rc;
  *out = H;
  return MPG_SUCCESS;
}

// The caller's own partition of the source cells (owned mode): owned_ids_host = this rank's sorted unique ids (any shape; the ranks'
// lists must be disjoint, and every cell some rank's rows reference must be in one of them).
int mpg_halo_build_owned(mpg_comm c, mpg_handle h, int64_t n_cells, const int32_t *owned_ids_host, int64_t n_owned, mpg_halo *out) {
  MPG_CHECK_INIT();
  MPG_ARG(c && h && out && n_cells > 0 && n_cells < 0x7fffffff && n_owned >= 0 && (owned_ids_host || n_owned == 0), "mpg_halo_build_owned: bad argument");
  MPG_ARG(!h->localized && h->n_pole == 0, "mpg_halo_build_owned: the handle was re-indexed already, or carries pole terms");
  MPG_ARG(h->refcount <= 1, "mpg_halo_build_owned: the handle is shared; re-indexing it in place would corrupt the other holder's indices");
  MPG_ARG(!mpg_handle_is_windowed(h), "mpg_halo_build_owned: the handle's mesh has a source window; reset the window first");
  for (int64_t i = 0; i < n_owned; ++i)
    MPG_ARG(owned_ids_host[i] >= 0 && owned_ids_host[i] < n_cells && (i == 0 || owned_ids_host[i] > owned_ids_host[i - 1]),
            "mpg_halo_build_owned: owned_ids must be sorted, unique and within [0, n_cells)");
  hipStream_t s = mpg_setup_stream();
  std::unique_lock<std::mutex> work;
  if (c->is_virtual) work = std::unique_lock<std::mutex>(c->vg->work_mu);
  auto allgather = [&](const void *snd, int64_t nb, void *rcv) {
    if (work.owns_lock()) work.unlock();
    const int r = mpg_comm_allgather(c, snd, nb, rcv);
    if (c->is_virtual) work.lock();
    return r;
  };
  std::vector<int32_t> ids;
  int rc = mpg_k_unique_sources(h, ids, false, s);
  if (rc) return rc;
  const int world = c->nranks, rank = c->rank;
  int64_t mine[2] = {(int64_t)ids.size(), n_owned};
  std::vector<int64_t> cnt(2 * (size_t)world);
  if (world > 1) {
    if ((rc = allgather(mine, sizeof(mine), cnt.data()))) return rc;
  } else {
    cnt[0] = mine[0];
    cnt[1] = mine[1];
  }
  int64_t mxn = 1, mxo = 1;
  std::vector<int64_t> n_needed(world), n_own(world);
  for (int q = 0; q < world; ++q) {
    n_needed[q] = cnt[2 * q];
    n_own[q] = cnt[2 * q + 1];
    mxn = std::max(mxn, n_needed[q]);
    mxo = std::max(mxo, n_own[q]);
  }
  // every rank's needed and owned lists travel once (one all-gather of both, padded to the longest)
  const size_t stride = (size_t)(mxn + mxo);
  std::vector<int32_t> padded(stride, 0x7fffffff), all(stride * world);
  std::copy(ids.begin(), ids.end(), padded.begin());
  std::copy(owned_ids_host, owned_ids_host + n_owned, padded.begin() + mxn);
  if (world > 1) {
    if ((rc = allgather(padded.data(), (int64_t)stride * 4, all.data()))) return rc;
  } else {
    all = padded;
  }
  std::vector<const int32_t *> nl(world), ol(world);
  for (int q = 0; q < world; ++q) {
    nl[q] = all.data() + (size_t)q * stride;
    ol[q] = nl[q] + mxn;
  }
  {   // the partition must be one: no cell owned twice
    std::vector<int64_t> at(world, 0);
    int32_t last = -1;
    for (;;) {
      int best = -1;
      for (int q = 0; q < world; ++q)
        if (at[q] < n_own[q] && (best < 0 || ol[q][at[q]] < ol[best][at[best]])) best = q;
      if (best < 0) break;
      const int32_t v = ol[best][at[best]++];
      if (v == last) {
        mpg_set_error("mpg_halo_build_owned: cell %d is owned by more than one rank", v);
        return MPG_ERR_INVALID_ARG;
      }
      last = v;
    }
  }
  {   // ... and cover what the rows reference: EVERY rank checks every rank's needs (all lists are here), so that all of them refuse
      // together -- a rank that went on alone would wait in its first exchange for a peer that has already given up
    std::vector<uint8_t> has((size_t)n_cells, 0);
    for (int q = 0; q < world; ++q)
      for (int64_t i = 0; i < n_own[q]; ++i) has[(size_t)ol[q][i]] = 1;
    for (int q = 0; q < world; ++q)
      for (int64_t i = 0; i < n_needed[q]; ++i)
        if (nl[q][i] < 0 || nl[q][i] >= n_cells || !has[(size_t)nl[q][i]]) {
          mpg_set_error("mpg_halo_build_owned: rank %d's rows reference cell %d, which no rank owns", q, nl[q][i]);
          return MPG_ERR_INVALID_ARG;
        }
  }
  mpg_halo_s *H = new mpg_halo_s();
  H->comm = c;
  HaloPlan &p = H->plan;
  p.rank = rank;
  p.nranks = world;
  const int64_t orphan = plan_owned(p, n_needed.data(), nl.data(), n_own.data(), ol.data());
  if (orphan >= 0) {   // (unreachable after the check above; kept as the plan's own guard)
    mpg_set_error("mpg_halo_build_owned: rank %d's rows reference cell %lld, which no rank owns", rank, (long long)orphan);
    delete H;
    return MPG_ERR_INVALID_ARG;
  }
  rc = halo_finish(H, h, ids, s);
  if (rc) return rc;
  *out = H;
  return MPG_SUCCESS;
}

// this rank's sorted owned ids as the schedule holds them are the caller's own; what it needs back: nothing.  (mpg_halo_info reports
// mode 2, own = {0, n_owned}: own_dev of mpg_halo_exchange_dev is [nrows][own_ld >= n_owned] in the order of owned_ids.)

int mpg_halo_info(mpg_halo H, int *mode, int64_t *n_local, int64_t *own, int64_t *base, int64_t *own_pos, int64_t *sent_per_row,
                  int64_t *received_per_row) {
  MPG_ARG(H, "mpg_halo_info: NULL schedule");
  const HaloPlan &p = H->plan;
  if (mode) *mode = p.mode;
  if (n_local) *n_local = p.n_local;
  if (own) { own[0] = p.own0; own[1] = p.own1; }
  if (base) *base = p.base;
  if (own_pos) { own_pos[0] = p.own_pos0; own_pos[1] = p.own_pos1; }
  if (sent_per_row) *sent_per_row = H->stot - (p.mode >= 1 ? p.send_count(p.rank) : 0);
  if (received_per_row) *received_per_row = H->rtot - (p.mode >= 1 ? p.recv_count(p.rank) : 0);
  return MPG_SUCCESS;
}

int mpg_halo_destroy(mpg_halo H) {
  if (!H) return MPG_SUCCESS;
  H->send_ids_dev.free();
  H->recv_ids_dev.free();
  H->sendbuf.free();
  H->recvbuf.free();
  delete H;
  return MPG_SUCCESS;
}

// own_dev: nrows rows of this rank's own block (own[1] - own[0] elements used, row stride own_ld elements); local_dev: nrows
// rows of n_local elements, filled in place.  Range mode: own_dev may point INTO local_dev (own data in place at own_pos[0],
// own_ld = n_local) -- then only the neighbours' strips move.  elem_bytes: 4 or 8 for cell-fast slabs ([level][cell]: nrows =
// nfields * nlev), or the bytes of one whole source ROW for slabs in file order ([cell][level]: nrows = nfields, elem_bytes =
// nlev * 4 or nlev * 8 -- in range form a neighbour's strip of such a slab is ONE contiguous byte range per field).
// Enqueued on hip_stream.
int mpg_halo_exchange_dev(mpg_halo H, const void *own_dev, int64_t own_ld, void *local_dev, int nrows, int elem_bytes, void *hip_stream) {
  MPG_CHECK_INIT();
  MPG_ARG(H && nrows >= 1 && elem_bytes >= 4 && elem_bytes % 4 == 0 && elem_bytes <= (1 << 20), "mpg_halo_exchange_dev: bad argument");
  const HaloPlan &p = H->plan;
  // (a rank whose rows reference nothing has an EMPTY local space and may pass NULL for it -- it still takes part: the others need its cells)
  MPG_ARG(local_dev || p.n_local == 0, "mpg_halo_exchange_dev: local_dev is NULL");
  MPG_ARG(own_dev || p.own1 == p.own0, "mpg_halo_exchange_dev: own_dev is NULL");
  hipStream_t s = (hipStream_t)hip_stream;
  const int world = p.nranks, rank = p.rank;
  const size_t es = (size_t)elem_bytes, need = (size_t)nrows * es * (size_t)std::max(H->stot, H->rtot);
  if ((int64_t)need > H->buf_bytes) {   // first use for this batch size: the packed buffers (kept afterwards)
    H->sendbuf.free();
    H->recvbuf.free();
    int rc;
    if ((rc = H->sendbuf.alloc(need + 16)) || (rc = H->recvbuf.alloc(need + 16))) return rc;
    H->buf_bytes = (int64_t)need;
  }
  const size_t n_local = (size_t)p.n_local;
  // 1. pack what each peer wants from the own block: [nrows][count] per peer
  for (int q = 0; q < world; ++q) {
    const int64_t n = p.send_count(q);
    if (!n) continue;
    char *dst = H->sendbuf.p + (size_t)nrows * es * (size_t)H->soff[q];
    if (p.mode == 0) {
      MPG_HIP(hipMemcpy2DAsync(dst, (size_t)n * es, (const char *)own_dev + (size_t)p.send_a[q] * es, (size_t)own_ld * es, (size_t)n * es, (size_t)nrows,
                               hipMemcpyDeviceToDevice, s));
    } else {
      const int rc = pack_ids(own_dev, own_ld, nrows, elem_bytes, H->send_ids_dev.p + H->ids_off[q], n, dst, s);
      if (rc) return rc;
    }
  }
  // 2. one grouped exchange: ncclSend / ncclRecv with every peer that has something (point-to-point over xGMI)
  if (world > 1) {
    std::vector<P2P> ops;
    for (int q = 0; q < world; ++q) {
      if (q == rank) continue;
      const size_t ns = (size_t)p.send_count(q) * nrows * es, nr = (size_t)p.recv_count(q) * nrows * es;
      if (ns || nr) ops.push_back({q, H->sendbuf.p + (size_t)nrows * es * (size_t)H->soff[q], ns, H->recvbuf.p + (size_t)nrows * es * (size_t)H->roff[q], nr});
    }
    const int rc = comm_exchange(H->comm, ops, s);
    if (rc) return rc;
  }
  // range form with the own block held elsewhere: it goes to its place in the local space first (in place when own_dev
  // already IS that place)
  if (p.mode == 0 && p.own1 > p.own0) {
    char *home = (char *)local_dev + (size_t)p.own_pos0 * es;
    const bool in_place = (const char *)own_dev == home && (size_t)own_ld == n_local;
    const char *lo = (const char *)local_dev, *hi = lo + (size_t)nrows * n_local * es;
    const char *olo = (const char *)own_dev, *ohi = olo + ((size_t)(nrows - 1) * (size_t)own_ld + (size_t)(p.own1 - p.own0)) * es;
    MPG_ARG(in_place || ohi <= lo || olo >= hi, "mpg_halo_exchange_dev: own_dev overlaps local_dev without being the in-place view "
                                                 "(local_dev + own_pos[0] elements, own_ld = n_local)");
    if (!in_place)
      MPG_HIP(hipMemcpy2DAsync(home, n_local * es, own_dev, (size_t)own_ld * es, (size_t)(p.own1 - p.own0) * es, (size_t)nrows, hipMemcpyDeviceToDevice, s));
  }
  // 3. unpack into the local index space (the rank's own share of a compact schedule never leaves the device)
  for (int q = 0; q < world; ++q) {
    const int64_t n = p.recv_count(q);
    if (!n) continue;
    const char *src = q == rank ? H->sendbuf.p + (size_t)nrows * es * (size_t)H->soff[q] : H->recvbuf.p + (size_t)nrows * es * (size_t)H->roff[q];
    if (p.mode == 2) {   // a peer's cells sit anywhere among the sorted needed ids: scattered by position
      const int rc = unpack_ids(src, n, nrows, elem_bytes, H->recv_ids_dev.p + H->rids_off[q], (int64_t)n_local, local_dev, s);
      if (rc) return rc;
      continue;
    }
    MPG_HIP(hipMemcpy2DAsync((char *)local_dev + (size_t)p.recv_a[q] * es, n_local * es, src, (size_t)n * es, (size_t)n * es, (size_t)nrows,
                             hipMemcpyDeviceToDevice, s));
  }
  return MPG_SUCCESS;
}

// dst[k][i] = src[k * ld + ids[i]] for nrows rows of elements of elem_bytes bytes (a multiple of 4): the pack step of the compact
// halo form for hosts that run the exchange themselves (mpassit_amd/dist.py over torch.distributed); device pointers
int mpg_pack_rows_dev(const void *src_dev, int64_t ld, int nrows, int elem_bytes, const int32_t *ids_dev, int64_t n_ids, void *dst_dev, void *hip_stream) {
  MPG_CHECK_INIT();
  MPG_ARG(src_dev && dst_dev && ids_dev && nrows >= 1 && n_ids >= 0 && ld >= 0 && elem_bytes >= 4 && elem_bytes % 4 == 0 && elem_bytes <= (1 << 20),
          "mpg_pack_rows_dev: bad argument");
  return pack_ids(src_dev, ld, nrows, elem_bytes, ids_dev, n_ids, dst_dev, (hipStream_t)hip_stream);
}

// ESMF_FieldGather (write_data.F90:1006-1453): every rank holds rows [j0, j1) of an [nlev][ny][nx] field as
// [nlev][j1 - j0][nx]; `root` receives the whole field.  Row blocks may be any partition of 0 .. ny (j0 / j1 of every rank
// are exchanged inside).  dst_dev is read on the root only.
int mpg_gather_rows(mpg_comm c, const void *rows_dev, int64_t j0, int64_t j1, int64_t nx, int64_t ny, int nlev, int elem_bytes, void *dst_dev,
                    int root, void *hip_stream) {
  MPG_CHECK_INIT();
  MPG_ARG(c && nx > 0 && ny > 0 && nlev >= 1 && j0 >= 0 && j1 >= j0 && j1 <= ny && elem_bytes > 0 && root >= 0 && root < c->nranks,
          "mpg_gather_rows: bad argument");
  MPG_ARG(rows_dev || j1 == j0, "mpg_gather_rows: rows_dev is NULL");
  MPG_ARG(c->rank != root || dst_dev, "mpg_gather_rows: dst_dev is NULL on the root");
  hipStream_t s = (hipStream_t)hip_stream;
  const int world = c->nranks;
  std::vector<int64_t> blk(2 * (size_t)world);
  int64_t mine[2] = {j0, j1};
  if (world > 1) {
    int rc = mpg_comm_allgather(c, mine, sizeof(mine), blk.data());
    if (rc) return rc;
  } else {
    blk[0] = j0;
    blk[1] = j1;
  }
  const size_t es = (size_t)elem_bytes, row = (size_t)nx * es;
  if (c->rank == root && j1 > j0)   // the root's own rows: a strided device copy
    MPG_HIP(hipMemcpy2DAsync((char *)dst_dev + (size_t)j0 * row, (size_t)ny * row, rows_dev, (size_t)(j1 - j0) * row, (size_t)(j1 - j0) * row,
                             (size_t)nlev, hipMemcpyDeviceToDevice, s));
  if (world > 1) {   // one group: a rank's block travels as nlev segments, each to its level of the root's field
    std::vector<P2P> ops;
    for (int k = 0; k < nlev; ++k) {
      if (c->rank != root) {
        if (j1 > j0) ops.push_back({root, (const char *)rows_dev + (size_t)k * (size_t)(j1 - j0) * row, (size_t)(j1 - j0) * row, nullptr, 0});
      } else {
        for (int q = 0; q < world; ++q) {
          const int64_t a = blk[2 * q], b = blk[2 * q + 1];
          if (q == root || b <= a) continue;
          ops.push_back({q, nullptr, 0, (char *)dst_dev + ((size_t)k * (size_t)ny + (size_t)a) * row, (size_t)(b - a) * row});
        }
      }
    }
    return comm_exchange(c, ops, s);
  }
  return MPG_SUCCESS;
}

}  // extern "C"

// mpg_init loads this translation unit's code object ahead of its first launch (mpg_api.hip: warm_modules)
const void *mpg_anchor_mpg_comm() { return (const void *)&k_pack_ids<uint32_t>; }
