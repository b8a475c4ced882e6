/*
 * vs_node.hip -- one batch over the GPUs of one node, in C (north star: "host code stays in C";
 * SURVEY.md section 8b/8e).
 *
 * Utterances are independent -- all carried state of the reference is per utterance
 * (flowgen_shimmer.c:121-122, vowel_new.c:90) -- so a batch is cut into contiguous blocks of
 * lanes, one block ("shard") per device, with NO data-path collective.  A lane's draw stream is
 * keyed by the seed in its own vs_lane record, so the result does not depend on the placement:
 * N shards give byte for byte what one device gives.
 *
 * The one exchange the north star names is the delivery of the finished PCM:
 *   vs_node_synth_gather  into device 0's HBM.  Every shard synthesises its block in chunks of
 *                         VS_NODE_CHUNK utterances; a finished chunk is copied to its place in the
 *                         root buffer by a peer DMA (hipMemcpy2DAsync between devices = one stream
 *                         per xGMI link into the root, the pattern SURVEY.md section 5 asks for:
 *                         7 concurrent point-to-point transfers, no ring) on the shard's copy
 *                         stream while its next chunk is being synthesised on its compute stream.
 *                         The root's own shard is synthesised in place.
 *   vs_node_synth_rows    to the host: every shard runs the pipeline of vs_synth_rows() on its own
 *                         device and PCIe link; the callback sees global row numbers.
 * A device may appear several times in the device list ("logical shards"): that is how the
 * N-device path is tested on a box with one GPU.
 *
 * Transport of the gather (vs_node_set_transport):
 *   VS_NODE_TRANSPORT_PEER  peer DMA as above (the default).  A shard whose device cannot reach the
 *                           root by peer access is reported as VS_NODE_LINK_STAGED by vs_node_link():
 *                           the runtime then carries its copies through host memory.
 *   VS_NODE_TRANSPORT_RCCL  one RCCL communicator over the node's (distinct) devices, owned by the
 *                           node object: a finished chunk leaves by ncclSend on the shard's copy
 *                           stream, the root posts the matching ncclRecv of every peer's chunk k as
 *                           ONE group on ONE stream -- RCCL fuses the point-to-point operations of a
 *                           group into one launch whose channels run the seven transfers side by side
 *                           (one per xGMI link, no ring); streams do not add to that.  librccl is
 *                           opened with dlopen() when this transport is chosen, so that the two
 *                           drop-in programs do not pay for loading it.
 *
 * The exchange is ALL OR NOTHING (vs_node_synth_gather): every shard first creates the plans of all
 * of its chunks; the shard threads then meet, and only if every one of them succeeded does anybody
 * enqueue a kernel, a copy, a send or a receive -- an unmatched ncclSend / ncclRecv never completes,
 * and a stream that holds one can never be waited for.  A failure AFTER that point (a launch or a
 * send refused) aborts the node's communicators (ncclCommAbort, which ends the kernels of the
 * operations in flight), every thread stops at its next chunk, the call returns the error and the
 * node is back on the peer transport.  Which rows travel in which round is plain C without a device
 * in it (vs_gather_round, csrc/vs_host.c), walked by sender and receiver alike.
 *
 * One host thread per shard (a vs_ctx is used by one thread at a time).  The multi-PROCESS form
 * of the same scheme -- one rank per GPU under torch.distributed.run, RCCL send/recv of the
 * chunks -- is what bench.py --gpus N runs; see voice_synth_amd/dist.py.
 */
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <new>
#include <thread>
#include <vector>

#include "vs_internal.h"

#define VS_NODE_CHUNK 16384

/* the handful of RCCL entry points the gather uses, resolved from librccl at run time (the types are
 * RCCL's: ncclComm_t is an opaque pointer, ncclResult_t and ncclDataType_t are ints, ncclInt8 == 0) */
typedef void *vs_nccl_comm;
struct VsRccl {
  void *lib;
  int (*CommInitAll)(vs_nccl_comm *, int, const int *);
  int (*CommDestroy)(vs_nccl_comm);
  int (*CommAbort)(vs_nccl_comm);
  int (*Send)(const void *, size_t, int, int, vs_nccl_comm, hipStream_t);
  int (*Recv)(void *, size_t, int, int, vs_nccl_comm, hipStream_t);
  int (*GroupStart)(void);
  int (*GroupEnd)(void);
};
#define VS_NCCL_INT8 0

struct vs_node {
  std::vector<vs_ctx *> ctx;   /* one per shard */
  std::vector<int> device;
  std::vector<hipStream_t> compute, copy;
  std::vector<hipEvent_t> ev_done[2], ev_copied[2];
  std::vector<int> link;       /* VS_NODE_LINK_* of every shard */
  std::vector<int> base_link;  /* ... as found at creation (self / peer / staged): what the peer transport uses */
  int transport;               /* VS_NODE_TRANSPORT_* */
  VsRccl rccl;
  std::vector<vs_nccl_comm> comm;      /* one per shard (rank = shard), RCCL transport only */
  std::vector<hipStream_t> recv;       /* on the root device: THE stream the root's receive groups are posted on */
  int last_rccl_error;
};

extern "C" int vs_node_create(const int *devices, int n_shards, vs_node **out)
{
  if (!devices || n_shards <= 0 || n_shards > 64 || !out) return VS_ERR_ARG;
  *out = nullptr;
  vs_node *nd = new (std::nothrow) vs_node();
  if (!nd) return VS_ERR_NOMEM;
  nd->transport = VS_NODE_TRANSPORT_PEER;
  memset(&nd->rccl, 0, sizeof(nd->rccl));
  nd->last_rccl_error = 0;
  int rc = VS_OK;
  for (int s = 0; s < n_shards && rc == VS_OK; s++) {
    vs_ctx *c = nullptr;
    rc = vs_ctx_create(devices[s], &c);
    if (rc != VS_OK) break;
    nd->ctx.push_back(c);
    nd->device.push_back(devices[s]);
    hipStream_t a = nullptr, b = nullptr;
    hipEvent_t e[4] = {nullptr, nullptr, nullptr, nullptr};
    hipError_t he = hipSetDevice(devices[s]);
    if (he == hipSuccess) he = hipStreamCreateWithFlags(&a, hipStreamNonBlocking);
    if (he == hipSuccess) he = hipStreamCreateWithFlags(&b, hipStreamNonBlocking);
    for (int k = 0; k < 4 && he == hipSuccess; k++) he = hipEventCreateWithFlags(&e[k], hipEventDisableTiming);
    nd->compute.push_back(a);
    nd->copy.push_back(b);
    nd->ev_done[0].push_back(e[0]);
    nd->ev_done[1].push_back(e[1]);
    nd->ev_copied[0].push_back(e[2]);
    nd->ev_copied[1].push_back(e[3]);
    if (he != hipSuccess) rc = VS_ERR_HIP;
    /* how this shard's PCM reaches the root: in place, by peer DMA, or -- when the device cannot
     * reach the root by peer access -- through host memory, which vs_node_link() says out loud */
    int link = VS_NODE_LINK_SELF;
    if (rc == VS_OK && devices[s] != devices[0]) {
      int can = 0;
      link = VS_NODE_LINK_STAGED;
      if (hipDeviceCanAccessPeer(&can, devices[s], devices[0]) == hipSuccess && can) {
        he = hipDeviceEnablePeerAccess(devices[0], 0);
        if (he != hipSuccess && he != hipErrorPeerAccessAlreadyEnabled) rc = VS_ERR_HIP;
        else link = VS_NODE_LINK_PEER;
        (void)hipGetLastError();
      }
    }
    nd->link.push_back(link);
    nd->base_link.push_back(link);
  }
  if (rc != VS_OK) {
    vs_node_destroy(nd);
    return rc;
  }
  *out = nd;
  return VS_OK;
}

static void vs_node_drop_rccl(vs_node *nd)
{
  for (size_t s = 0; s < nd->comm.size(); s++) {
    if (nd->comm[s] && nd->rccl.CommDestroy) {
      (void)hipSetDevice(nd->device[s]);
      (void)nd->rccl.CommDestroy(nd->comm[s]);
    }
  }
  nd->comm.clear();
  if (!nd->recv.empty()) (void)hipSetDevice(nd->device[0]);
  for (hipStream_t st : nd->recv)
    if (st) (void)hipStreamDestroy(st);
  nd->recv.clear();
  if (nd->rccl.lib) dlclose(nd->rccl.lib);
  memset(&nd->rccl, 0, sizeof(nd->rccl));
}

extern "C" int vs_node_set_transport(vs_node *nd, int transport)
{
  if (!nd || (transport != VS_NODE_TRANSPORT_PEER && transport != VS_NODE_TRANSPORT_RCCL)) return VS_ERR_ARG;
  if (transport == nd->transport) return VS_OK;
  if (transport == VS_NODE_TRANSPORT_PEER) {
    vs_node_drop_rccl(nd);
    nd->transport = transport;
    nd->link = nd->base_link;
    return VS_OK;
  }
  /* RCCL puts one rank on one device: logical shards of one device cannot form a communicator */
  const size_t S = nd->device.size();
  for (size_t a = 0; a < S; a++)
    for (size_t b = a + 1; b < S; b++)
      if (nd->device[a] == nd->device[b]) return VS_ERR_UNSUPPORTED;
  VsRccl &R = nd->rccl;
  /* the RCCL that belongs to the HIP runtime this process runs on: the one next to libamdhip64
   * (a process may hold a second ROCm, e.g. the copy bundled with PyTorch, and RCCL on the wrong HSA
   * runtime finds no device), then whatever the loader finds by name */
  {
    Dl_info info;
    char path[1024];
    if (dladdr((void *)&hipGetDeviceCount, &info) && info.dli_fname) {
      const char *slash = strrchr(info.dli_fname, '/');
      if (slash && (size_t)(slash - info.dli_fname) + 16 < sizeof(path)) {
        const size_t dir = (size_t)(slash - info.dli_fname) + 1;
        memcpy(path, info.dli_fname, dir);
        strcpy(path + dir, "librccl.so.1");
        R.lib = dlopen(path, RTLD_NOW | RTLD_LOCAL);
        if (!R.lib) {
          strcpy(path + dir, "librccl.so");
          R.lib = dlopen(path, RTLD_NOW | RTLD_LOCAL);
        }
      }
    }
  }
  if (!R.lib) R.lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
  if (!R.lib) R.lib = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
  if (!R.lib) return VS_ERR_UNSUPPORTED;
  R.CommInitAll = (int (*)(vs_nccl_comm *, int, const int *))dlsym(R.lib, "ncclCommInitAll");
  R.CommDestroy = (int (*)(vs_nccl_comm))dlsym(R.lib, "ncclCommDestroy");
  R.CommAbort = (int (*)(vs_nccl_comm))dlsym(R.lib, "ncclCommAbort");
  R.Send = (int (*)(const void *, size_t, int, int, vs_nccl_comm, hipStream_t))dlsym(R.lib, "ncclSend");
  R.Recv = (int (*)(void *, size_t, int, int, vs_nccl_comm, hipStream_t))dlsym(R.lib, "ncclRecv");
  R.GroupStart = (int (*)(void))dlsym(R.lib, "ncclGroupStart");
  R.GroupEnd = (int (*)(void))dlsym(R.lib, "ncclGroupEnd");
  if (!R.CommInitAll || !R.CommDestroy || !R.CommAbort || !R.Send || !R.Recv || !R.GroupStart || !R.GroupEnd) {
    vs_node_drop_rccl(nd);
    return VS_ERR_UNSUPPORTED;
  }
  nd->comm.assign(S, nullptr);
  const int e = R.CommInitAll(nd->comm.data(), (int)S, nd->device.data());
  if (e != 0) {
    nd->last_rccl_error = e;
    nd->comm.clear();
    vs_node_drop_rccl(nd);
    return VS_ERR_HIP;
  }
  /* the root posts its receive groups on ONE stream of its own (concurrency inside a group of
   * point-to-point operations comes from RCCL's channels, not from streams) */
  hipError_t he = hipSetDevice(nd->device[0]);
  if (he == hipSuccess) {
    hipStream_t st = nullptr;
    he = hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    nd->recv.push_back(st);
  }
  if (he != hipSuccess) {
    vs_node_drop_rccl(nd);
    return VS_ERR_HIP;
  }
  nd->transport = transport;
  for (size_t s = 1; s < S; s++) nd->link[s] = VS_NODE_LINK_RCCL;
  return VS_OK;
}

extern "C" int vs_node_link(const vs_node *nd, int shard)
{
  if (!nd || shard < 0 || shard >= (int)nd->link.size()) return VS_ERR_ARG;
  return nd->link[(size_t)shard];
}

extern "C" int vs_node_last_rccl_error(const vs_node *nd) { return nd ? nd->last_rccl_error : 0; }

extern "C" void vs_node_destroy(vs_node *nd)
{
  if (!nd) return;
  vs_node_drop_rccl(nd);
  for (size_t s = 0; s < nd->ctx.size(); s++) {
    (void)hipSetDevice(nd->device[s]);
    if (s < nd->compute.size() && nd->compute[s]) (void)hipStreamDestroy(nd->compute[s]);
    if (s < nd->copy.size() && nd->copy[s]) (void)hipStreamDestroy(nd->copy[s]);
    for (int k = 0; k < 2; k++) {
      if (s < nd->ev_done[k].size() && nd->ev_done[k][s]) (void)hipEventDestroy(nd->ev_done[k][s]);
      if (s < nd->ev_copied[k].size() && nd->ev_copied[k][s]) (void)hipEventDestroy(nd->ev_copied[k][s]);
    }
    vs_ctx_destroy(nd->ctx[s]);
  }
  delete nd;
}

extern "C" int vs_node_shards(const vs_node *nd) { return nd ? (int)nd->ctx.size() : 0; }

extern "C" int vs_node_ctx(vs_node *nd, int shard, vs_ctx **ctx)
{
  if (!nd || !ctx || shard < 0 || shard >= (int)nd->ctx.size()) return VS_ERR_ARG;
  *ctx = nd->ctx[(size_t)shard];
  return VS_OK;
}

extern "C" int vs_node_set_arith(vs_node *nd, int arith)
{
  if (!nd) return VS_ERR_ARG;
  for (vs_ctx *c : nd->ctx) {
    const int rc = vs_ctx_set_arith(c, arith);
    if (rc != VS_OK) return rc;
  }
  return VS_OK;
}

/* lanes [lo, hi) of shard s: the cut of csrc/vs_host.c (vs_shard_cut), which the one-process-per-GPU
 * path makes too (voice_synth_amd/configs.py::shard_range) */
static void shard_range(size_t n_lanes, size_t shards, size_t s, size_t *lo, size_t *hi)
{
  (void)vs_shard_cut(n_lanes, (int)shards, (int)s, lo, hi);
}

extern "C" int vs_node_shard_range(const vs_node *nd, size_t n_lanes, int shard, size_t *lo, size_t *hi)
{
  if (!nd || !lo || !hi || shard < 0 || shard >= (int)nd->ctx.size()) return VS_ERR_ARG;
  shard_range(n_lanes, nd->ctx.size(), (size_t)shard, lo, hi);
  return VS_OK;
}

namespace {
/* where the shard threads of one gather meet: between "all of my chunk plans exist" and "the first
 * kernel / copy / send / receive is enqueued" */
struct GatherSync {
  std::mutex m;
  std::condition_variable cv;
  size_t parties = 0, waiting = 0;
  unsigned generation = 0;
  bool ok = true;                    /* cleared by a shard that failed to prepare */
  std::atomic<bool> aborted{false};  /* set by a shard that failed once the exchange had started */
  std::mutex abort_m;

  /* every shard calls this once; returns the verdict of all of them */
  bool arrive(bool mine)
  {
    std::unique_lock<std::mutex> lk(m);
    if (!mine) ok = false;
    const unsigned gen = generation;
    if (++waiting == parties) {
      waiting = 0;
      generation++;
      cv.notify_all();
    } else {
      cv.wait(lk, [&] { return generation != gen; });
    }
    return ok;
  }
};

struct ShardJob {
  vs_node *nd;
  GatherSync *sync;
  size_t s;
  const vs_lane *lanes;
  size_t lo, hi, n_samples, n_total;
  int16_t *root;      /* device pointer on device[0]: int16 [n_lanes][root_pitch] */
  size_t root_pitch;
  int flags;          /* VS_NODE_OVERLAP, VS_NODE_STAGE_ALL */
  int rc;
  double compute_ms;  /* host clock: first launch .. last kernel done */
};

/* A shard failed after the exchange had started: peers may hold sends nobody will receive, the root
 * receives nobody will send.  ncclCommAbort ends the kernels of the operations in flight, so that the
 * streams they sit on can be waited for again; the communicators are gone afterwards (the caller of
 * vs_node_synth_gather drops the transport).  Once per gather, whoever comes first. */
void abort_exchange(vs_node *nd, GatherSync *sync)
{
  std::lock_guard<std::mutex> lk(sync->abort_m);
  if (sync->aborted.exchange(true)) return;
  if (nd->transport != VS_NODE_TRANSPORT_RCCL) return;
  for (size_t p = 0; p < nd->comm.size(); p++) {
    if (nd->comm[p]) {
      (void)nd->rccl.CommAbort(nd->comm[p]);
      nd->comm[p] = nullptr;
    }
  }
}

void shard_gather(ShardJob *j)
{
  vs_node *nd = j->nd;
  GatherSync *sync = j->sync;
  const size_t s = j->s;
  const size_t S = nd->ctx.size();
  vs_ctx *ctx = nd->ctx[s];
  j->rc = VS_OK;
  j->compute_ms = 0.0;
  const bool rccl = nd->transport == VS_NODE_TRANSPORT_RCCL;
  const size_t rows_all = j->hi - j->lo;
  /* the root's own shard is synthesised in place, in ONE launch: nothing travels, so there is
   * nothing to overlap, and a whole shard fills the chip where a chunk fills a quarter of it.  (Its
   * receives are posted behind that launch: the peers' first chunks are not finished before the
   * root's own kernel is either) */
  const bool in_place = (nd->device[s] == nd->device[0]) && !((j->flags & VS_NODE_STAGE_ALL) && !rccl);
  const size_t chunk = in_place ? std::max<size_t>(rows_all, 1) : VS_NODE_CHUNK;
  /* an RCCL message is one contiguous range: chunks are synthesised at pitch n_samples and land in a
   * root buffer of that pitch (checked by the caller) */
  const size_t pitch = rccl ? j->n_samples : ((j->n_samples + 7) & ~(size_t)7);
  VsPool &P = ctx->pool;
  hipStream_t saved = ctx->stream;
  std::vector<vs_plan *> plans;
  struct Chunk { size_t row0, rows; };
  std::vector<Chunk> chunks;

  /* ---- prepare: buffers and the plans of ALL chunks; nothing is enqueued yet ---- */
  if (hipSetDevice(nd->device[s]) != hipSuccess) j->rc = VS_ERR_HIP;
  if (j->rc == VS_OK && rows_all > 0) {
    for (size_t round = 0;; round++) {
      size_t row0 = 0, rows = 0;
      if (in_place) {
        if (round > 0) break;
        row0 = j->lo;
        rows = rows_all;
      } else if (vs_gather_round(j->n_total, (int)S, (int)s, VS_NODE_CHUNK, round, &row0, &rows) != VS_OK || rows == 0) {
        break;
      }
      chunks.push_back({row0, rows});
    }
    if (!in_place) {
      for (int k = 0; k < 2 && j->rc == VS_OK && (k == 0 || chunks.size() > 1); k++)
        j->rc = vs_pool_device(ctx, &P.d_out[k], &P.d_out_bytes[k], std::min(rows_all, chunk) * pitch * sizeof(int16_t));
    }
    ctx->stream = nd->compute[s];
    for (size_t c = 0; c < chunks.size() && j->rc == VS_OK; c++) {
      vs_plan *plan = nullptr;
      j->rc = vs_plan_create_impl(ctx, j->lanes + chunks[c].row0, chunks[c].rows, j->n_samples, VS_PLAN_POOL_SCRATCH, &plan);
      if (j->rc == VS_OK) plans.push_back(plan);
    }
    if (j->rc == VS_OK && ctx->tuning.fault == VS_FAULT_SHARD_PREPARE) j->rc = VS_ERR_INTERNAL; /* tests */
  }
  /* ---- all or nothing: one shard that cannot go on keeps every shard from starting ---- */
  const bool go = sync->arrive(j->rc == VS_OK);

  const auto t0 = std::chrono::steady_clock::now();
  bool used[2] = {false, false};
  int k = 0;
  for (size_t c = 0; go && c < chunks.size() && j->rc == VS_OK && !sync->aborted.load(); c++, k ^= 1) {
    const size_t r0 = chunks[c].row0, rows = chunks[c].rows;
    int16_t *dst = j->root + r0 * j->root_pitch;
    if (in_place) {
      j->rc = vs_plan_launch(plans[c], VS_KIND_SYNTH, nullptr, 0, dst, j->root_pitch, nullptr, 0, nullptr);
      if (j->rc == VS_OK && ctx->tuning.fault == VS_FAULT_SHARD_HANDOVER) j->rc = VS_ERR_INTERNAL; /* tests */
      continue;
    }
    /* the buffer must have been copied out before it is overwritten */
    hipError_t e = hipSuccess;
    if (used[k]) e = hipStreamWaitEvent(nd->compute[s], nd->ev_copied[k][s], 0);
    if (e == hipSuccess) {
      j->rc = vs_plan_launch(plans[c], VS_KIND_SYNTH, nullptr, 0, (int16_t *)P.d_out[k], pitch, nullptr, 0, nullptr);
      if (j->rc == VS_OK && c == 0 && ctx->tuning.fault == VS_FAULT_SHARD_HANDOVER) j->rc = VS_ERR_INTERNAL; /* tests */
      if (j->rc != VS_OK) break;
      e = hipEventRecord(nd->ev_done[k][s], nd->compute[s]);
    }
    if (e == hipSuccess && rccl) {
      /* the chunk leaves by ncclSend behind its kernel; the root's thread posts the matching receive */
      e = hipStreamWaitEvent(nd->copy[s], nd->ev_done[k][s], 0);
      if (e == hipSuccess) {
        const int ne = nd->rccl.Send(P.d_out[k], rows * j->n_samples * sizeof(int16_t), VS_NCCL_INT8, 0, nd->comm[s], nd->copy[s]);
        if (ne != 0) {
          nd->last_rccl_error = ne;
          j->rc = VS_ERR_HIP;
          break;
        }
        e = hipEventRecord(nd->ev_copied[k][s], nd->copy[s]);
      }
      used[k] = true;
    } else if (e == hipSuccess && (j->flags & VS_NODE_OVERLAP)) {
      e = hipStreamWaitEvent(nd->copy[s], nd->ev_done[k][s], 0);
      if (e == hipSuccess)
        e = hipMemcpy2DAsync(dst, j->root_pitch * 2, P.d_out[k], pitch * 2, j->n_samples * 2, rows,
                             hipMemcpyDefault, nd->copy[s]);
      if (e == hipSuccess) e = hipEventRecord(nd->ev_copied[k][s], nd->copy[s]);
      used[k] = true;
    } else if (e == hipSuccess) {
      /* un-overlapped: this chunk's copy is issued behind ALL kernels; with more than two chunks
       * the two buffers force a wait here, which is the point of the comparison */
      e = hipStreamSynchronize(nd->compute[s]);
      if (e == hipSuccess)
        e = hipMemcpy2DAsync(dst, j->root_pitch * 2, P.d_out[k], pitch * 2, j->n_samples * 2, rows,
                             hipMemcpyDefault, nd->copy[s]);
      if (e == hipSuccess) e = hipStreamSynchronize(nd->copy[s]);
    }
    if (e != hipSuccess) {
      ctx->last_hip_error = (int)e;
      j->rc = VS_ERR_HIP;
    }
  }
  if (go && rccl && s == 0 && j->rc == VS_OK && S > 1) {
    /* the root's side of the exchange: round k = chunk k of every peer that has one, as one group on
     * one stream, straight into the peer's rows of the root buffer */
    const size_t rounds = vs_gather_rounds(j->n_total, (int)S, VS_NODE_CHUNK);
    for (size_t kk = 0; kk < rounds && j->rc == VS_OK && !sync->aborted.load(); kk++) {
      int ne = nd->rccl.GroupStart();
      for (size_t p = 1; p < S && ne == 0; p++) {
        size_t r0 = 0, rows = 0;
        if (vs_gather_round(j->n_total, (int)S, (int)p, VS_NODE_CHUNK, kk, &r0, &rows) != VS_OK || rows == 0) continue;
        ne = nd->rccl.Recv(j->root + r0 * j->root_pitch, rows * j->n_samples * sizeof(int16_t), VS_NCCL_INT8, (int)p,
                           nd->comm[0], nd->recv[0]);
      }
      const int ge = nd->rccl.GroupEnd();
      if (ne == 0) ne = ge;
      if (ne != 0) {
        nd->last_rccl_error = ne;
        j->rc = VS_ERR_HIP;
      }
    }
  }
  /* a failure behind the meeting point: the others have started, end what is in flight */
  if (go && j->rc != VS_OK) abort_exchange(nd, sync);
  if (go && hipSetDevice(nd->device[s]) == hipSuccess) {
    hipError_t e = hipStreamSynchronize(nd->compute[s]);
    j->compute_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    if (e == hipSuccess) e = hipStreamSynchronize(nd->copy[s]);
    if (e == hipSuccess && rccl && s == 0 && !nd->recv.empty()) e = hipStreamSynchronize(nd->recv[0]);
    if (e != hipSuccess && j->rc == VS_OK) {
      ctx->last_hip_error = (int)e;
      j->rc = VS_ERR_HIP;
    }
    for (vs_plan *pl : plans) {
      const int st = vs_plan_status(pl, nullptr);
      if (j->rc == VS_OK && st != VS_OK && !sync->aborted.load()) j->rc = st;
    }
  }
  for (vs_plan *pl : plans) vs_plan_destroy(pl);
  ctx->stream = saved;
}
}  // namespace

extern "C" int vs_node_synth_gather(vs_node *nd, const vs_lane *lanes, size_t n_lanes, size_t n_samples,
                                    int16_t *root_dev, size_t root_pitch, int flags, double *total_ms,
                                    double *max_compute_ms)
{
  if (!nd || !lanes || !root_dev || n_lanes == 0 || n_samples == 0 || root_pitch < n_samples) return VS_ERR_ARG;
  /* RCCL messages are contiguous: the root buffer must be packed */
  if (nd->transport == VS_NODE_TRANSPORT_RCCL && root_pitch != n_samples) return VS_ERR_UNSUPPORTED;
  const size_t S = nd->ctx.size();
  std::vector<ShardJob> jobs(S);
  std::vector<std::thread> th;
  GatherSync sync;
  sync.parties = S;
  const auto t0 = std::chrono::steady_clock::now();
  for (size_t s = 0; s < S; s++) {
    ShardJob &j = jobs[s];
    j.nd = nd;
    j.sync = &sync;
    j.s = s;
    j.lanes = lanes;
    shard_range(n_lanes, S, s, &j.lo, &j.hi);
    j.n_samples = n_samples;
    j.n_total = n_lanes;
    j.root = root_dev;
    j.root_pitch = root_pitch;
    j.flags = flags;
    j.rc = VS_OK;
  }
  try {
    for (size_t s = 0; s < S; s++) th.emplace_back(shard_gather, &jobs[s]);
  } catch (...) {
    /* the shards that did start are waiting for the ones that never will: stand in for those, with a no */
    for (size_t s = th.size(); s < S; s++) (void)sync.arrive(false);
    for (auto &x : th) x.join();
    return VS_ERR_NOMEM;
  }
  for (auto &x : th) x.join();
  if (total_ms) *total_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  double mc = 0.0;
  int rc = VS_OK;
  for (size_t s = 0; s < S; s++) {
    mc = std::max(mc, jobs[s].compute_ms);
    if (rc == VS_OK && jobs[s].rc != VS_OK) rc = jobs[s].rc;
  }
  if (max_compute_ms) *max_compute_ms = mc;
  if (sync.aborted.load() && nd->transport == VS_NODE_TRANSPORT_RCCL) {
    /* the communicators were aborted: the node is back on the peer transport (vs_node_set_transport makes new ones) */
    vs_node_drop_rccl(nd);
    nd->transport = VS_NODE_TRANSPORT_PEER;
    nd->link = nd->base_link;
  }
  return rc;
}

namespace {
struct RowsShift {
  vs_rows_cb cb;
  void *user;
  size_t lo;
};
int shifted(void *u, size_t row0, size_t rows, const int16_t *pcm)
{
  RowsShift *r = (RowsShift *)u;
  return r->cb(r->user, r->lo + row0, rows, pcm);
}
}  // namespace

extern "C" int vs_node_synth_rows(vs_node *nd, const vs_lane *lanes, size_t n_lanes, size_t n_samples,
                                  vs_rows_cb cb, void *user)
{
  if (!nd || !lanes || !cb || n_lanes == 0 || n_samples == 0) return VS_ERR_ARG;
  const size_t S = nd->ctx.size();
  std::vector<int> rcs(S, VS_OK);
  std::vector<RowsShift> sh(S);
  std::vector<std::thread> th;
  try {
    for (size_t s = 0; s < S; s++) {
      size_t lo, hi;
      shard_range(n_lanes, S, s, &lo, &hi);
      if (lo >= hi) continue;
      sh[s].cb = cb;
      sh[s].user = user;
      sh[s].lo = lo;
      th.emplace_back([=, &rcs, &sh]() {
        rcs[s] = vs_synth_rows(nd->ctx[s], lanes + lo, hi - lo, n_samples, shifted, &sh[s]);
      });
    }
  } catch (...) {
    for (auto &x : th) x.join();
    return VS_ERR_NOMEM;
  }
  for (auto &x : th) x.join();
  for (size_t s = 0; s < S; s++)
    if (rcs[s] != VS_OK) return rcs[s];
  return VS_OK;
}
