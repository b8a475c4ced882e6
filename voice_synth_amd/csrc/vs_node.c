/*
 * vs_node.c -- one batch over the GPUs of one node, in C (north star: "host code stays in C";
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
 * One host thread per shard (pthreads; a vs_ctx is used by one thread at a time).  The multi-PROCESS form
 * of the same scheme -- one rank per GPU under torch.distributed.run, RCCL send/recv of the
 * chunks -- is what bench.py --gpus N runs; see voice_synth_amd/dist.py.
 */
#define _GNU_SOURCE /* dladdr */
#include <dlfcn.h>
#include <pthread.h>
#include <stdatomic.h>
#include <stdlib.h>
#include <string.h>

#include "vs_commguard.h"
#include "vs_internal.h"

#define VS_NODE_CHUNK 16384
#define VS_NODE_MAX_SHARDS 64

/* the handful of RCCL entry points the gather uses, resolved from librccl at run time (the types are
 * RCCL's: ncclComm_t is an opaque pointer, ncclResult_t and ncclDataType_t are ints, ncclInt8 == 0) */
typedef void *vs_nccl_comm;
typedef struct VsRccl {
  void *lib;
  int (*CommInitAll)(vs_nccl_comm *, int, const int *);
  int (*CommDestroy)(vs_nccl_comm);
  int (*CommAbort)(vs_nccl_comm);
  int (*CommCount)(vs_nccl_comm, int *);
  int (*Send)(const void *, size_t, int, int, vs_nccl_comm, hipStream_t);
  int (*Recv)(void *, size_t, int, int, vs_nccl_comm, hipStream_t);
  int (*GroupStart)(void);
  int (*GroupEnd)(void);
} VsRccl;
#define VS_NCCL_INT8 0

struct vs_node {
  int n_shards;
  vs_ctx *ctx[VS_NODE_MAX_SHARDS];   /* one per shard */
  int device[VS_NODE_MAX_SHARDS];
  hipStream_t compute[VS_NODE_MAX_SHARDS], copy[VS_NODE_MAX_SHARDS];
  hipEvent_t ev_done[2][VS_NODE_MAX_SHARDS], ev_copied[2][VS_NODE_MAX_SHARDS];
  int link[VS_NODE_MAX_SHARDS];       /* VS_NODE_LINK_* of every shard */
  int base_link[VS_NODE_MAX_SHARDS];  /* ... as found at creation (self / peer / staged): what the peer transport uses */
  int transport;                      /* VS_NODE_TRANSPORT_* */
  VsRccl rccl;
  int n_comm;                         /* communicators made (0 or n_shards), RCCL transport only */
  /* one communicator per shard (rank = shard), each behind a guard (csrc/vs_commguard.h): a host call into RCCL -- the
   * shard's ncclSend, the root's GroupStart .. GroupEnd -- counts itself in and is made with NO lock held (it may block
   * until the peer answers); the ncclCommAbort of a failing shard's thread marks the communicator dead first, so that
   * nobody enters any more, aborts it -- which is what makes a blocked call return --, and forgets it once the callers
   * have left */
  VsCommGuard guard[VS_NODE_MAX_SHARDS];
  bool guards_ready;
  void *rccl_lib_aborted;             /* librccl of an aborted exchange: kept open until vs_node_destroy (its proxy
                                         threads may still be winding down when the gather returns) */
  hipStream_t recv;                   /* on the root device: THE stream the root's receive groups are posted on */
  int last_rccl_error;
};

int vs_node_create(const int *devices, int n_shards, vs_node **out)
{
  if (!devices || n_shards <= 0 || n_shards > VS_NODE_MAX_SHARDS || !out) return VS_ERR_ARG;
  *out = NULL;
  vs_node *nd = (vs_node *)calloc(1, sizeof(vs_node));
  if (!nd) return VS_ERR_NOMEM;
  nd->transport = VS_NODE_TRANSPORT_PEER;
  for (int s = 0; s < VS_NODE_MAX_SHARDS; s++) vs_commguard_init(&nd->guard[s]);
  nd->guards_ready = true;
  int rc = VS_OK;
  for (int s = 0; s < n_shards && rc == VS_OK; s++) {
    vs_ctx *c = NULL;
    rc = vs_ctx_create(devices[s], &c);
    if (rc != VS_OK) break;
    nd->ctx[s] = c;
    nd->device[s] = devices[s];
    nd->n_shards = s + 1;
    hipError_t he = hipSetDevice(devices[s]);
    if (he == hipSuccess) he = hipStreamCreateWithFlags(&nd->compute[s], hipStreamNonBlocking);
    if (he == hipSuccess) he = hipStreamCreateWithFlags(&nd->copy[s], hipStreamNonBlocking);
    for (int k = 0; k < 2 && he == hipSuccess; k++) {
      he = hipEventCreateWithFlags(&nd->ev_done[k][s], hipEventDisableTiming);
      if (he == hipSuccess) he = hipEventCreateWithFlags(&nd->ev_copied[k][s], hipEventDisableTiming);
    }
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
    nd->link[s] = link;
    nd->base_link[s] = link;
  }
  if (rc != VS_OK) {
    vs_node_destroy(nd);
    return rc;
  }
  *out = nd;
  return VS_OK;
}

/* keep_lib: the exchange was ABORTED -- the library stays loaded until vs_node_destroy (ncclCommAbort returns
 * while RCCL's proxy threads may still be running its code) */
static void vs_node_drop_rccl(vs_node *nd, bool keep_lib)
{
  for (int s = 0; s < nd->n_comm; s++) {
    vs_nccl_comm c = vs_commguard_take(&nd->guard[s]); /* NULL: aborted (gone already), or never made */
    if (c && nd->rccl.CommDestroy) {
      (void)hipSetDevice(nd->device[s]);
      (void)nd->rccl.CommDestroy(c);
    }
  }
  nd->n_comm = 0;
  if (nd->recv) {
    (void)hipSetDevice(nd->device[0]);
    (void)hipStreamDestroy(nd->recv);
    nd->recv = NULL;
  }
  if (nd->rccl.lib) {
    if (keep_lib && !nd->rccl_lib_aborted) nd->rccl_lib_aborted = nd->rccl.lib; /* one reference is enough to pin it */
    else dlclose(nd->rccl.lib);
  }
  memset(&nd->rccl, 0, sizeof(nd->rccl));
}

/* Before the transport is relied upon: one message from the root's communicator to ITSELF -- the only exchange that needs
 * no other thread -- through the very entry points the gather uses (GroupStart, Recv, Send, GroupEnd, on the root's receive
 * stream, bytes as ncclInt8), compared byte for byte.  A librccl whose entry points do not behave as the handful of
 * prototypes above say (another major version, the wrong library found by the loader) is refused HERE, with the transport
 * still on peer copies, not in the middle of an exchange; and on a one-GPU box it is what lets the tests run the send and
 * receive entry points against the real library at all (tests/test_gpu_node.py). */
static int rccl_link_check(vs_node *nd)
{
  enum { CHECK_BYTES = 1 << 16 };
  unsigned char *host = (unsigned char *)malloc(2 * CHECK_BYTES);
  void *src = NULL, *dst = NULL;
  int rc = host ? VS_OK : VS_ERR_NOMEM;
  hipError_t he = (rc == VS_OK) ? hipSetDevice(nd->device[0]) : hipSuccess;
  if (rc == VS_OK && he == hipSuccess) he = hipMalloc(&src, CHECK_BYTES);
  if (rc == VS_OK && he == hipSuccess) he = hipMalloc(&dst, CHECK_BYTES);
  if (rc == VS_OK && he == hipSuccess) {
    for (int i = 0; i < CHECK_BYTES; i++) host[i] = (unsigned char)((i * 131 + (i >> 8) + 7) & 0xFF);
    he = hipMemcpy(src, host, CHECK_BYTES, hipMemcpyHostToDevice);
    if (he == hipSuccess) he = hipMemset(dst, 0, CHECK_BYTES);
  }
  if (rc == VS_OK && he == hipSuccess) {
    vs_nccl_comm comm0 = vs_commguard_enter(&nd->guard[0]);
    if (!comm0) rc = VS_ERR_INTERNAL;
    else {
      int ne = nd->rccl.GroupStart();
      if (ne == 0) ne = nd->rccl.Recv(dst, CHECK_BYTES, VS_NCCL_INT8, 0, comm0, nd->recv);
      if (ne == 0) ne = nd->rccl.Send(src, CHECK_BYTES, VS_NCCL_INT8, 0, comm0, nd->recv);
      const int ge = nd->rccl.GroupEnd();
      vs_commguard_leave(&nd->guard[0]);
      if (ne == 0) ne = ge;
      if (ne != 0) {
        nd->last_rccl_error = ne;
        rc = VS_ERR_HIP;
      }
    }
    if (rc == VS_OK) he = hipStreamSynchronize(nd->recv);
    if (rc == VS_OK && he == hipSuccess) he = hipMemcpy(host + CHECK_BYTES, dst, CHECK_BYTES, hipMemcpyDeviceToHost);
    if (rc == VS_OK && he == hipSuccess && memcmp(host, host + CHECK_BYTES, CHECK_BYTES) != 0) rc = VS_ERR_INTERNAL;
  }
  if (rc == VS_OK && he != hipSuccess) rc = VS_ERR_HIP;
  if (src) (void)hipFree(src);
  if (dst) (void)hipFree(dst);
  free(host);
  return rc;
}

int vs_node_set_transport(vs_node *nd, int transport)
{
  if (!nd || (transport != VS_NODE_TRANSPORT_PEER && transport != VS_NODE_TRANSPORT_RCCL)) return VS_ERR_ARG;
  if (transport == nd->transport) return VS_OK;
  if (transport == VS_NODE_TRANSPORT_PEER) {
    vs_node_drop_rccl(nd, false);
    nd->transport = transport;
    memcpy(nd->link, nd->base_link, sizeof(nd->link));
    return VS_OK;
  }
  /* RCCL puts one rank on one device: logical shards of one device cannot form a communicator */
  const int S = nd->n_shards;
  for (int a = 0; a < S; a++)
    for (int b = a + 1; b < S; b++)
      if (nd->device[a] == nd->device[b]) return VS_ERR_UNSUPPORTED;
  VsRccl *R = &nd->rccl;
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
        R->lib = dlopen(path, RTLD_NOW | RTLD_LOCAL);
        if (!R->lib) {
          strcpy(path + dir, "librccl.so");
          R->lib = dlopen(path, RTLD_NOW | RTLD_LOCAL);
        }
      }
    }
  }
  if (!R->lib) R->lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
  if (!R->lib) R->lib = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
  if (!R->lib) return VS_ERR_UNSUPPORTED;
  *(void **)&R->CommInitAll = dlsym(R->lib, "ncclCommInitAll");
  *(void **)&R->CommDestroy = dlsym(R->lib, "ncclCommDestroy");
  *(void **)&R->CommAbort = dlsym(R->lib, "ncclCommAbort");
  *(void **)&R->CommCount = dlsym(R->lib, "ncclCommCount"); /* (optional: the pre-flight's figure) */
  *(void **)&R->Send = dlsym(R->lib, "ncclSend");
  *(void **)&R->Recv = dlsym(R->lib, "ncclRecv");
  *(void **)&R->GroupStart = dlsym(R->lib, "ncclGroupStart");
  *(void **)&R->GroupEnd = dlsym(R->lib, "ncclGroupEnd");
  if (!R->CommInitAll || !R->CommDestroy || !R->CommAbort || !R->Send || !R->Recv || !R->GroupStart || !R->GroupEnd) {
    vs_node_drop_rccl(nd, false);
    return VS_ERR_UNSUPPORTED;
  }
  vs_nccl_comm comms[VS_NODE_MAX_SHARDS];
  memset(comms, 0, sizeof(comms));
  const int e = R->CommInitAll(comms, S, nd->device);
  if (e != 0) {
    nd->last_rccl_error = e;
    nd->n_comm = 0;
    vs_node_drop_rccl(nd, false);
    return VS_ERR_HIP;
  }
  for (int s = 0; s < S; s++) vs_commguard_set(&nd->guard[s], comms[s]);
  nd->n_comm = S;
  /* the root posts its receive groups on ONE stream of its own (concurrency inside a group of
   * point-to-point operations comes from RCCL's channels, not from streams) */
  hipError_t he = hipSetDevice(nd->device[0]);
  if (he == hipSuccess) he = hipStreamCreateWithFlags(&nd->recv, hipStreamNonBlocking);
  if (he != hipSuccess) {
    vs_node_drop_rccl(nd, false);
    return VS_ERR_HIP;
  }
  const int lc = rccl_link_check(nd);
  if (lc != VS_OK) {
    vs_node_drop_rccl(nd, false);
    return lc;
  }
  nd->transport = transport;
  for (int s = 1; s < S; s++) nd->link[s] = VS_NODE_LINK_RCCL;
  return VS_OK;
}

int vs_node_link(const vs_node *nd, int shard)
{
  if (!nd || shard < 0 || shard >= nd->n_shards) return VS_ERR_ARG;
  return nd->link[shard];
}

int vs_node_last_rccl_error(const vs_node *nd) { return nd ? nd->last_rccl_error : 0; }

int vs_node_rccl_ranks(vs_node *nd, int shard)
{
  if (!nd || shard < 0 || shard >= nd->n_shards) return VS_ERR_ARG;
  if (nd->transport != VS_NODE_TRANSPORT_RCCL || shard >= nd->n_comm) return 0;
  if (!nd->rccl.CommCount) return VS_ERR_UNSUPPORTED;
  vs_nccl_comm c = vs_commguard_enter(&nd->guard[shard]);
  if (!c) return 0;
  int count = 0;
  const int e = nd->rccl.CommCount(c, &count);
  vs_commguard_leave(&nd->guard[shard]);
  if (e != 0) {
    nd->last_rccl_error = e;
    return VS_ERR_HIP;
  }
  return count;
}

void vs_node_destroy(vs_node *nd)
{
  if (!nd) return;
  vs_node_drop_rccl(nd, false);
  if (nd->rccl_lib_aborted) dlclose(nd->rccl_lib_aborted);
  for (int s = 0; s < nd->n_shards; s++) {
    (void)hipSetDevice(nd->device[s]);
    if (nd->compute[s]) (void)hipStreamDestroy(nd->compute[s]);
    if (nd->copy[s]) (void)hipStreamDestroy(nd->copy[s]);
    for (int k = 0; k < 2; k++) {
      if (nd->ev_done[k][s]) (void)hipEventDestroy(nd->ev_done[k][s]);
      if (nd->ev_copied[k][s]) (void)hipEventDestroy(nd->ev_copied[k][s]);
    }
    vs_ctx_destroy(nd->ctx[s]);
  }
  if (nd->guards_ready)
    for (int s = 0; s < VS_NODE_MAX_SHARDS; s++) vs_commguard_destroy(&nd->guard[s]);
  free(nd);
}

int vs_node_shards(const vs_node *nd) { return nd ? nd->n_shards : 0; }

int vs_node_ctx(vs_node *nd, int shard, vs_ctx **ctx)
{
  if (!nd || !ctx || shard < 0 || shard >= nd->n_shards) return VS_ERR_ARG;
  *ctx = nd->ctx[shard];
  return VS_OK;
}

int vs_node_set_arith(vs_node *nd, int arith)
{
  if (!nd) return VS_ERR_ARG;
  for (int s = 0; s < nd->n_shards; s++) {
    const int rc = vs_ctx_set_arith(nd->ctx[s], arith);
    if (rc != VS_OK) return rc;
  }
  return VS_OK;
}

/* lanes [lo, hi) of a shard: the cut of csrc/vs_host.c (vs_shard_cut), which the one-process-per-GPU
 * path makes too (voice_synth_amd/configs.py::shard_range) */
int vs_node_shard_range(const vs_node *nd, size_t n_lanes, int shard, size_t *lo, size_t *hi)
{
  if (!nd || !lo || !hi || shard < 0 || shard >= nd->n_shards) return VS_ERR_ARG;
  return vs_shard_cut(n_lanes, nd->n_shards, shard, lo, hi);
}

/* where the shard threads of one gather meet: between "all of my chunk plans exist" and "the first
 * kernel / copy / send / receive is enqueued" */
typedef struct GatherSync {
  pthread_mutex_t m;
  pthread_cond_t cv;
  int parties, waiting;
  unsigned generation;
  bool ok;              /* cleared by a shard that failed to prepare */
  atomic_bool aborted;  /* set by a shard that failed once the exchange had started */
  pthread_mutex_t abort_m;
} GatherSync;

/* every shard calls this once; returns the verdict of all of them */
static bool gather_arrive(GatherSync *g, bool mine)
{
  pthread_mutex_lock(&g->m);
  if (!mine) g->ok = false;
  const unsigned gen = g->generation;
  if (++g->waiting == g->parties) {
    g->waiting = 0;
    g->generation++;
    pthread_cond_broadcast(&g->cv);
  } else {
    while (g->generation == gen) pthread_cond_wait(&g->cv, &g->m);
  }
  const bool verdict = g->ok;
  pthread_mutex_unlock(&g->m);
  return verdict;
}

typedef struct ShardJob {
  vs_node *nd;
  GatherSync *sync;
  int s;
  const vs_lane *lanes;
  size_t lo, hi, n_samples, n_total;
  int16_t *root;      /* device pointer on device[0]: int16 [n_lanes][root_pitch] */
  size_t root_pitch;
  int flags;          /* VS_NODE_OVERLAP, VS_NODE_STAGE_ALL */
  int rc;
  double compute_ms;  /* host clock: first launch .. last kernel done */
} ShardJob;

/* A shard failed after the exchange had started: peers may hold sends nobody will receive, the root
 * receives nobody will send.  ncclCommAbort ends the kernels of the operations in flight, so that the
 * streams they sit on can be waited for again; the communicators are gone afterwards (the caller of
 * vs_node_synth_gather drops the transport).  Once per gather, whoever comes first. */
static void abort_exchange(vs_node *nd, GatherSync *sync)
{
  pthread_mutex_lock(&sync->abort_m);
  if (!atomic_exchange(&sync->aborted, true) && nd->transport == VS_NODE_TRANSPORT_RCCL) {
    /* every communicator is closed before the first one is aborted: a thread that comes back from a call on one of
     * them must not walk into the next.  The aborts themselves run with no lock of the guards held (vs_commguard_abort):
     * a peer -- or the root -- may be blocked INSIDE ncclSend / its receive group, waiting for the side of the exchange
     * the failed shard will never post, and ncclCommAbort is what brings it back */
    for (int p = 0; p < nd->n_comm; p++) vs_commguard_close(&nd->guard[p]);
    for (int p = 0; p < nd->n_comm; p++) (void)vs_commguard_abort(&nd->guard[p], nd->rccl.CommAbort);
  }
  pthread_mutex_unlock(&sync->abort_m);
}

typedef struct Chunk {
  size_t row0, rows;
} Chunk;

static void *shard_gather(void *arg)
{
  ShardJob *j = (ShardJob *)arg;
  vs_node *nd = j->nd;
  GatherSync *sync = j->sync;
  const int s = j->s;
  const int S = nd->n_shards;
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
  const size_t chunk = in_place ? (rows_all ? rows_all : 1) : VS_NODE_CHUNK;
  /* an RCCL message is one contiguous range: chunks are synthesised at pitch n_samples and land in a
   * root buffer of that pitch (checked by the caller) */
  const size_t pitch = rccl ? j->n_samples : ((j->n_samples + 7) & ~(size_t)7);
  VsPool *P = &ctx->pool;
  hipStream_t saved = ctx->stream;
  const size_t n_chunks = rows_all ? (rows_all + chunk - 1) / chunk : 0;
  Chunk *chunks = (Chunk *)calloc(n_chunks ? n_chunks : 1, sizeof(Chunk));
  vs_plan **plans = (vs_plan **)calloc(n_chunks ? n_chunks : 1, sizeof(vs_plan *));
  size_t n_plans = 0;

  /* ---- prepare: buffers and the plans of ALL chunks; nothing is enqueued yet ---- */
  if (!chunks || !plans) j->rc = VS_ERR_NOMEM;
  if (j->rc == VS_OK && hipSetDevice(nd->device[s]) != hipSuccess) j->rc = VS_ERR_HIP;
  if (j->rc == VS_OK && rows_all > 0) {
    for (size_t c = 0; c < n_chunks; c++) {
      if (in_place) {
        chunks[c].row0 = j->lo;
        chunks[c].rows = rows_all;
      } else if (vs_gather_round(j->n_total, S, s, VS_NODE_CHUNK, c, &chunks[c].row0, &chunks[c].rows) != VS_OK || chunks[c].rows == 0) {
        j->rc = VS_ERR_INTERNAL; /* cannot happen: n_chunks comes from the same cut */
        break;
      }
    }
    if (!in_place) {
      const size_t first = rows_all < chunk ? rows_all : chunk;
      for (int k = 0; k < 2 && j->rc == VS_OK && (k == 0 || n_chunks > 1); k++)
        j->rc = vs_pool_device(ctx, &P->d_out[k], &P->d_out_bytes[k], first * pitch * sizeof(int16_t));
    }
    ctx->stream = nd->compute[s];
    for (size_t c = 0; c < n_chunks && j->rc == VS_OK; c++) {
      vs_plan *plan = NULL;
      j->rc = vs_plan_create_impl(ctx, j->lanes + chunks[c].row0, chunks[c].rows, j->n_samples, VS_PLAN_POOL_SCRATCH, &plan);
      if (j->rc == VS_OK) plans[n_plans++] = plan;
    }
    if (j->rc == VS_OK && ctx->tuning.fault == VS_FAULT_SHARD_PREPARE) j->rc = VS_ERR_INTERNAL; /* tests */
  }
  /* ---- all or nothing: one shard that cannot go on keeps every shard from starting ---- */
  const bool go = gather_arrive(sync, j->rc == VS_OK);

  const double t0 = vs_now_ms();
  bool used[2] = {false, false};
  int k = 0;
  for (size_t c = 0; go && c < n_plans && j->rc == VS_OK && !atomic_load(&sync->aborted); c++, k ^= 1) {
    const size_t r0 = chunks[c].row0, rows = chunks[c].rows;
    int16_t *dst = j->root + r0 * j->root_pitch;
    if (in_place) {
      j->rc = vs_plan_launch(plans[c], VS_KIND_SYNTH, NULL, 0, dst, j->root_pitch, NULL, 0, NULL);
      if (j->rc == VS_OK && ctx->tuning.fault == VS_FAULT_SHARD_HANDOVER) j->rc = VS_ERR_INTERNAL; /* tests */
      continue;
    }
    /* the buffer must have been copied out before it is overwritten */
    hipError_t e = hipSuccess;
    if (used[k]) e = hipStreamWaitEvent(nd->compute[s], nd->ev_copied[k][s], 0);
    if (e == hipSuccess) {
      j->rc = vs_plan_launch(plans[c], VS_KIND_SYNTH, NULL, 0, (int16_t *)P->d_out[k], pitch, NULL, 0, NULL);
      if (j->rc == VS_OK && c == 0 && ctx->tuning.fault == VS_FAULT_SHARD_HANDOVER) j->rc = VS_ERR_INTERNAL; /* tests */
      if (j->rc != VS_OK) break;
      e = hipEventRecord(nd->ev_done[k][s], nd->compute[s]);
    }
    if (e == hipSuccess && rccl) {
      /* the chunk leaves by ncclSend behind its kernel; the root's thread posts the matching receive */
      e = hipStreamWaitEvent(nd->copy[s], nd->ev_done[k][s], 0);
      if (e == hipSuccess) {
        vs_nccl_comm comm = vs_commguard_enter(&nd->guard[s]);
        if (!comm) break; /* another shard failed and aborted the exchange: stop, no error of ours */
        const int ne = nd->rccl.Send(P->d_out[k], rows * j->n_samples * sizeof(int16_t), VS_NCCL_INT8, 0, comm, nd->copy[s]);
        vs_commguard_leave(&nd->guard[s]);
        if (ne != 0 && atomic_load(&sync->aborted)) break; /* ended by the abort of a failing shard, not a failure of this one */
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
        e = hipMemcpy2DAsync(dst, j->root_pitch * 2, P->d_out[k], pitch * 2, j->n_samples * 2, rows,
                             hipMemcpyDefault, nd->copy[s]);
      if (e == hipSuccess) e = hipEventRecord(nd->ev_copied[k][s], nd->copy[s]);
      used[k] = true;
    } else if (e == hipSuccess) {
      /* un-overlapped: this chunk's copy is issued behind ALL kernels; with more than two chunks
       * the two buffers force a wait here, which is the point of the comparison */
      e = hipStreamSynchronize(nd->compute[s]);
      if (e == hipSuccess)
        e = hipMemcpy2DAsync(dst, j->root_pitch * 2, P->d_out[k], pitch * 2, j->n_samples * 2, rows,
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
    const size_t rounds = vs_gather_rounds(j->n_total, S, VS_NODE_CHUNK);
    for (size_t kk = 0; kk < rounds && j->rc == VS_OK && !atomic_load(&sync->aborted); kk++) {
      vs_nccl_comm comm0 = vs_commguard_enter(&nd->guard[0]);
      if (!comm0) break; /* aborted by a failing shard between two rounds */
      int ne = nd->rccl.GroupStart();
      for (int p = 1; p < S && ne == 0; p++) {
        size_t r0 = 0, rows = 0;
        if (vs_gather_round(j->n_total, S, p, VS_NODE_CHUNK, kk, &r0, &rows) != VS_OK || rows == 0) continue;
        ne = nd->rccl.Recv(j->root + r0 * j->root_pitch, rows * j->n_samples * sizeof(int16_t), VS_NCCL_INT8, p,
                           comm0, nd->recv);
      }
      const int ge = nd->rccl.GroupEnd();
      vs_commguard_leave(&nd->guard[0]);
      if (ne == 0) ne = ge;
      if (ne != 0 && atomic_load(&sync->aborted)) break; /* ended by a failing shard's abort */
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
    j->compute_ms = vs_now_ms() - t0;
    if (e == hipSuccess) e = hipStreamSynchronize(nd->copy[s]);
    if (e == hipSuccess && rccl && s == 0 && nd->recv) e = hipStreamSynchronize(nd->recv);
    if (e != hipSuccess && j->rc == VS_OK) {
      ctx->last_hip_error = (int)e;
      j->rc = VS_ERR_HIP;
    }
    for (size_t c = 0; c < n_plans; c++) {
      const int st = vs_plan_status(plans[c], NULL);
      if (j->rc == VS_OK && st != VS_OK && !atomic_load(&sync->aborted)) j->rc = st;
    }
  }
  for (size_t c = 0; c < n_plans; c++) vs_plan_destroy(plans[c]);
  ctx->stream = saved;
  free(chunks);
  free(plans);
  return NULL;
}

int vs_node_synth_gather(vs_node *nd, const vs_lane *lanes, size_t n_lanes, size_t n_samples,
                         int16_t *root_dev, size_t root_pitch, int flags, double *total_ms,
                         double *max_compute_ms)
{
  if (!nd || !lanes || !root_dev || n_lanes == 0 || n_samples == 0 || root_pitch < n_samples) return VS_ERR_ARG;
  /* RCCL messages are contiguous: the root buffer must be packed */
  if (nd->transport == VS_NODE_TRANSPORT_RCCL && root_pitch != n_samples) return VS_ERR_UNSUPPORTED;
  const int S = nd->n_shards;
  ShardJob jobs[VS_NODE_MAX_SHARDS];
  pthread_t th[VS_NODE_MAX_SHARDS];
  GatherSync sync;
  memset(&sync, 0, sizeof(sync));
  pthread_mutex_init(&sync.m, NULL);
  pthread_mutex_init(&sync.abort_m, NULL);
  pthread_cond_init(&sync.cv, NULL);
  sync.parties = S;
  sync.ok = true;
  atomic_init(&sync.aborted, false);
  const double t0 = vs_now_ms();
  for (int s = 0; s < S; s++) {
    ShardJob *j = &jobs[s];
    j->nd = nd;
    j->sync = &sync;
    j->s = s;
    j->lanes = lanes;
    (void)vs_shard_cut(n_lanes, S, s, &j->lo, &j->hi);
    j->n_samples = n_samples;
    j->n_total = n_lanes;
    j->root = root_dev;
    j->root_pitch = root_pitch;
    j->flags = flags;
    j->rc = VS_OK;
    j->compute_ms = 0.0;
  }
  int started = 0;
  for (int s = 0; s < S; s++) {
    if (pthread_create(&th[s], NULL, shard_gather, &jobs[s]) != 0) break;
    started++;
  }
  /* the shards that did start are waiting for the ones that never will: stand in for those, with a no */
  for (int s = started; s < S; s++) (void)gather_arrive(&sync, false);
  for (int s = 0; s < started; s++) pthread_join(th[s], NULL);
  if (total_ms) *total_ms = vs_now_ms() - t0;
  double mc = 0.0;
  int rc = (started < S) ? VS_ERR_NOMEM : VS_OK;
  for (int s = 0; s < started; s++) {
    if (jobs[s].compute_ms > mc) mc = jobs[s].compute_ms;
    if (rc == VS_OK && jobs[s].rc != VS_OK) rc = jobs[s].rc;
  }
  if (max_compute_ms) *max_compute_ms = mc;
  if (atomic_load(&sync.aborted) && nd->transport == VS_NODE_TRANSPORT_RCCL) {
    /* the communicators were aborted: the node is back on the peer transport (vs_node_set_transport makes new
     * ones); librccl itself stays loaded until the node is destroyed */
    vs_node_drop_rccl(nd, true);
    nd->transport = VS_NODE_TRANSPORT_PEER;
    memcpy(nd->link, nd->base_link, sizeof(nd->link));
  }
  pthread_cond_destroy(&sync.cv);
  pthread_mutex_destroy(&sync.abort_m);
  pthread_mutex_destroy(&sync.m);
  return rc;
}

typedef struct RowsJob {
  vs_node *nd;
  int s;
  const vs_lane *lanes;
  size_t lo, hi, n_samples;
  vs_rows_cb cb;
  void *user;
  int rc;
} RowsJob;

static int shifted(void *u, size_t row0, size_t rows, const int16_t *pcm)
{
  RowsJob *r = (RowsJob *)u;
  return r->cb(r->user, r->lo + row0, rows, pcm);
}

static void *shard_rows(void *arg)
{
  RowsJob *r = (RowsJob *)arg;
  r->rc = vs_synth_rows(r->nd->ctx[r->s], r->lanes + r->lo, r->hi - r->lo, r->n_samples, shifted, r);
  return NULL;
}

int vs_node_synth_rows(vs_node *nd, const vs_lane *lanes, size_t n_lanes, size_t n_samples,
                       vs_rows_cb cb, void *user)
{
  if (!nd || !lanes || !cb || n_lanes == 0 || n_samples == 0) return VS_ERR_ARG;
  const int S = nd->n_shards;
  RowsJob jobs[VS_NODE_MAX_SHARDS];
  pthread_t th[VS_NODE_MAX_SHARDS];
  bool running[VS_NODE_MAX_SHARDS];
  int rc = VS_OK;
  for (int s = 0; s < S; s++) {
    RowsJob *r = &jobs[s];
    running[s] = false;
    r->nd = nd;
    r->s = s;
    r->lanes = lanes;
    (void)vs_shard_cut(n_lanes, S, s, &r->lo, &r->hi);
    r->n_samples = n_samples;
    r->cb = cb;
    r->user = user;
    r->rc = VS_OK;
    if (r->lo >= r->hi) continue;
    if (pthread_create(&th[s], NULL, shard_rows, r) != 0) {
      rc = VS_ERR_NOMEM;
      break;
    }
    running[s] = true;
  }
  for (int s = 0; s < S; s++)
    if (running[s]) pthread_join(th[s], NULL);
  for (int s = 0; s < S && rc == VS_OK; s++)
    if (running[s] && jobs[s].rc != VS_OK) rc = jobs[s].rc;
  return rc;
}
