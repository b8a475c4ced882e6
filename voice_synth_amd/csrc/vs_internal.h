/*
 * vs_internal.h -- what the translation units of libvoicesynth share behind the C ABI:
 * the context and plan records, the context's buffer pool, internal entry points.
 * Plain C: the host side of the library (vs_host.c, vs_planhost.c, vs_api.c, vs_delivery.c, vs_node.c) is compiled by
 * the C compiler against the HIP runtime's C API; only the kernels and their launchers (vs_kernels.hip)
 * are HIP C++.
 */
#ifndef VS_INTERNAL_H
#define VS_INTERNAL_H

#include <hip/hip_runtime_api.h>
#include <stdatomic.h>
#include <stdbool.h>

#include "../../include/voice_synth.h"
#include "vs_device.h"
#include "vs_planhost.h"

#define VS_DELIVERY_THREADS 4     /* row-chunk delivery workers (each: one stream + one pinned staging buffer) */
#define VS_STAGING_BYTES (16u << 20)

/* Buffers a context keeps between calls (grown on demand, freed by vs_ctx_destroy or
 * vs_ctx_trim): nothing on the host-buffer paths calls hipMalloc / hipHostMalloc per call. */
typedef struct VsPool {
  void *d_out[2];           /* device PCM of the compute chunk in flight / being delivered */
  size_t d_out_bytes[2];
  void *d_in;               /* filter-only kind: the uploaded flow */
  size_t d_in_bytes;
  void *d_aux;              /* cycle log + cycle counts of vs_source */
  size_t d_aux_bytes;
  void *d_flow;             /* wide plans made by the pipelines: the flow between source and wide filter kernel */
  size_t d_flow_bytes;
  void *staging[VS_DELIVERY_THREADS]; /* pinned host memory, staging_bytes each (VS_STAGING_BYTES, more when one row is longer) */
  size_t staging_bytes;
  hipStream_t copy_stream[VS_DELIVERY_THREADS];
  hipStream_t compute_stream; /* compute chunks of vs_synth_rows when the caller set no stream */
  hipStream_t upload_stream;  /* lane records + cos rows of the NEXT chunk's plan, while this chunk's kernel runs */
  hipEvent_t done[2];       /* kernel of the chunk in d_out[k] has finished */
  int streams_ready;
  void *zc_io;              /* small calls (vs_source / vs_filter of a few utterances: the drop-in programs): PCM in and out, */
  size_t zc_io_bytes;       /* cycle log and counts in ONE block of pinned, device-mapped host memory -- no copy at all */
} VsPool;

/* Device blocks of plans that have been destroyed, kept for the plans to come: a caller who synthesises batch after batch
 * of new utterances makes and destroys a plan per batch, and hipFree waits for the DEVICE -- for the kernels of the batches
 * behind, which have nothing to do with the block (cli/vs_bench.c --fresh: 4.3 ms per batch where the kernel takes 2.5).
 * A retired block carries the event its plan recorded behind its last launch (vs_plan_launch re-records it every time: a
 * few microseconds); the plan that takes the block over waits for THAT -- the one kernel that read the block -- not for
 * the device, and not for whatever else the stream has taken on since. */
#define VS_PLAN_CACHE_SLOTS 32
typedef struct VsRetire { /* shared by the blocks of one destroyed plan */
  hipEvent_t ev;          /* recorded behind the plan's LAST launch, when that launch was enqueued */
  int refs;
} VsRetire;
typedef struct VsBlock {
  void *ptr;
  size_t bytes;
  VsRetire *retired;   /* NULL: the plan was never launched */
  unsigned long stamp; /* age: the oldest goes when the cache is full */
} VsBlock;

struct vs_ctx {
  int device;
  int arith;
  hipStream_t stream;
  hipStream_t upload;   /* when set (the chunk pipelines): plan records go up on this stream */
  hipStream_t own_upload; /* ... else on this one, created with the first plan that copies: NEVER on `stream`, where the
                             copy would queue behind whatever kernel the caller has running there -- a caller who makes
                             batch k + 1's plan while batch k's kernel runs would wait for that kernel in vs_plan_create */
  _Atomic int last_hip_error; /* (the one word both halves of a pipelining caller may write: include/voice_synth.h, "Threading") */
  char name[128];
  int cu_count;
  vs_tuning tuning; /* all zero = the library's own choices */
  VsPool pool;
  void *plan_pin;          /* page-locked host block the small parts of a plan go up from (cos rows + taps, group table, the zero word) */
  size_t plan_pin_bytes;
  struct VsPlanWs *planws; /* plan creation's worker threads and host buffers between calls (csrc/vs_planhost.c; a context is used by one thread at a time) */
  hipEvent_t timer[2];    /* vs_ctx_timer_mark: made on first use */
  VsBlock plan_cache[VS_PLAN_CACHE_SLOTS]; /* touched by vs_plan_create / vs_plan_destroy only (one thread at a time) */
  unsigned long plan_cache_stamp;
  int copy_warm;          /* the runtime's copy path has been set up (vs_copy_path_warm) */
  double copy_warm_ms;    /* ... and what that cost */
  /* where the hardware deals the wavefronts of a workgroup (vs_ctx_simd_dealing): asked once, the first time a
   * plan wants a layout that is built on it */
  int simd_probed;        /* 0: not yet; 1: done; -1: the probe itself failed (treated as "not cyclic") */
  int simd_cyclic12;      /* wavefront w of a 12-wavefront workgroup ran next to wavefront w % 4 (first four on different SIMDs) in every workgroup probed */
  int simd_cyclic8;       /* ... of an 8-wavefront workgroup */
  unsigned simd_odd_wgs;  /* workgroups of the probe that were dealt differently (selftest counter [6]) */
};

/* the smallest host-to-device copy the runtime hands to a DMA engine instead of a copy kernel (measured: 16 KiB kernel,
 * 64 KiB DMA; tools/overlap_probe.py) */
#define VS_SMALL_BLOCK_MIN ((size_t)65536)

struct vs_plan {
  vs_ctx *ctx;
  size_t n_lanes, n_samples;
  VsDevLane *d_lanes;
  hipEvent_t last_launch;   /* recorded behind every launch (and reseed) of this plan; NULL until the first: what its blocks retire behind */
  hipStream_t last_stream;  /* ... and the stream it was last recorded on */
  size_t cap_lanes, cap_small, cap_sink, cap_ondw, cap_odone; /* what the blocks below really hold (they may come from the context's cache of retired blocks) */
  char *d_small;     /* plans that copy: ONE device block for cos rows + taps, the mixed-rings table and the error word (d_costab, d_group_map, d_err point into it) */
  double *d_costab;
  double *d_taps;    /* the tap table: [rows][22] (rows 0..9 the ten tables, then the lanes' own sets) */
  int ring_slots;
  int ready_min;
  int ltab_entries;
  size_t lds_bytes;  /* dynamic LDS of a wave-specialised workgroup's groups (mixed rings: the largest workgroup's sum); one group's ring + cos rows otherwise */
  size_t lds_one_wave; /* ... of one group on the one-wave kernel (source-only kind, cycle log), whatever the fused launch takes */
  unsigned grid;
  unsigned long long *d_diag; /* VS_DIAG builds: [grid][8] cycle counters, else NULL */
  int *d_err;                 /* spin-limit word of the wave-specialised kernel */
  int16_t *d_sink;            /* VsKernelArgs.sink */
  float *d_ondw;              /* vowel -n: NoiseDistWidth per frame [n_lanes][ondw_pitch], NULL if unused */
  long ondw_pitch;
  int32_t *d_odone;           /* ... frames per row the fused kernel has dealt with itself (VsKernelArgs.odone), NULL if it never does */
  int pow_lframe;             /* ... the frame length all lanes share when it does, else 0 */
  int wave_specialised;
  int ws_pairs;      /* groups of 64 utterances per workgroup of the wave-specialised launch */
  int ws_roles;      /* wavefronts per group: 2 or 3 (VsKernelArgs.ws_roles) */
  int ws_layout;     /* VS_WS_LAYOUT_* (VsKernelArgs.ws_layout) */
  VsGroupSlot *d_group_map; /* mixed rings: [workgroups][ws_pairs] on the device, NULL for uniform rings */
  int ring_slots_min;       /* mixed rings: the shallowest ring of the plan (ring_slots holds the deepest) */
  int simd_fallback; /* the plan wanted three roles and took two because the wavefronts are not dealt four at a time */
  int ws_shared_simd; /* the wavefronts of a group share a SIMD (grids beyond two groups per CU) */
  int group_lanes;   /* utterances per wavefront: VS_WAVE, or VS_NARROW_LANES for periods beyond the 64-column ring */
  int ws_pair_bytes; /* LDS bytes of one pair */
  int filter_only;   /* made by vs_filter(): no source records, no ring, VS_KIND_FILTER launches only */
  int pre1;          /* every lane has pre_emphasis == 1.0 (the reference's default) */
  int wide;          /* some lane carries a coefficient set of 23..40 taps: source kernel -> flow in HBM -> wide filter kernel */
  double *d_awide;   /* wide plans: A[1..40] per lane record */
  int16_t *d_flow;   /* wide plans that synthesise: the flow between the two kernels [n_lanes][flow_pitch] */
  size_t flow_pitch;
  int owns_flow;     /* d_flow was allocated for this plan (else: the context pool's, shared by the pipeline's chunk plans) */
  vs_tuning tuning;  /* the context's tuning when the plan was made */
  unsigned long long *d_seeds; /* vs_plan_reseed: [2][n_lanes] seeds / out_seeds on the device, allocated with the first reseed */
  unsigned long long *h_seeds; /* ... and in pinned host memory: the caller's arrays are copied there, so they are his again at once */
  hipEvent_t seeds_copied;     /* the upload out of h_seeds has run (a second reseed waits for it before it overwrites h_seeds) */
  void *zc_host;     /* zero-copy plans: lane records, cos rows, error word (and wide taps) in one pinned, device-mapped host block */
  int *zc_err;       /* ... the error word as the host sees it */
  double host_ms;    /* host time of vs_plan_create: expansion, sorting, tables */
  double upload_ms;  /* ... and of the allocation + upload + wait that follows */
};

#define VS_HIP(ctx, call)                        \
  do {                                           \
    hipError_t e_ = (call);                      \
    if (e_ != hipSuccess) {                      \
      (ctx)->last_hip_error = (int)e_;           \
      return VS_ERR_HIP;                         \
    }                                            \
  } while (0)

/* the launchers of csrc/vs_kernels.hip (extern "C" there) */
#ifdef __cplusplus
extern "C" {
#endif
hipError_t vs_launch_selftest(unsigned long long *bad_dev, hipStream_t stream);
hipError_t vs_launch_simd_probe(int waves, unsigned grid, size_t lds_bytes, unsigned *out_dev, hipStream_t stream);
hipError_t vs_launch_reseed(VsDevLane *lanes, const unsigned long long *seeds, const unsigned long long *out_seeds,
                            int n_lanes, hipStream_t stream);
hipError_t vs_launch_out_noise(const VsKernelArgs *args, hipStream_t stream);
hipError_t vs_launch_filter_wide(int arith, const VsKernelArgs *args, unsigned grid, hipStream_t stream);
hipError_t vs_launch_kernel(int arith, int kind, bool log, bool wave_specialised, bool pre1, const VsKernelArgs *args,
                            unsigned grid, size_t lds_bytes, hipStream_t stream);
#ifdef __cplusplus
}
#endif

/* milliseconds on the monotonic clock */
double vs_now_ms(void);

/* mode: VS_PLAN_* bits */
#define VS_PLAN_FILTER_ONLY 1  /* made by vs_filter(): no source records, no ring, VS_KIND_FILTER launches only */
#define VS_PLAN_POOL_SCRATCH 2 /* chunk plans of a pipeline: launches are serial on one stream, so a wide plan's
                                  flow buffer is the context pool's instead of one allocation per chunk */
#define VS_PLAN_ZERO_COPY 4    /* a handful of utterances (the drop-in programs): the plan's records live in pinned,
                                  device-mapped host memory and the kernel reads them over PCIe -- no hipMemcpy, so a
                                  process that only ever makes such plans never pays the 27 ms the runtime takes to set
                                  up its copy path (profiles/r05_cli_startup.txt) */
/* the runtime's first host-to-device copy sets up its copy path (27 ms, whatever the size); paid once per context,
 * outside the plan's own timing, the first time a plan that copies is made */
int vs_copy_path_warm(vs_ctx *ctx);
int vs_plan_create_impl(vs_ctx *ctx, const vs_lane *lanes, size_t n_lanes, size_t n_samples,
                        int mode, vs_plan **out);
/* bookkeeping of a node's gather (csrc/vs_host.c, plain C): the cut of a batch over the shards and the rows
 * that travel in each round; walked by the sending shards AND by the receiving root */
#ifdef __cplusplus
extern "C" {
#endif
int vs_shard_cut(size_t n_lanes, int n_shards, int shard, size_t *lo, size_t *hi);
size_t vs_gather_rounds(size_t n_lanes, int n_shards, size_t chunk);
int vs_gather_round(size_t n_lanes, int n_shards, int shard, size_t chunk, size_t round, size_t *row0, size_t *rows);
#ifdef __cplusplus
}
#endif
/* grows *ptr (device memory) to at least bytes; VS_OK or VS_ERR_HIP */
int vs_pool_device(vs_ctx *ctx, void **ptr, size_t *have, size_t bytes);
/* creates the delivery streams, events and pinned staging buffers (at least row_bytes each) on first use */
int vs_pool_streams(vs_ctx *ctx, size_t row_bytes);
void vs_pool_release(vs_ctx *ctx);
/* the cache of retired plan blocks (csrc/vs_api.c): hipFree of all of them */
void vs_plan_cache_release(vs_ctx *ctx);

#endif
