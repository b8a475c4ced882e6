/*
 * voice_synth.h -- C ABI of the MI355X (gfx950) batched vowel-synthesis engine.
 *
 * This is the drop-in boundary for ONE hot path of jsansao/voice_synth:
 *
 *     flowgen_shimmer (glottal source)  -->  vowel (order-22 all-pole vocal-tract filter)
 *
 * The reference exposes no library or FFI surface -- each stage is the body of a main()
 * (reference flowgen_shimmer.c:246-423 and vowel_new.c:237-331) -- so the entry points
 * below are what a binding for this path binds instead of those loops.  Every entry cites
 * the reference lines it replaces.  Plain C types only: pointers, sizes, fixed-width
 * integers.  No function calls exit(); every failure is a negative return code.
 *
 * Threading: a vs_ctx and the plans made from it may be used by one thread at a time -- with one
 * exception, made for callers who synthesise batch after batch of NEW utterances: while one thread
 * launches, reseeds, waits and times (vs_plan_launch, vs_plan_reseed, vs_plan_status, vs_ctx_synchronize,
 * vs_ctx_timer_*), ONE other thread may be inside vs_plan_create() or vs_plan_destroy() of the same
 * context -- plan creation touches nothing a launch reads, works on host threads and a stream of its
 * own, and the plan it returns is complete (cli/vs_bench.c --fresh does exactly that: the plan of
 * batch k + 1 is made while kernel k runs).  Different contexts are independent.  There is no global
 * mutable state, and the library reads no environment variable after vs_ctx_create() (see
 * vs_ctx_set_tuning()).
 *
 * There is no CPU fallback: if no gfx950 device is usable, vs_ctx_create() fails with
 * VS_ERR_NODEVICE and nothing can be synthesised.
 */
#ifndef VOICE_SYNTH_H
#define VOICE_SYNTH_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Filter order.  The reference runs Order = 22 for every table (vowel_new.c:172) inside arrays
 * bounded by MAX_ORDER 40 (vowel_new.c:33); its loop (vowel_new.c:279-281, 287-289) is written for
 * any Order.  A VS_VOWEL_CUSTOM coefficient set carries its own order (vs_lane.order, 1..40):
 *   - up to 22 taps it rides the fused kernel like a table (missing taps are zeros: acc - 0*y == acc,
 *     so the result equals the reference's loop run with the smaller Order);
 *   - 23..40 taps take the WIDE path: the source kernel writes the flow to HBM and a filter kernel
 *     with a 48-sample register window reads it back (un-fused: 6 bytes of HBM traffic per sample
 *     instead of 2, and ~2x the arithmetic).  A plan is wide as a whole as soon as one of its lanes
 *     is. */
#define VS_ORDER 22
#define VS_NCOEF (VS_ORDER + 1)
#define VS_MAX_ORDER 40
#define VS_MAX_NCOEF (VS_MAX_ORDER + 1)

/* return codes */
#define VS_OK 0
#define VS_ERR_ARG (-1)          /* NULL pointer, zero size, bad enum */
#define VS_ERR_RANGE (-2)        /* a parameter the reference would reject with usage() */
#define VS_ERR_UNSUPPORTED (-3)  /* legal for the reference's parser, but undefined behaviour
                                    there or beyond this engine's limits */
#define VS_ERR_HIP (-4)          /* a HIP runtime call failed; see vs_ctx_last_hip_error() */
#define VS_ERR_NOMEM (-5)
#define VS_ERR_NODEVICE (-6)     /* no usable gfx950 device: there is no CPU path */
#define VS_ERR_IO (-7)
#define VS_USAGE (-8)            /* argv parsers: the reference would print usage() and exit(0) */
#define VS_ERR_INTERNAL (-9)     /* a device-side bounded wait ran out (kernel bug, never expected) */

/* vs_lane.flags: which perturbation options were GIVEN on the command line.  The reference
 * tests "arg.X != -1" (flowgen_shimmer.c:248, 295, 373), not only the value. */
#define VS_FLAG_JITTER 0x1u
#define VS_FLAG_SHIMMER 0x2u
#define VS_FLAG_NOISE 0x4u

/* vs_lane.vowel: the reference's -v menu (vowel_new.c:153-156, 548-627) or an explicit
 * coefficient set. */
#define VS_VOWEL_CUSTOM 0

/*
 * One utterance ("lane").  Source fields are struct PAR of flowgen_shimmer.c:73-87 AFTER
 * initialization() (flowgen_shimmer.c:463-546) has converted the command-line units;
 * filter fields are the globals of vowel_new.c:76-77 plus the coefficient choice.
 * Duration is not per lane: a batch has one sample count (see vs_num_samples()).
 */
typedef struct vs_lane {
  float jitter;       /* mean jitter as a fraction: "-j x" / 100            fg:477 */
  float cq;           /* closed quotient                                    fg:490 */
  float K;            /* speed of closure                                   fg:484 */
  float Fg;           /* glottal formant, validation only                   fg:496 */
  float F0;           /* fundamental frequency, 50 <= F0 < Fg               fg:504 */
  float DC;           /* ABSOLUTE DC flow: "-l x" * amp, or 0.25 after -n   fg:182,524 */
  float noise;        /* linear SNR: pow(10, "-n x" / 10)                   fg:511 */
  float Kvar;         /* closure-speed variation                            fg:530 */
  float shimmer;      /* shimmer as a fraction: "-s x" / 100                fg:544 */
  int32_t fs;         /* sampling rate                                      fg:538 */
  int32_t amp;        /* maximum amplitude                                  fg:518 */
  uint32_t flags;     /* VS_FLAG_* */
  uint64_t seed;      /* Philox key of this lane's draw stream (replaces srandom(time), fg:241) */
  float gain;         /* vowel -g                                           vw:131 */
  float pre_emphasis; /* vowel -p                                           vw:126 */
  int32_t vowel;      /* 'a','i','u','1'..'7', or VS_VOWEL_CUSTOM           vw:152 */
  float out_snr;      /* vowel -n: linear SNR pow(10, x/10) of the white noise added to the
                         filtered signal frame by frame, 0 = off                vw:141-143, 302-324 */
  double A[VS_MAX_NCOEF]; /* A(z) when vowel == VS_VOWEL_CUSTOM: A[0] must be 1.0, A[1..order] the taps */
  int32_t order;      /* taps of the VS_VOWEL_CUSTOM set, 1..VS_MAX_ORDER; 0 means VS_ORDER (22) */
  int32_t reserved_;  /* keeps out_seed 8-byte aligned without implicit padding; must be 0 */
  uint64_t out_seed;  /* Philox key of the vowel stage's own draw stream (the reference's vowel
                         process calls srandom(time) itself, vw:234): one draw per sample */
} vs_lane;

/* Per-cycle diagnostics the reference prints inside its loop (flowgen_shimmer.c:307, 409).
 * Only the single-utterance CLI asks for them. */
typedef struct vs_cycle_rec {
  float S;      /* shimmer draw of the cycle ("%5.2f \n"), 0 when shimmer is off */
  float x_pow;  /* open-phase power   fg:378 */
  float w_pow;  /* noise power        fg:407 ; SNRdb = 10*log10(x_pow / w_pow) */
  int32_t T;    /* period of the cycle in samples */
} vs_cycle_rec;

/* Arithmetic of the filter recurrence.
 * VS_ARITH_EXACT: products and subtractions rounded one by one in the reference's order
 *                 (vowel_new.c:279-281); the double state equals the reference's bit for bit.
 * VS_ARITH_FMA:   fused multiply-adds in two partial sums; the double state differs in the last
 *                 bits, so the int16 output is not guaranteed identical.  For the reference's
 *                 tables: 0 differences in 1.05e9 samples of BASELINE config 3 and in 6e8 samples
 *                 of the option fuzz, bound +-1 LSB.  For explicit sets the distance follows the
 *                 set's conditioning (+-1 LSB measured for max |A| <= 500; a direct form of order
 *                 40 with coefficients of 1e4 moves a few samples by more).  With the vowel stage's own noise
 *                 (out_snr, vowel -n) the bound is +-2 LSB: the width of a frame's noise follows the frame's power, which a
 *                 sample that moved by one LSB moves in its last bits (tools/fuzz_fma.py: 9.6e9 fuzzed samples, 2 LSB in four
 *                 utterances, all of them with vowel -n). */
/* VS_ARITH_F32:   the recurrence in single precision, two taps per packed multiply-add -- 1.25 x the speed of VS_ARITH_FMA
 *                 for a MEASURED distance from the reference (single precision is marginal against 1e-5):
 *                 per table at gain 10 / pre-emphasis 1, RMS of full scale 4.6e-6 (table 7) .. 1.9e-5 (/i/), above 1e-5
 *                 for /i/, /u/ and table 1; at most 6 LSB there, 25 LSB for table 5 without pre-emphasis; 7.5e-6 over
 *                 BASELINE config 3's mix of tables.  The whole table
 *                 is tests/golden/f32_bounds.json (made by tools/f32_survey.py on the device); tests/test_gpu_f32.py holds
 *                 the kernels to it.  The distance is RELATIVE to the filter's state, i.e. it grows with the gain: over the
 *                 option fuzz's ordinary ranges (gain 1..20) RMS 8.4e-6, at most 61 LSB in 1.2e9 samples; with the corner
 *                 draws (vowel -g 100 and 1000: a state of 1e5..1e6 that the output clips) RMS 2.0e-5..2.9e-5 and single
 *                 unclipped samples off by hundreds of LSB, at most 1986 (tools/fuzz_fma.py ... f32, profiles/r06_f32_mode_measured.txt).  Only the fused wave-specialised kernels have this arithmetic: source-only and
 *                 filter-only launches, the one-wave kernel and coefficient sets of 23..40 taps run VS_ARITH_FMA. */
#define VS_ARITH_EXACT 0
#define VS_ARITH_FMA 1
#define VS_ARITH_F32 2

/* What a plan launch computes. */
#define VS_KIND_SYNTH 0   /* source -> filter, flow never leaves the chip      (fg:246-423 + vw:237-331) */
#define VS_KIND_SOURCE 1  /* source only: int16 glottal flow                   (fg:246-423) */
#define VS_KIND_FILTER 2  /* filter only: int16 flow in, int16 speech out      (vw:237-331) */

typedef struct vs_ctx vs_ctx;
typedef struct vs_plan vs_plan;

/* ---- parameter helpers (host only, no device needed) -------------------------------- */

/* Reference defaults: par initialiser flowgen_shimmer.c:87, vowel_new.c:76-77, vowel 'a'. */
int vs_lane_defaults(vs_lane *lane);

/* nSamples = (unsigned long) par.fs * par.dur, a FLOAT product (flowgen_shimmer.c:242). */
int vs_num_samples(int32_t fs, float dur, uint64_t *n_samples);

/* Row pitch (in samples) that suits the kernels' stores for rows of n_samples: a caller who allocates the PCM buffer
 * [n_lanes][pitch] may pick any pitch >= n_samples, and the choice shows in a full-chip launch -- about 2 % between
 * dense rows of 16000 samples and this pitch, 10-13 % against rows a power of two apart (16384 or 32768 samples, dense).
 * Why: a wavefront's store instruction writes 16 bytes into each of 64 rows at the SAME offset, so the distance between
 * the rows decides how those 64 writes spread over the memory channels (profiles/r05_row_pitch.txt,
 * tools/pitch_probe.py).  Returned: n_samples rounded up to a whole number of 128-byte lines, that number being
 * 3 (mod 4); rows shorter than 2 KiB are only rounded up to 16 bytes.  Every pitch >= n_samples remains VALID
 * (vs_plan_launch takes what it is given); this one avoids the slow ones. */
size_t vs_row_pitch(size_t n_samples);

/* The denominator tables of coefficients(), vowel_new.c:430-633.  A receives 23 doubles. */
int vs_vowel_coefficients(int vowel, double *A);

/* Order of the lane's all-pole filter: VS_ORDER for the ten tables, vs_lane.order (1..VS_MAX_ORDER,
 * 0 = VS_ORDER) for a VS_VOWEL_CUSTOM set; VS_ERR_RANGE beyond MAX_ORDER (vowel_new.c:33). */
int vs_lane_order(const vs_lane *lane, int *order);
/* The label coefficients() prints for the entry ("/a/ JPHS", ...), vowel_new.c:550-622. */
const char *vs_vowel_name(int vowel);

/* Range checks of initialization() (flowgen_shimmer.c:470-546) and of vowel's option loop
 * (vowel_new.c:126-143).  VS_ERR_RANGE where the reference prints usage(); VS_ERR_UNSUPPORTED
 * where the reference would run into undefined behaviour or this engine's limits. */
int vs_lane_validate(const vs_lane *lane);

const char *vs_strerror(int code);

/* ---- command-line surface (host only) ------------------------------------------------ */

typedef struct vs_flowgen_cmd {
  vs_lane lane;
  float dur;            /* -d */
  int wav_arg;          /* argv index of the output file name (arg.wav, fg:140) */
} vs_flowgen_cmd;

typedef struct vs_vowel_cmd {
  float gain, pre_emphasis, snr; /* snr already pow(10, x/10), 0 when -n absent */
  int vowel;
  int input_arg, output_arg, noise_arg;
} vs_vowel_cmd;

/* The option loop + initialization() of flowgen_shimmer.c:128-222, 463-546, without the
 * exit(): VS_USAGE where the reference calls usage(). */
int vs_flowgen_parse(int argc, char **argv, vs_flowgen_cmd *cmd);
/* The option loop of vowel_new.c:116-192. */
int vs_vowel_parse(int argc, char **argv, vs_vowel_cmd *cmd);

/* RIFF header as the reference lays it out (flowgen_shimmer.c:49-63, 550-565).
 * header_bytes is 44 (ILP32 build, the standard layout) or 72 (LP64 build).
 * Returns the number of bytes written into buf (>= 72 must be available) or < 0. */
int vs_wav_header_write(unsigned char *buf, int header_bytes, int32_t fs, float dur);
/* Parses either layout (vowel_new.c:196-205 reads its own struct).  Returns header size. */
int vs_wav_header_read(const unsigned char *buf, size_t avail, int32_t *fs, int *format_tag,
                       int *bits_per_sample, uint64_t *data_bytes);

/* ---- device context ------------------------------------------------------------------ */

int vs_ctx_create(int device, vs_ctx **ctx);
void vs_ctx_destroy(vs_ctx *ctx);
/* Use an existing hipStream_t for all launches of this context (NULL = default stream). */
int vs_ctx_set_stream(vs_ctx *ctx, void *hip_stream);
int vs_ctx_set_arith(vs_ctx *ctx, int arith);
int vs_ctx_last_hip_error(const vs_ctx *ctx);

/* Launch tuning.  All zero (the default) = the library's own choices; the fields exist for
 * measurements (tools/) and tests.  Values are validated here, once, and copied into every plan
 * made afterwards; nothing else can change what a plan launches -- in particular no environment
 * variable does, unless VS_DEBUG_TUNING=1 asks vs_ctx_create() to read the experiment knobs
 * (VS_KERNEL, VS_RING_SLOTS, VS_READY_MIN, VS_WS_PAIRS, VS_WS_ROLES, VS_GEN_LOW, VS_GEN_MIN, VS_WS_PRIO, VS_MIXED_RINGS) through this
 * same function.  NULL resets. */
#define VS_KERNEL_AUTO 0
#define VS_KERNEL_SINGLE 1 /* one wavefront per 64 utterances generates and filters */
#define VS_KERNEL_WS 2     /* wave-specialised: two or three wavefronts per 64 utterances, one job each */
#define VS_FAULT_WITHHOLD_PROGRESS 1 /* tests: the generator wavefront never publishes its progress */
#define VS_FAULT_SHORT_COS_ROWS 2    /* tests: the kernel finds no room for its cos rows (plan and kernel disagree) */
#define VS_FAULT_SHARD_PREPARE 3     /* tests: a context that serves a shard of a node fails to prepare its chunks (vs_node_synth_gather) */
#define VS_FAULT_SHARD_HANDOVER 4    /* tests: ... fails while handing its first chunk over, after the others have started */
#define VS_FAULT_SIMD_DEALING 5      /* tests: plans behave as if vs_ctx_simd_dealing() had found the wavefronts NOT dealt four at a time */
#define VS_FAULT_REROUND 6           /* tests: every seventh super-step of the kernels that take vowel -n's frame powers along rounds its results twice, as if
                                        round2int()'s quirk set had been hit -- the frames concerned must come from the streaming pass instead */
typedef struct vs_tuning {
  int32_t kernel;     /* VS_KERNEL_* */
  int32_t ring_slots; /* LDS ring capacity per utterance in samples (rounded to 24, clamped to what fits) */
  int32_t ready_min;  /* 1..64: a super-step runs when ready lanes * 64 >= live lanes * ready_min */
  int32_t ws_pairs;   /* 1, 2 or 4 generator/filter pairs per workgroup (as many as fit the LDS) */
  int32_t gen_low;    /* >= 24: a lane with fewer buffered samples starts a generator round at once */
  int32_t gen_min;    /* 1..64: a round starts when wanting lanes * 64 >= needing lanes * gen_min */
  int32_t spin_limit; /* polls before a waiting wavefront gives up with VS_ERR_INTERNAL */
  int32_t fault;      /* VS_FAULT_* */
  int32_t ws_filter_prio; /* s_setprio of the filter wavefront: 0 = default (3), 1..3, -1 = leave it at 0 */
  int32_t ws_roles;   /* wavefronts per 64 utterances of the wave-specialised launch: 0 = the library's choice,
                         2 = generator | filter, 3 = open phase | noise | filter (full grids) */
  int32_t mixed_rings; /* batches whose groups differ in period: 0 = the library's choice (a workgroup holds groups from
                          across the period range, each with the ring depth ITS periods need), -1 = never (uniform rings),
                          > 1 = the same with this many slots as the shallowest ring (measurements) */
} vs_tuning;
int vs_ctx_set_tuning(vs_ctx *ctx, const vs_tuning *tuning);
/* Device self-test of the arithmetic shortcuts the kernels take: [0] division shortcut
 * (exhaustive over all 2^31 draws), [1] Philox known answers, [2] integer square root,
 * [3] rounding, [4] one-fma noise sample (exhaustive over the draws at 16 widths), [5] two-block
 * Philox with prepared round keys, [6] workgroups of the wave-to-SIMD probe below that were NOT dealt
 * "wavefront w next to wavefront w % 4" (a performance assumption, not a correctness one -- but on the hardware this
 * library is written for it holds, and a chip where it does not is worth a failed self-test), [7] the output-noise sample of vowel -n
 * (conversion of a draw and the one-instruction rounding, exhaustive over the draws and over every float).  failures
 * (optional) receives VS_SELFTEST_COUNTERS counters; VS_OK if all are zero, else VS_ERR_INTERNAL. */
#define VS_SELFTEST_COUNTERS 8
int vs_ctx_selftest(vs_ctx *ctx, uint64_t *failures);
/* How the hardware deals the wavefronts of a workgroup to the four SIMDs of a compute unit, read from HW_ID by a
 * one-workgroup-per-CU probe launch (once per context, cached): *cyclic12 / *cyclic8 = 1 if in every probed
 * 12- / 8-wavefront workgroup the first four wavefronts ran on four different SIMDs and wavefront w ran on the SIMD
 * of wavefront w % 4 (MI355X: four rotations of the order 0, 2, 1, 3) -- what the three-role layouts of the fused
 * kernel are built on (the three wavefronts of ONE group share a SIMD; on half-filled chips the filter wavefront has one to
 * itself).  Where it does not hold, plans take the two-role kernel instead of running the three roles in an order
 * that is 2.5 x slower (vs_plan_roles says so). */
int vs_ctx_simd_dealing(vs_ctx *ctx, int *cyclic12, int *cyclic8);
/* Name, CU count of the device in use. */
int vs_ctx_device_info(const vs_ctx *ctx, char *name, size_t name_len, int *cu_count);
/* PCI bus id of the device in use ("0000:05:00.0", hipDeviceGetPCIBusId): what tells two devices of a node
 * apart when a multi-GPU run has to show that N DIFFERENT devices took part (bench.py, vs_bench). */
int vs_ctx_device_pci(const vs_ctx *ctx, char *bus_id, size_t len);

/* ---- plans: host preparation once, any number of launches ---------------------------- */

/* Validates the lanes, builds the per-T2 cosine tables with the host libm (flowgen_shimmer.c:
 * 319, 328 call cos() per sample; T2 = ceil(.5*cq*P) is fixed per utterance), and uploads the
 * lane records.  The lanes array may be freed afterwards.
 * Limits (VS_ERR_UNSUPPORTED beyond them): n_lanes < 2^31 - 64, n_samples < 2^31 - 256, periods
 * whose ring does not fit a compute unit's LDS: fs/F0 up to ~930 (with jitter; ~1120 without) runs
 * on the kernels with 64 utterances per wavefront, up to ~3800 (~4500) on the narrow build of the
 * one-wave kernel (16 utterances per wavefront, slow: vs_plan_kernel_name says which).  An utterance's draw stream is
 * indexed with 32 bits: about one draw per sample, so the sample limit keeps it in range for every
 * setting short of rejection loops that retry thousands of times per cycle.
 * Cost (65536 utterances: about 1 ms on the host and 0.2 ms of upload, profiles/r05_plan_cost.txt): batches of 8192
 * utterances and more are expanded by worker threads of the context's own -- up to 15, started with the first such plan,
 * asleep between plans, signals blocked, ended by vs_ctx_destroy -- into page-locked memory, and go up as DMA transfers
 * that run next to a launch that is under way: a caller who synthesises NEW utterances makes the plan of batch k + 1
 * while batch k's kernel runs (for new DRAWS of the same utterances there is vs_plan_reseed).  Like the HIP runtime
 * under it, a context does not survive fork(). */
int vs_plan_create(vs_ctx *ctx, const vs_lane *lanes, size_t n_lanes, size_t n_samples,
                   vs_plan **plan);
/* May be called while launches of the plan are still running: the plan's device blocks (records, tables, the output-noise
 * tables) are not freed -- hipFree waits for the whole device, i.e. for the kernels of the batches behind -- but kept by the
 * context (up to 32 of them: 9 MB per plan of 65536 utterances) and handed to the next plan of that size once the launches
 * that read them are over (an event recorded behind the plan's last launch).  vs_ctx_trim() / vs_ctx_destroy() free them. */
void vs_plan_destroy(vs_plan *plan);

/* Launch on the context's stream; returns without waiting for the device.
 *   in_dev   : VS_KIND_FILTER only, int16 [n_lanes][in_pitch] glottal flow (device pointer)
 *   out_dev  : int16 [n_lanes][out_pitch] (device pointer), out_pitch >= n_samples (vs_row_pitch() names the fast one)
 *   log_dev  : optional vs_cycle_rec [n_lanes][log_pitch] (device pointer) or NULL
 *   ncyc_dev : optional int32 [n_lanes], cycles generated per lane, or NULL */
int vs_plan_launch(vs_plan *plan, int kind, const int16_t *in_dev, size_t in_pitch,
                   int16_t *out_dev, size_t out_pitch, vs_cycle_rec *log_dev, size_t log_pitch,
                   int32_t *ncyc_dev);
int vs_ctx_synchronize(vs_ctx *ctx);
/* Device time between two points of the context's launch stream, for callers who have no HIP of their own (the C programs
 * of this package): vs_ctx_timer_mark(ctx, 0) and (ctx, 1) record an event each behind what has been enqueued so far;
 * vs_ctx_timer_elapsed waits for mark 1 and gives the milliseconds from mark 0 to it. */
int vs_ctx_timer_mark(vs_ctx *ctx, int which);
int vs_ctx_timer_elapsed(vs_ctx *ctx, double *ms);
/* Waits for the context's stream, then reports the health word of the plan's launches:
 * VS_OK, or VS_ERR_INTERNAL if a device-side check failed (*flags, optional, gets the raw bits:
 * 1, 2, 4 = a bounded wait of the generator / filter / noise wavefront ran out, 8 = plan and kernel
 * disagree about the room for the cos rows).  The one-call conveniences below check it themselves.
 * What such a launch has written is NOT the utterances (lanes whose check failed synthesise from whatever
 * the LDS holds): every row of every launch of the plan since the last VS_OK status must be discarded.  The
 * chunked paths (vs_synth_rows, vs_node_synth_gather, vs_node_synth_rows) read the status of a chunk only
 * after it has been delivered, so rows a callback has already seen, or that already lie in the caller's
 * buffer, are to be discarded as well when the call returns VS_ERR_INTERNAL. */
int vs_plan_status(vs_plan *plan, int *flags);
/* The same utterances with NEW draws: replaces every lane's seed (and out_seed: out_seeds may be NULL = the same values)
 * in the plan's device records -- what running the reference's programs again does, which seed from the clock
 * (flowgen_shimmer.c:241, vowel_new.c:234).  seeds[i] belongs to lanes[i] of vs_plan_create, whatever order the plan
 * keeps its records in; the arrays are the caller's again when the call returns.  Stream-ordered with the plan's
 * launches on the context's stream (launches enqueued before see the old seeds, launches after the new ones); 16 bytes
 * per lane go up instead of a whole new plan (65536 lanes: 0.1 ms against 3 ms).  A lane's seed does not change which
 * kernel or ring the plan uses. */
int vs_plan_reseed(vs_plan *plan, const uint64_t *seeds, const uint64_t *out_seeds);

/* Host cost of vs_plan_create(): host_ms = validation, parameter expansion, sorting, cosine
 * tables (cut over up to 8 host threads for batches >= 8192); upload_ms = device allocation,
 * upload and the wait for it.  Neither is part of a launch. */
int vs_plan_timing(const vs_plan *plan, double *host_ms, double *upload_ms);
/* Name of the kernel a launch of this kind runs ("vs_synth_kernel<0, 0, false, true>", ...), as
 * rocprofv3 prints it; for measurement scripts. */
int vs_plan_kernel_name(const vs_plan *plan, int kind, char *buf, size_t len);

/* Dynamic LDS bytes per 64-lane workgroup and launch geometry a plan will use. */
/* Launch shape of the fused kind: *roles = wavefronts per 64 utterances (1 = the one-wave kernel, 2, 3), *layout =
 * 0 role-major / 1 spread (the filter wavefront alone on its SIMD), *simd_fallback = 1 if the plan wanted three
 * roles and took two because vs_ctx_simd_dealing() found the wavefronts dealt differently.  Any pointer may be NULL. */
int vs_plan_roles(const vs_plan *plan, int *roles, int *layout, int *simd_fallback);
int vs_plan_info(const vs_plan *plan, size_t *lds_bytes, size_t *n_workgroups,
                 size_t *ring_slots);

/* ---- host-buffer entry points (plan, launch, deliver) --------------------------------- */

/* fg:246-423 then vw:237-331 for every lane; pcm is int16 [n_lanes][n_samples].
 * The reference writes its samples out cycle by cycle (flowgen_shimmer.c:413-421) and frame by
 * frame (vowel_new.c:327); here the finished rows cross PCIe in 16 MiB blocks while later
 * chunks of the batch are still being synthesised (chunks of 16384 utterances, two device
 * buffers, four DMA workers with pinned staging buffers owned by the context).  If pcm is
 * PINNED host memory (vs_host_alloc, hipHostMalloc, hipHostRegister) the blocks are DMAed straight
 * into it; otherwise each block is copied from its staging buffer into pcm by its worker.  A
 * staging buffer always holds at least one whole row: utterances of more than 8 388 608 samples
 * make the context allocate larger ones (four of them, pinned). */
int vs_synth(vs_ctx *ctx, const vs_lane *lanes, size_t n_lanes, size_t n_samples, int16_t *pcm);

/* The same pipeline with the caller in the place of the memcpy: cb receives `rows` finished
 * consecutive rows starting at `row0` (int16 [rows][n_samples], contiguous) in a pinned staging
 * buffer that is valid only during the call.  cb runs on the library's delivery threads, up to
 * four calls at a time for different blocks, in no particular order; every row is delivered
 * exactly once.  A non-zero return stops the pipeline with VS_ERR_IO.  (vs_batch writes its
 * .wav files from here: header + payload, fwrite after fwrite, as the reference does.) */
typedef int (*vs_rows_cb)(void *user, size_t row0, size_t rows, const int16_t *pcm);
int vs_synth_rows(vs_ctx *ctx, const vs_lane *lanes, size_t n_lanes, size_t n_samples,
                  vs_rows_cb cb, void *user);

/* Pinned host memory for callers without a HIP binding of their own. */
int vs_host_alloc(vs_ctx *ctx, size_t bytes, void **ptr);
int vs_host_free(vs_ctx *ctx, void *ptr);
/* Releases the buffers the context keeps between calls (device PCM chunks, staging, streams, the device blocks of
 * destroyed plans). */
int vs_ctx_trim(vs_ctx *ctx);
/* fg:246-423; flow is int16 [n_lanes][n_samples].  recs/ncyc optional (NULL). */
int vs_source(vs_ctx *ctx, const vs_lane *lanes, size_t n_lanes, size_t n_samples, int16_t *flow,
              vs_cycle_rec *recs, size_t recs_pitch, int32_t *ncyc);
/* vw:237-331; only gain, pre_emphasis, vowel/A, out_snr/out_seed and fs (frame length of the
 * output noise) of each lane are used; the source fields are not even validated, so a flow of
 * any sample rate can be filtered (vowel_new.c:196-205). */
int vs_filter(vs_ctx *ctx, const vs_lane *lanes, size_t n_lanes, size_t n_samples,
              const int16_t *flow, int16_t *pcm);

/* ---- one batch over the GPUs of a node --------------------------------------------- */

/* Utterances are independent (all carried state of the reference is per utterance:
 * flowgen_shimmer.c:121-122, vowel_new.c:90), so a batch shards as contiguous blocks of lanes,
 * block s of S (blocks differ by at most one lane, vs_node_shard_range), with no data-path collective; a lane's draws
 * are keyed by the seed in its own record, so S shards give byte for byte what one device
 * gives.  devices[] lists one device per shard; a device may appear more than once ("logical
 * shards": how the N-device path is exercised on one GPU).  devices[0] is the root.  One
 * context, two streams and one host thread per shard; calls on a node are not re-entrant. */
typedef struct vs_node vs_node;
int vs_node_create(const int *devices, int n_shards, vs_node **node);
void vs_node_destroy(vs_node *node);
int vs_node_shards(const vs_node *node);
int vs_node_ctx(vs_node *node, int shard, vs_ctx **ctx); /* e.g. for vs_ctx_set_tuning */
int vs_node_set_arith(vs_node *node, int arith);
int vs_node_shard_range(const vs_node *node, size_t n_lanes, int shard, size_t *lo, size_t *hi);
/* How vs_node_synth_gather moves a finished chunk into the root's memory.
 *   VS_NODE_TRANSPORT_PEER (default): peer DMA, one copy stream per shard.
 *   VS_NODE_TRANSPORT_RCCL (EXPERIMENTAL until a multi-GPU node has run it: what one GPU can exercise -- the
 *     communicator, the all-or-nothing start, the abort path -- is tested; a send and a receive between two devices
 *     are not): ncclSend / ncclRecv on ONE RCCL communicator over the node's devices,
 *     created here and owned by the node (librccl is opened with dlopen at this call).  Needs every
 *     shard on a device of its own (VS_ERR_UNSUPPORTED otherwise, or when librccl is not there) and a
 *     packed root buffer (root_pitch == n_samples).
 *     The exchange is all or nothing (every shard prepares all of its chunks before anybody enqueues anything); a
 *     failure behind that point aborts the communicators (ncclCommAbort) and leaves the node on the peer transport.
 * vs_node_link(): how shard's PCM reaches the root -- VS_NODE_LINK_SELF (same device, in place),
 * _PEER (peer DMA), _STAGED (no peer access between the two devices: the copies go through host
 * memory), _RCCL.  vs_node_last_rccl_error(): the ncclResult_t of the last failing RCCL call.
 * vs_node_rccl_ranks(): ncclCommCount of the shard's communicator -- the number of ranks RCCL itself says it spans (the
 * node's shard count on a healthy node), 0 while the node is on the peer transport, < 0 on error: what a pre-flight prints
 * next to the PCI bus ids (cli/vs_bench.c --gpus N --rccl).  Switching to the RCCL transport ends with a link check: 64 KiB
 * from the root's communicator to itself through the entry points the gather uses (group, receive, send), compared byte
 * for byte; a library that fails it is refused (the ncclResult_t in vs_node_last_rccl_error, or VS_ERR_INTERNAL for bytes
 * that differ) and the node stays on peer copies. */
#define VS_NODE_TRANSPORT_PEER 0
#define VS_NODE_TRANSPORT_RCCL 1
#define VS_NODE_LINK_SELF 0
#define VS_NODE_LINK_PEER 1
#define VS_NODE_LINK_STAGED 2
#define VS_NODE_LINK_RCCL 3
int vs_node_set_transport(vs_node *node, int transport);
int vs_node_link(const vs_node *node, int shard);
int vs_node_last_rccl_error(const vs_node *node);
int vs_node_rccl_ranks(vs_node *node, int shard);
/* Synthesis with the final PCM gathered into the ROOT device's memory (root_dev: int16
 * [n_lanes][root_pitch] on devices[0]).  Every shard works through its block in chunks of 16384
 * utterances; with VS_NODE_OVERLAP a finished chunk travels to its rows of root_dev by a peer
 * DMA (one transfer stream per shard = per xGMI link into the root, no ring) while the shard's
 * next chunk is being synthesised; without it copies and kernels alternate (the comparison
 * case).  Shards on the root device are synthesised in place unless VS_NODE_STAGE_ALL sends them
 * through the chunk buffers and the copy too (tests of the transfer path on one GPU).
 * total_ms: host clock over the whole call; max_compute_ms: the slowest shard from its first
 * launch to its last kernel's end.  Both optional. */
#define VS_NODE_OVERLAP 1
#define VS_NODE_STAGE_ALL 2
int vs_node_synth_gather(vs_node *node, const vs_lane *lanes, size_t n_lanes, size_t n_samples,
                         int16_t *root_dev, size_t root_pitch, int flags, double *total_ms,
                         double *max_compute_ms);
/* Synthesis with host delivery: every shard runs the vs_synth_rows() pipeline over its own PCIe
 * link; cb sees global row numbers (and is called from up to 4 threads per shard). */
int vs_node_synth_rows(vs_node *node, const vs_lane *lanes, size_t n_lanes, size_t n_samples,
                       vs_rows_cb cb, void *user);

/* Raw device memory for callers without a HIP binding of their own (the CLIs). */
int vs_dev_alloc(vs_ctx *ctx, size_t bytes, void **ptr);
int vs_dev_free(vs_ctx *ctx, void *ptr);
int vs_dev_upload(vs_ctx *ctx, void *dst_dev, const void *src_host, size_t bytes);
int vs_dev_download(vs_ctx *ctx, void *dst_host, const void *src_dev, size_t bytes);

/* Library version string. */
const char *vs_version(void);

#ifdef __cplusplus
}
#endif
#endif /* VOICE_SYNTH_H */
