/*
 * vs_host.c -- host-side C of the engine: parameter records, the two command-line parsers,
 * validation and the RIFF header.  No device code, no HIP calls.
 *
 * Mirrors the reference's front matter for the hot path:
 *   flowgen_shimmer.c:73-102 (PAR/ARG), :128-222 (option loop), :463-565 (initialization()),
 *   vowel_new.c:76-77, :116-192 (option loop), :45-59 (header struct).
 * Must be compiled without floating-point contraction (Makefile passes -ffp-contract=off):
 * the float/double conversions below are part of the parity contract.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/voice_synth.h"
#include "vs_tables.h"

#define VS_VERSION_STRING "voice_synth_amd 0.1 (gfx950)"

const char *vs_version(void) { return VS_VERSION_STRING; }

const char *vs_strerror(int code)
{
  switch (code) {
    case VS_OK: return "ok";
    case VS_ERR_ARG: return "invalid argument";
    case VS_ERR_RANGE: return "parameter outside the range the reference accepts";
    case VS_ERR_UNSUPPORTED: return "parameter combination undefined in the reference or beyond engine limits";
    case VS_ERR_HIP: return "HIP runtime error";
    case VS_ERR_NOMEM: return "out of memory";
    case VS_ERR_NODEVICE: return "no usable gfx950 device (there is no CPU path)";
    case VS_ERR_IO: return "I/O error";
    case VS_USAGE: return "usage";
    case VS_ERR_INTERNAL: return "device-side wait ran out (internal error)";
    default: return "unknown error";
  }
}

/* par initialiser, flowgen_shimmer.c:87; pre_emphasis/gain, vowel_new.c:76-77 */
int vs_lane_defaults(vs_lane *lane)
{
  if (!lane) return VS_ERR_ARG;
  memset(lane, 0, sizeof(*lane));
  lane->jitter = .0f;
  lane->cq = 0.55f;
  lane->K = 0.65f;
  lane->Fg = 125;
  lane->F0 = 120;
  lane->DC = 0.0f;
  lane->noise = 0.0f;
  lane->fs = 22050;
  lane->amp = 12000;
  lane->Kvar = 0.0f;
  lane->shimmer = 0.0f;
  lane->flags = 0;
  lane->seed = 0;
  lane->gain = 10.0f;
  lane->pre_emphasis = 1.0f;
  lane->vowel = 'a';
  lane->A[0] = 1.0;
  return VS_OK;
}

/* flowgen_shimmer.c:242: nSamples = (unsigned long) par.fs*par.dur -- the cast binds to
 * par.fs only, the product is a float */
int vs_num_samples(int32_t fs, float dur, uint64_t *n_samples)
{
  if (!n_samples || fs <= 0) return VS_ERR_ARG;
  /* the same float product; a negative, infinite or NaN one (or one beyond 2^63) is undefined in the
   * reference's conversion -- the parser lets "-d inf" pass as the reference's does, the engine stops here */
  const float prod = (unsigned long)fs * dur;
  if (!(prod >= 0.0f) || !(prod < 9.0e18f)) return VS_ERR_UNSUPPORTED;
  unsigned long n = (unsigned long)fs * dur;
  *n_samples = (uint64_t)n;
  return VS_OK;
}

/* the row pitch the kernels' stores like (include/voice_synth.h; measured: profiles/r05_row_pitch.txt) */
size_t vs_row_pitch(size_t n_samples)
{
  if (n_samples > ((size_t)-1) / 4) return n_samples;   /* no room to round: the caller's rows as they are */
  const size_t bytes = n_samples * sizeof(int16_t);
  if (bytes < 2048) return (n_samples + 7) & ~(size_t)7; /* short rows: 16-byte alignment only */
  size_t lines = (bytes + 127) / 128;
  while ((lines & 3) != 3) lines++;
  return lines * (128 / sizeof(int16_t));
}

/* taps of the lane's filter: 22 for the tables (vowel_new.c:172), vs_lane.order for an explicit set */
int vs_lane_order(const vs_lane *lane, int *order)
{
  if (!lane || !order) return VS_ERR_ARG;
  *order = VS_ORDER;
  if (lane->vowel != VS_VOWEL_CUSTOM) return VS_OK;
  if (lane->order < 0 || lane->order > VS_MAX_ORDER) return VS_ERR_RANGE; /* MAX_ORDER, vowel_new.c:33 */
  if (lane->order > 0) *order = lane->order;
  return VS_OK;
}

int vs_vowel_coefficients(int vowel, double *A)
{
  if (!A) return VS_ERR_ARG;
  for (int t = 0; t < VS_TAB_NTABLES; t++) {
    if (vs_tab_ids[t] == (char)vowel) {
      memcpy(A, vs_tab_A[t], sizeof(double) * VS_NCOEF);
      return VS_OK;
    }
  }
  return VS_ERR_RANGE;
}

/* the ten tables by number, 0..9 in the order of the table ids (a i u 1..7): the rows of a plan's tap table */
int vs_vowel_index(int vowel)
{
  for (int t = 0; t < VS_TAB_NTABLES; t++)
    if (vs_tab_ids[t] == (char)vowel) return t;
  return -1;
}
int vs_vowel_by_index(int index) { return (index >= 0 && index < VS_TAB_NTABLES) ? (int)vs_tab_ids[index] : 0; }

const char *vs_vowel_name(int vowel)
{
  for (int t = 0; t < VS_TAB_NTABLES; t++)
    if (vs_tab_ids[t] == (char)vowel) return vs_tab_names[t];
  return NULL;
}

int vs_lane_validate(const vs_lane *lane)
{
  if (!lane) return VS_ERR_ARG;
  /* flowgen initialization(), in the reference's order (fg:476-546) */
  if (!(lane->jitter >= 0.0 && lane->jitter <= 10.0)) return VS_ERR_RANGE;   /* fg:478 */
  if (!(lane->K >= 0.50)) return VS_ERR_RANGE;                                /* fg:484 */
  if (!(lane->cq >= 0.0 && lane->cq <= 1.0)) return VS_ERR_RANGE;            /* fg:490 */
  if (!(lane->Fg >= 50)) return VS_ERR_RANGE;                                 /* fg:496 */
  if (!((lane->F0 >= 50) && (lane->F0 < lane->Fg))) return VS_ERR_RANGE;      /* fg:504 */
  if (lane->flags & VS_FLAG_NOISE) {
    if (!(lane->noise >= 1.0f && lane->noise <= 100000.0f)) return VS_ERR_RANGE; /* 0..50 dB, fg:510 */
  }
  if (!(lane->amp >= 0 && lane->amp < 32767)) return VS_ERR_RANGE;            /* fg:518 */
  if (!(lane->DC >= 0)) return VS_ERR_RANGE;                                  /* fg:524 */
  if (!(lane->Kvar >= 0 && lane->Kvar <= 1)) return VS_ERR_RANGE;             /* fg:530 */
  if (lane->fs <= 0) return VS_ERR_RANGE;
  if (!(lane->shimmer >= 0 && lane->shimmer <= 1)) return VS_ERR_RANGE;       /* fg:544 */
  /* vowel option loop */
  if (!(lane->pre_emphasis >= 0.0 && lane->pre_emphasis <= 1.0)) return VS_ERR_RANGE; /* vw:127 */
  if (!(lane->gain >= 1)) return VS_ERR_RANGE;                                        /* vw:132 */
  if (lane->vowel != VS_VOWEL_CUSTOM) {
    if (vs_vowel_index(lane->vowel) < 0) { /* not one of the ten tables */
      /* upper-case A/I/U pass the reference's check but load no coefficients */
      if (lane->vowel == 'A' || lane->vowel == 'I' || lane->vowel == 'U') return VS_ERR_UNSUPPORTED;
      return VS_ERR_RANGE;
    }
  } else {
    int order = 0;
    const int rc = vs_lane_order(lane, &order);
    if (rc != VS_OK) return rc;
    for (int j = 0; j <= order; j++)
      if (!isfinite(lane->A[j])) return VS_ERR_RANGE;
    if (lane->A[0] != 1.0) return VS_ERR_RANGE;
  }
  /* places where the reference computes garbage (NaN widths, zero-length pulses) */
  int P = (int)((float)lane->fs / lane->F0);                                  /* fg:244 */
  if (P < 2) return VS_ERR_UNSUPPORTED;
  int T2 = (int)ceil(0.5 * lane->cq * P);                                     /* fg:317 */
  if (T2 < 1) return VS_ERR_UNSUPPORTED; /* cq == 0: no pulse, x_pow = 0/0 with -n */
  return VS_OK;
}

/* ------------------------------------------------------------------------------------------
 * The two command lines.  Both programs of the reference take "-x value" pairs, look at the first
 * letter behind the dash only, in either case (flowgen_shimmer.c:130-218, vowel_new.c:118-190), and
 * answer anything else with usage().  Here each program is a TABLE of its options -- letter, how
 * the text becomes a number, the range the reference accepts, where the value goes -- and one
 * scanner walks the argument vector for both.  What differs between the two programs is WHEN a value
 * is looked at, and the tables say so:
 *   flowgen_shimmer only notes where each value stands while it scans (a later -d overrides an
 *     earlier one unseen) and converts afterwards, in the fixed order of its initialization()
 *     (fg:470-546) -- which is the order of vs_flowgen_opts[], and matters: F0 is held against the
 *     Fg of the same command line, the DC flow is a fraction of its amplitude;
 *   vowel converts and checks every value where it stands (vw:123-187): "-g 0 -g 5" is refused.
 * tests/test_parser_vs_reference.py holds both against the compiled reference on random lines.
 * ---------------------------------------------------------------------------------------- */
enum { VS_CV_NONE, VS_CV_FLOAT, VS_CV_PERCENT, VS_CV_INT, VS_CV_LONG, VS_CV_VOWEL };
enum {
  VS_TO_NOTHING, VS_TO_DUR, VS_TO_JITTER, VS_TO_K, VS_TO_CQ, VS_TO_FG, VS_TO_F0, VS_TO_NOISE_DB, VS_TO_AMP,
  VS_TO_DC_FRACTION, VS_TO_KVAR, VS_TO_FS, VS_TO_SHIMMER_PERCENT, VS_TO_PRE, VS_TO_GAIN, VS_TO_SNR_DB, VS_TO_VOWEL
};
#define VS_OPT_BELOW_FG 0x1    /* upper bound: strictly below the Fg of the same command line (fg:504) */
#define VS_OPT_NOT_22050 0x2   /* every rate but an explicit 22050 (fg:537) */
#define VS_OPT_DC_QUARTER 0x4  /* seeing the option sets the DC flow to .25 at once (fg:182) */
#define VS_OPT_REJECTS 0x8     /* the reference tests "outside -> usage()" (a NaN passes) instead of
                                  "inside -> accept" (a NaN fails): vowel's -p, -g, -n */
#define VS_OPT_LO_OPEN 0x10    /* the lower bound itself is refused (vowel -n: snr <= 0, vw:142) */
typedef struct vs_opt {
  char letter;         /* lower case */
  unsigned char conv;  /* VS_CV_* */
  unsigned char to;    /* VS_TO_* */
  unsigned char how;   /* VS_OPT_* */
  double lo, hi;       /* the accepted range, both ends included unless `how` says otherwise */
  uint32_t lane_flag;  /* VS_FLAG_* raised when the option is given */
} vs_opt;

/* "no upper bound" is +infinity, not a large number: the reference only tests the lower bound of these
 * (if(f < 0.5) usage(), flowgen_shimmer.c:472, 484, 496; if(gain < 1) usage(), vowel_new.c:132, 142), and a
 * value that overflows its float -- "1e39", "inf" -- passes there, so it must pass here: f <= HUGE_VAL. */
#define VS_NO_LIMIT HUGE_VAL
static const vs_opt vs_flowgen_opts[] = {
    /* in the order initialization() converts them, fg:470-546 */
    {'d', VS_CV_FLOAT, VS_TO_DUR, 0, 0.5, VS_NO_LIMIT, 0},                           /* fg:470-474 */
    {'j', VS_CV_PERCENT, VS_TO_JITTER, 0, 0.0, 10.0, VS_FLAG_JITTER},                /* fg:476-480 */
    {'k', VS_CV_FLOAT, VS_TO_K, 0, 0.50, VS_NO_LIMIT, 0},                            /* fg:482-486 */
    {'c', VS_CV_FLOAT, VS_TO_CQ, 0, 0.0, 1.0, 0},                                    /* fg:488-492 */
    {'g', VS_CV_FLOAT, VS_TO_FG, 0, 50, VS_NO_LIMIT, 0},                             /* fg:494-498 */
    {'f', VS_CV_FLOAT, VS_TO_F0, VS_OPT_BELOW_FG, 50, 0, 0},                         /* fg:500-506 */
    {'n', VS_CV_FLOAT, VS_TO_NOISE_DB, VS_OPT_DC_QUARTER, 0.0, 50, VS_FLAG_NOISE},   /* fg:508-514 */
    {'a', VS_CV_INT, VS_TO_AMP, 0, 0, 32766, 0},                                     /* fg:516-520 */
    {'l', VS_CV_FLOAT, VS_TO_DC_FRACTION, 0, 0, 0.3, 0},                             /* fg:522-526 */
    {'z', VS_CV_FLOAT, VS_TO_KVAR, 0, 0, 1, 0},                                      /* fg:528-532 */
    {'r', VS_CV_LONG, VS_TO_FS, VS_OPT_NOT_22050, 0, 0, 0},                          /* fg:534-540 */
    {'s', VS_CV_FLOAT, VS_TO_SHIMMER_PERCENT, 0, 0, 100, VS_FLAG_SHIMMER},           /* fg:542-546 */
    {'o', VS_CV_NONE, VS_TO_NOTHING, 0, 0, 0, 0},                                    /* the file name */
};
static const vs_opt vs_vowel_opts[] = {
    {'p', VS_CV_FLOAT, VS_TO_PRE, VS_OPT_REJECTS, 0.0, 1.0, 0},                          /* vw:125-128 */
    {'g', VS_CV_FLOAT, VS_TO_GAIN, VS_OPT_REJECTS, 1, VS_NO_LIMIT, 0},                   /* vw:130-133 */
    {'n', VS_CV_FLOAT, VS_TO_SNR_DB, VS_OPT_REJECTS | VS_OPT_LO_OPEN, 0, VS_NO_LIMIT, 0}, /* vw:139-144 */
    {'v', VS_CV_VOWEL, VS_TO_VOWEL, 0, 0, 0, 0},                                         /* vw:153-178 */
    {'i', VS_CV_NONE, VS_TO_NOTHING, 0, 0, 0, 0},
    {'o', VS_CV_NONE, VS_TO_NOTHING, 0, 0, 0, 0},
};
#define VS_COUNT(t) ((int)(sizeof(t) / sizeof((t)[0])))

/* everything a value can be stored into */
typedef struct vs_opt_sink {
  vs_lane *lane;
  float *dur, *pre, *gain, *snr;
  int32_t *vowel;
} vs_opt_sink;

/* One value through its rule: conversion with the reference's own types (the reference converts
 * into a `float f`, so ranges are tested on the ROUNDED value: "-l 0.3" is 0.3f > 0.3 and refused),
 * range, store.  VS_OK or VS_USAGE. */
static int vs_opt_apply(const vs_opt *o, const char *text, const vs_opt_sink *k)
{
  vs_lane *par = k->lane;
  if (o->conv == VS_CV_NONE) return VS_OK;
  if (o->conv == VS_CV_VOWEL) { /* vw:155-171: the first character, one of these */
    if (text[0] == '\0' || !strchr("iauIAU1234567", text[0])) return VS_USAGE;
    *k->vowel = (int)text[0];
    return VS_OK;
  }
  if (o->conv == VS_CV_INT) {
    const int v = atoi(text);
    if (!(v >= (int)o->lo && v <= (int)o->hi)) return VS_USAGE;
    par->amp = v;
    return VS_OK;
  }
  if (o->conv == VS_CV_LONG) {
    const long v = atol(text);
    if ((o->how & VS_OPT_NOT_22050) && v == 22050L) return VS_USAGE;
    par->fs = (int32_t)v;
    return VS_OK;
  }
  const float f = (o->conv == VS_CV_PERCENT) ? (float)(atof(text) / 100.0) : (float)atof(text);
  int inside;
  if (o->how & VS_OPT_REJECTS) {
    const int outside = ((o->how & VS_OPT_LO_OPEN) ? (f <= o->lo) : (f < o->lo)) || (f > o->hi);
    inside = !outside;
  } else {
    inside = (f >= o->lo) && ((o->how & VS_OPT_BELOW_FG) ? (f < par->Fg) : (f <= o->hi));
  }
  if (!inside) return VS_USAGE;
  switch (o->to) {
    case VS_TO_DUR: *k->dur = f; break;
    case VS_TO_JITTER: par->jitter = f; break;
    case VS_TO_K: par->K = f; break;
    case VS_TO_CQ: par->cq = f; break;
    case VS_TO_FG: par->Fg = f; break;
    case VS_TO_F0: par->F0 = f; break;
    case VS_TO_NOISE_DB: par->noise = pow(10, f / 10); break;      /* fg:511 */
    case VS_TO_DC_FRACTION: par->DC = f * par->amp; break;         /* fg:524: of the amplitude set above */
    case VS_TO_KVAR: par->Kvar = f; break;
    case VS_TO_SHIMMER_PERCENT: par->shimmer = f / 100; break;     /* fg:544 */
    case VS_TO_PRE: *k->pre = f; break;
    case VS_TO_GAIN: *k->gain = f; break;
    case VS_TO_SNR_DB: *k->snr = pow(10, f / 10); break;           /* vw:143 */
    default: break;
  }
  par->flags |= o->lane_flag;
  return VS_OK;
}

/* The scanner: "-x value" pairs from argv[1] on, first letter behind the dash, either case.
 * at[r] receives the argv index of the LAST value given for rule r (-1: not given).  on_sight:
 * every value goes through its rule where it stands (vowel); otherwise only the parse-time side
 * effects happen here (flowgen's -n).  A trailing word is tolerated if it starts with 'i', as in
 * both programs (fg:219, vw:191).  VS_OK or VS_USAGE. */
static int vs_opt_scan(int argc, char **argv, const vs_opt *tab, int n_opts, int on_sight, const vs_opt_sink *k, int *at)
{
  for (int r = 0; r < n_opts; r++) at[r] = -1;
  if (argc < 2) return VS_USAGE;
  int i = 1;
  while (i < argc && argv[i][0] == '-') {
    if (i + 1 >= argc) return VS_USAGE; /* an option without its value */
    char c = argv[i][1];
    if (c >= 'A' && c <= 'Z') c = (char)(c - 'A' + 'a');
    int r = 0;
    while (r < n_opts && tab[r].letter != c) r++;
    if (r == n_opts) return VS_USAGE;
    at[r] = i + 1;
    if (tab[r].how & VS_OPT_DC_QUARTER) k->lane->DC = .25;
    if (on_sight && vs_opt_apply(&tab[r], argv[i + 1], k) != VS_OK) return VS_USAGE;
    i += 2;
  }
  if (i != argc && argv[i][0] != 'i') return VS_USAGE;
  return VS_OK;
}

static int vs_opt_index(const vs_opt *tab, int n_opts, char letter)
{
  for (int r = 0; r < n_opts; r++)
    if (tab[r].letter == letter) return r;
  return -1;
}

int vs_flowgen_parse(int argc, char **argv, vs_flowgen_cmd *cmd)
{
  if (!cmd || !argv) return VS_ERR_ARG;
  vs_lane *par = &cmd->lane;
  vs_lane_defaults(par);
  cmd->dur = 1.0f;
  cmd->wav_arg = -1;
  const vs_opt_sink sink = {par, &cmd->dur, NULL, NULL, NULL, NULL};
  int at[VS_COUNT(vs_flowgen_opts)];
  if (vs_opt_scan(argc, argv, vs_flowgen_opts, VS_COUNT(vs_flowgen_opts), 0, &sink, at) != VS_OK) return VS_USAGE;
  const int o = vs_opt_index(vs_flowgen_opts, VS_COUNT(vs_flowgen_opts), 'o');
  if (at[o] == -1) return VS_USAGE; /* fg:219: no output file */
  for (int r = 0; r < VS_COUNT(vs_flowgen_opts); r++)
    if (at[r] != -1 && vs_opt_apply(&vs_flowgen_opts[r], argv[at[r]], &sink) != VS_OK) return VS_USAGE;
  cmd->wav_arg = at[o];
  return VS_OK;
}

int vs_vowel_parse(int argc, char **argv, vs_vowel_cmd *cmd)
{
  if (!cmd || !argv) return VS_ERR_ARG;
  cmd->gain = 10.0f;
  cmd->pre_emphasis = 1.0f;
  cmd->snr = 0.0f;
  cmd->vowel = 0;
  cmd->input_arg = cmd->output_arg = cmd->noise_arg = -1;
  vs_lane unused; /* no vowel option touches a lane; the sink wants one for the rules that do */
  memset(&unused, 0, sizeof(unused));
  const vs_opt_sink sink = {&unused, NULL, &cmd->pre_emphasis, &cmd->gain, &cmd->snr, &cmd->vowel};
  int at[VS_COUNT(vs_vowel_opts)];
  if (vs_opt_scan(argc, argv, vs_vowel_opts, VS_COUNT(vs_vowel_opts), 1, &sink, at) != VS_OK) return VS_USAGE;
  cmd->input_arg = at[vs_opt_index(vs_vowel_opts, VS_COUNT(vs_vowel_opts), 'i')];
  cmd->output_arg = at[vs_opt_index(vs_vowel_opts, VS_COUNT(vs_vowel_opts), 'o')];
  cmd->noise_arg = at[vs_opt_index(vs_vowel_opts, VS_COUNT(vs_vowel_opts), 'n')];
  if (cmd->input_arg == -1 || at[vs_opt_index(vs_vowel_opts, VS_COUNT(vs_vowel_opts), 'v')] == -1) return VS_USAGE; /* vw:191 */
  return VS_OK;
}

/* ------------------------------------------------------------------------------------------
 * RIFF header: struct at flowgen_shimmer.c:49-63, values fg:550-565
 * ---------------------------------------------------------------------------------------- */
static void put_le(unsigned char *p, uint64_t v, int nbytes)
{
  for (int b = 0; b < nbytes; b++) p[b] = (unsigned char)(v >> (8 * b));
}
static uint64_t get_le(const unsigned char *p, int nbytes)
{
  uint64_t v = 0;
  for (int b = 0; b < nbytes; b++) v |= (uint64_t)p[b] << (8 * b);
  return v;
}

int vs_wav_header_write(unsigned char *buf, int header_bytes, int32_t fs, float dur)
{
  if (!buf || (header_bytes != 44 && header_bytes != 72)) return VS_ERR_ARG;
  long par_fs = fs;
  unsigned long datasize = (long int)(dur * par_fs * 2); /* fg:555, float arithmetic */
  long filesize = (long int)datasize + 44L - 8L;         /* fg:556 */
  unsigned short nBlockAlign = (int)(16 / 8 * 1);
  unsigned long nAvg = (long int)(nBlockAlign * (unsigned long)par_fs);
  memset(buf, 0, 72);
  if (header_bytes == 44) {
    /* ILP32 layout: long = 4 bytes, no padding */
    memcpy(buf + 0, "RIFF", 4);
    put_le(buf + 4, (uint64_t)filesize, 4);
    memcpy(buf + 8, "WAVE", 4);
    memcpy(buf + 12, "fmt ", 4);
    put_le(buf + 16, 16, 4);
    put_le(buf + 20, 1, 2);
    put_le(buf + 22, 1, 2);
    put_le(buf + 24, (uint64_t)par_fs, 4);
    put_le(buf + 28, nAvg, 4);
    put_le(buf + 32, nBlockAlign, 2);
    put_le(buf + 34, 16, 2);
    memcpy(buf + 36, "data", 4);
    put_le(buf + 40, datasize, 4);
    return 44;
  }
  /* LP64 layout: long = 8 bytes, aligned to 8 */
  memcpy(buf + 0, "RIFF", 4);
  put_le(buf + 8, (uint64_t)filesize, 8);
  memcpy(buf + 16, "WAVE", 4);
  memcpy(buf + 20, "fmt ", 4);
  put_le(buf + 24, 16, 8);
  put_le(buf + 32, 1, 2);
  put_le(buf + 34, 1, 2);
  put_le(buf + 40, (uint64_t)par_fs, 8);
  put_le(buf + 48, nAvg, 8);
  put_le(buf + 56, nBlockAlign, 2);
  put_le(buf + 58, 16, 2);
  memcpy(buf + 60, "data", 4);
  put_le(buf + 64, datasize, 8);
  return 72;
}

int vs_wav_header_read(const unsigned char *buf, size_t avail, int32_t *fs, int *format_tag,
                       int *bits_per_sample, uint64_t *data_bytes)
{
  if (!buf || avail < 44) return VS_ERR_IO;
  if (memcmp(buf, "RIFF", 4) != 0) return VS_ERR_IO;
  if (memcmp(buf + 8, "WAVE", 4) == 0) {
    if (format_tag) *format_tag = (int)(int16_t)get_le(buf + 20, 2);
    if (fs) *fs = (int32_t)get_le(buf + 24, 4);
    if (bits_per_sample) *bits_per_sample = (int)get_le(buf + 34, 2);
    if (data_bytes) *data_bytes = get_le(buf + 40, 4);
    return 44;
  }
  if (avail >= 72 && memcmp(buf + 16, "WAVE", 4) == 0) {
    if (format_tag) *format_tag = (int)(int16_t)get_le(buf + 32, 2);
    if (fs) *fs = (int32_t)get_le(buf + 40, 8);
    if (bits_per_sample) *bits_per_sample = (int)get_le(buf + 58, 2);
    if (data_bytes) *data_bytes = get_le(buf + 64, 8);
    return 72;
  }
  return VS_ERR_IO;
}

/* ------------------------------------------------------------------------------------------
 * The bookkeeping of a node's gather (vs_node_synth_gather, csrc/vs_node.c), without any device in
 * it: which lanes a shard owns, and which of its rows travel to the root in which round.  The sender
 * (a shard working through its chunks) and the receiver (the root posting one group of receives per
 * round) BOTH walk these functions, so the two sides of the exchange agree by construction;
 * tests/test_host_logic.py goes through ragged cuts, more shards than chunks and empty shards.
 * Utterances are independent (flowgen_shimmer.c:121-122, vowel_new.c:90): any cut is a valid cut.
 * ---------------------------------------------------------------------------------------- */
/* lanes [*lo, *hi) of shard `shard` of `n_shards`: contiguous blocks that differ by at most one lane
 * (the same cut voice_synth_amd/configs.py::shard_range makes for the one-process-per-GPU path) */
int vs_shard_cut(size_t n_lanes, int n_shards, int shard, size_t *lo, size_t *hi)
{
  if (!lo || !hi || n_shards <= 0 || shard < 0 || shard >= n_shards) return VS_ERR_ARG;
  const size_t S = (size_t)n_shards, s = (size_t)shard;
  const size_t base = n_lanes / S, rem = n_lanes % S;
  *lo = s * base + (s < rem ? s : rem);
  *hi = *lo + base + (s < rem ? 1 : 0);
  return VS_OK;
}
/* rounds a gather in chunks of `chunk` utterances takes: the chunks of the LARGEST shard */
size_t vs_gather_rounds(size_t n_lanes, int n_shards, size_t chunk)
{
  if (n_shards <= 0 || chunk == 0) return 0;
  size_t lo, hi;
  vs_shard_cut(n_lanes, n_shards, 0, &lo, &hi); /* shard 0 is never smaller than another */
  return (hi - lo + chunk - 1) / chunk;
}
/* rows [*row0, *row0 + *rows) of the batch that shard `shard` hands over in round `round`; *rows == 0: none
 * (the shard has fewer chunks than there are rounds, or no lanes at all) */
int vs_gather_round(size_t n_lanes, int n_shards, int shard, size_t chunk, size_t round, size_t *row0, size_t *rows)
{
  size_t lo, hi;
  if (!row0 || !rows || chunk == 0) return VS_ERR_ARG;
  const int rc = vs_shard_cut(n_lanes, n_shards, shard, &lo, &hi);
  if (rc != VS_OK) return rc;
  *row0 = lo;
  *rows = 0;
  if (round > (hi - lo) / chunk) return VS_OK;   /* keeps round * chunk from wrapping */
  const size_t a = lo + round * chunk;
  if (a >= hi) return VS_OK;
  *row0 = a;
  *rows = (hi - a < chunk) ? (hi - a) : chunk;
  return VS_OK;
}
