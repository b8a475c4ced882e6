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
  unsigned long n = (unsigned long)fs * dur;
  *n_samples = (uint64_t)n;
  return VS_OK;
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
    double A[VS_NCOEF];
    if (vs_vowel_coefficients(lane->vowel, A) != VS_OK) {
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
 * flowgen_shimmer command line: option loop fg:128-219, initialization() fg:463-546
 * ---------------------------------------------------------------------------------------- */
int vs_flowgen_parse(int argc, char **argv, vs_flowgen_cmd *cmd)
{
  /* struct ARG, fg:90-102 */
  int a_wav = -1, a_dur = -1, a_jitter = -1, a_cq = -1, a_K = -1, a_Fg = -1, a_F0 = -1,
      a_DC = -1, a_noise = -1, a_fs = -1, a_amp = -1, a_Kvar = -1, a_Shimmer = -1;
  int i, j;
  float f;
  long l;
  if (!cmd || !argv) return VS_ERR_ARG;
  vs_lane *par = &cmd->lane;
  vs_lane_defaults(par);
  cmd->dur = 1.0f;
  cmd->wav_arg = -1;

  if (argc < 2) return VS_USAGE; /* fg:128 */

  for (i = 1; i < argc && *argv[i] == '-'; i++) { /* fg:130 */
    j = i + 1;
    if (argc <= j) return VS_USAGE; /* fg:135 */
    switch (argv[i++][1]) {
      case 'o': case 'O': a_wav = i; break;
      case 'g': case 'G': a_Fg = i; break;
      case 'f': case 'F': a_F0 = i; break;
      case 'd': case 'D': a_dur = i; break;
      case 'c': case 'C': a_cq = i; break;
      case 'j': case 'J': a_jitter = i; break;
      case 'k': case 'K': a_K = i; break;
      case 'n': case 'N':
        par->DC = .25; /* fg:182: -n sets the DC flow at parse time */
        a_noise = i;
        break;
      case 'r': case 'R': a_fs = i; break;
      case 'a': case 'A': a_amp = i; break;
      case 'l': case 'L': a_DC = i; break;
      case 'z': case 'Z': a_Kvar = i; break;
      case 's': case 'S': a_Shimmer = i; break;
      default: return VS_USAGE; /* fg:212 */
    }
  }
  if ((i != argc && *argv[i] != 'i') || a_wav == -1) return VS_USAGE; /* fg:219 */

  /* initialization(), same order as fg:470-546 */
  if (a_dur != -1) {
    f = atof(argv[a_dur]);
    if (f >= 0.5) cmd->dur = f;
    else return VS_USAGE;
  }
  if (a_jitter != -1) {
    f = atof(argv[a_jitter]) / 100.0;
    if (f >= 0.0 && f <= 10.0) par->jitter = f;
    else return VS_USAGE;
    par->flags |= VS_FLAG_JITTER;
  }
  if (a_K != -1) {
    f = atof(argv[a_K]);
    if (f >= 0.50) par->K = f;
    else return VS_USAGE;
  }
  if (a_cq != -1) {
    f = atof(argv[a_cq]);
    if (f >= 0.0 && f <= 1.0) par->cq = f;
    else return VS_USAGE;
  }
  if (a_Fg != -1) {
    f = atof(argv[a_Fg]);
    if (f >= 50) par->Fg = f;
    else return VS_USAGE;
  }
  if (a_F0 != -1) {
    f = atof(argv[a_F0]);
    if ((f >= 50) && (f < par->Fg)) par->F0 = f;
    else return VS_USAGE;
  }
  if (a_noise != -1) {
    f = atof(argv[a_noise]);
    if (f >= 0.0 && f <= 50) {
      par->noise = pow(10, f / 10);
    } else return VS_USAGE;
    par->flags |= VS_FLAG_NOISE;
  }
  if (a_amp != -1) {
    int iv = atoi(argv[a_amp]);
    if (iv >= 0 && iv < 32767) par->amp = iv;
    else return VS_USAGE;
  }
  if (a_DC != -1) {
    f = atof(argv[a_DC]);
    if (f >= 0 && f <= 0.3) par->DC = f * par->amp;
    else return VS_USAGE;
  }
  if (a_Kvar != -1) {
    f = atof(argv[a_Kvar]);
    if (f >= 0 && f <= 1) par->Kvar = f;
    else return VS_USAGE;
  }
  if (a_fs != -1) {
    l = atol(argv[a_fs]);
    /* fg:537 accepts everything except 22050 */
    if ((l == 44100L) || (l != 22050L) || (l == 11025L)) par->fs = (int32_t)l;
    else return VS_USAGE;
  }
  if (a_Shimmer != -1) {
    f = atof(argv[a_Shimmer]);
    if (f >= 0 && f <= 100) par->shimmer = f / 100;
    else return VS_USAGE;
    par->flags |= VS_FLAG_SHIMMER;
  }
  cmd->wav_arg = a_wav;
  return VS_OK;
}

/* ------------------------------------------------------------------------------------------
 * vowel command line: vowel_new.c:116-192
 * ---------------------------------------------------------------------------------------- */
int vs_vowel_parse(int argc, char **argv, vs_vowel_cmd *cmd)
{
  int i, j;
  if (!cmd || !argv) return VS_ERR_ARG;
  cmd->gain = 10.0f;
  cmd->pre_emphasis = 1.0f;
  cmd->snr = 0.0f;
  cmd->vowel = 0;
  cmd->input_arg = cmd->output_arg = cmd->noise_arg = -1;
  int algorithm_arg = -1;

  if (argc < 2) return VS_USAGE; /* vw:116 */
  for (i = 1; i < argc && *argv[i] == '-'; i++) {
    if (argc <= i + 1) return VS_USAGE; /* vw:121 */
    switch (argv[i++][1]) {
      case 'p': case 'P':
        cmd->pre_emphasis = atof(argv[i]);
        if (cmd->pre_emphasis < 0.0 || cmd->pre_emphasis > 1.0) return VS_USAGE;
        break;
      case 'g': case 'G':
        cmd->gain = atof(argv[i]);
        if (cmd->gain < 1) return VS_USAGE;
        break;
      case 'i': case 'I': cmd->input_arg = i; break;
      case 'n': case 'N':
        cmd->noise_arg = i;
        cmd->snr = atof(argv[i]);
        if (cmd->snr <= 0) return VS_USAGE;
        else cmd->snr = pow(10, cmd->snr / 10);
        break;
      case 'o': case 'O': cmd->output_arg = i; break;
      case 'v': case 'V':
        algorithm_arg = i;
        j = (int)argv[i][0];
        if (j == 'i' || j == 'a' || j == 'u' || j == 'I' || j == 'A' || j == 'U' || j == '1' ||
            j == '2' || j == '3' || j == '4' || j == '5' || j == '6' || j == '7') {
          cmd->vowel = j;
        } else return VS_USAGE;
        break;
      default: return VS_USAGE;
    }
  }
  if ((i != argc && *argv[i] != 'i') || cmd->input_arg == -1 || algorithm_arg == -1)
    return VS_USAGE; /* vw:191 */
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
