/*
 * vs_batch -- batch front-end over the C ABI: N utterances
 * described by the REFERENCE's own command lines, synthesised in one fused GPU launch per
 * distinct sample count, written as N .wav files.
 *
 *     vs_batch [--gpus N] manifest.txt
 *
 * --gpus N shards every launch over devices 0..N-1 (contiguous blocks of utterances, one host
 * thread and one PCIe link per device: vs_node_synth_rows).
 *
 * One utterance per manifest line:
 *
 *     [seed=N] <flowgen_shimmer arguments> | <vowel arguments>
 *
 * e.g.   seed=7 -o a1.wav -r 16000 -d 1 -j 1 -s 5.76 -n 20 | -v 1 -g 10
 *
 * The flowgen "-o" names the FINAL speech file (there is no intermediate flow file: the flow
 * never leaves the chip); the vowel side takes -v, -g, -p, -n (its -i/-o are implied).
 * Arguments are parsed by the same vs_flowgen_parse()/vs_vowel_parse() the drop-in programs
 * use, so ranges, defaults and quirks are the reference's.  Without "seed=" a line's Philox key
 * is VS_SEED (default: time) + line number.  Empty lines and lines starting with '#' are
 * skipped.  Exit code 0 on success, 1 on any error (nothing is exit()ed from the library).
 *
 * Output path (what replaces the fwrite()s of flowgen_shimmer.c:413-421 and vowel_new.c:327): the
 * files are written from the pinned staging blocks of vs_synth_rows() by the library's delivery
 * threads -- header, then payload -- while the device is synthesising the next chunk of the
 * batch and the DMA engines are moving the next blocks; there is no batch-sized host buffer.
 */
#include <ctype.h>

#include "cli_common.h"

#define MAX_TOK 64

typedef struct {
  vs_lane lane;
  float dur;
  uint64_t n_samples;
  char *path;
  int done;
} job;

static int split(char *s, char **tok, int max)
{
  int n = 0;
  while (*s && n < max - 1) {
    while (*s && isspace((unsigned char)*s)) *s++ = 0;
    if (!*s) break;
    tok[n++] = s;
    while (*s && !isspace((unsigned char)*s)) s++;
  }
  tok[n] = NULL;
  return n;
}

/* one launch's worth of files: rows of the launch -> jobs */
typedef struct {
  job *jobs;
  const size_t *index;
  uint64_t ns;
  int header_bytes;
  volatile int failed;
} sink;

static int cmp_job_path(const void *a, const void *b)
{
  return strcmp((*(const job *const *)a)->path, (*(const job *const *)b)->path);
}

/* vs_rows_cb: called from the delivery threads, concurrently for different blocks (distinct
 * files, so no locking) */
static int write_rows(void *user, size_t row0, size_t rows, const int16_t *pcm)
{
  sink *sk = (sink *)user;
  for (size_t r = 0; r < rows; r++) {
    job *j = &sk->jobs[sk->index[row0 + r]];
    unsigned char header[72];
    const int hbytes = vs_wav_header_write(header, sk->header_bytes, j->lane.fs, j->dur);
    FILE *f = fopen(j->path, "wb");
    if (!f || hbytes <= 0 || fwrite(header, (size_t)hbytes, 1, f) != 1 ||
        fwrite(pcm + r * (size_t)sk->ns, sizeof(int16_t), (size_t)sk->ns, f) != (size_t)sk->ns) {
      fprintf(stderr, "vs_batch: cannot write %s\n", j->path);
      if (f) fclose(f);
      sk->failed = 1;
      return 1;
    }
    if (fclose(f) != 0) { /* a write error may only surface here (a full disk): the file is not done */
      fprintf(stderr, "vs_batch: cannot write %s\n", j->path);
      sk->failed = 1;
      return 1;
    }
    j->done = 1;
  }
  return 0;
}

int main(int argc, char **argv)
{
  int gpus = 1;
  if (argc == 4 && strcmp(argv[1], "--gpus") == 0) {
    gpus = atoi(argv[2]);
    argv += 2;
    argc -= 2;
  }
  if (argc != 2 || gpus < 1 || gpus > 64) {
    fprintf(stderr, "usage: vs_batch [--gpus N] manifest.txt\n"
                    "  line: [seed=N] <flowgen_shimmer args incl. -o out.wav> | <vowel args: -v x [-g x] [-p x] [-n x]>\n");
    return 1;
  }
  FILE *mf = fopen(argv[1], "r");
  if (!mf) {
    fprintf(stderr, "vs_batch: cannot open %s\n", argv[1]);
    return 1;
  }
  size_t cap = 1024, n_jobs = 0;
  job *jobs = (job *)malloc(cap * sizeof(job));
  char line[4096];
  const uint64_t seed0 = vs_cli_seed();
  long lineno = 0;
  while (jobs && fgets(line, sizeof(line), mf)) {
    lineno++;
    char *p = line;
    while (*p && isspace((unsigned char)*p)) p++;
    if (!*p || *p == '#') continue;
    char *bar = strchr(p, '|');
    if (!bar) {
      fprintf(stderr, "vs_batch: line %ld: missing '|' between the flowgen and the vowel arguments\n", lineno);
      return 1;
    }
    *bar = 0;
    uint64_t seed = seed0 + (uint64_t)lineno;
    char *ftok[MAX_TOK + 1], *vtok[MAX_TOK + 4];
    ftok[0] = (char *)"flowgen_shimmer";
    int nf = 1 + split(p, ftok + 1, MAX_TOK);
    if (nf > 1 && strncmp(ftok[1], "seed=", 5) == 0) {
      seed = (uint64_t)strtoull(ftok[1] + 5, NULL, 0);
      memmove(ftok + 1, ftok + 2, (size_t)(nf - 1) * sizeof(char *));
      nf--;
    }
    vtok[0] = (char *)"vowel";
    vtok[1] = (char *)"-i";
    vtok[2] = (char *)"-";
    int nv = 3 + split(bar + 1, vtok + 3, MAX_TOK);
    vs_flowgen_cmd fc;
    vs_vowel_cmd vc;
    int rc = vs_flowgen_parse(nf, ftok, &fc);
    if (rc == VS_OK) rc = vs_vowel_parse(nv, vtok, &vc);
    if (rc != VS_OK) {
      fprintf(stderr, "vs_batch: line %ld: %s\n", lineno,
              rc == VS_USAGE ? "arguments the reference would answer with usage()" : vs_strerror(rc));
      return 1;
    }
    if (n_jobs == cap) {
      cap *= 2;
      jobs = (job *)realloc(jobs, cap * sizeof(job));
      if (!jobs) break;
    }
    job *j = &jobs[n_jobs];
    j->lane = fc.lane;
    j->lane.gain = vc.gain;
    j->lane.pre_emphasis = vc.pre_emphasis;
    j->lane.vowel = vc.vowel;
    j->lane.out_snr = (vc.noise_arg != -1) ? vc.snr : 0.0f;
    j->lane.seed = seed;
    j->lane.out_seed = seed;
    j->dur = fc.dur;
    rc = vs_num_samples(fc.lane.fs, fc.dur, &j->n_samples);
    j->path = strdup(ftok[fc.wav_arg]);
    j->done = 0;
    if (rc == VS_OK) rc = vs_lane_validate(&j->lane);
    if (rc != VS_OK) {
      fprintf(stderr, "vs_batch: line %ld: %s\n", lineno, vs_strerror(rc));
      return 1;
    }
    n_jobs++;
  }
  fclose(mf);
  if (!jobs) {
    fprintf(stderr, "vs_batch: out of memory\n");
    return 1;
  }
  /* two lines naming one output file would be written by two delivery threads at once */
  {
    const job **by_path = (const job **)malloc((n_jobs ? n_jobs : 1) * sizeof(job *));
    if (!by_path) return 1;
    for (size_t k = 0; k < n_jobs; k++) by_path[k] = &jobs[k];
    qsort(by_path, n_jobs, sizeof(job *), cmp_job_path);
    for (size_t k = 1; k < n_jobs; k++) {
      if (strcmp(by_path[k - 1]->path, by_path[k]->path) == 0) {
        fprintf(stderr, "vs_batch: %s is the output of more than one line\n", by_path[k]->path);
        return 1;
      }
    }
    free(by_path);
  }
  if (n_jobs == 0) {
    fprintf(stderr, "vs_batch: empty manifest\n");
    return 1;
  }

  vs_ctx *ctx = NULL;
  vs_node *node = NULL;
  if (gpus == 1) {
    if (vs_cli_open_ctx(&ctx) != VS_OK) return 1;
  } else {
    /* devices 0..N-1, or the ordinals listed in VS_DEVICES ("0,0,0,0": logical shards of one
     * device, how the sharded path is exercised on a one-GPU box) */
    int devs[64];
    for (int d = 0; d < gpus; d++) devs[d] = d;
    const char *list = getenv("VS_DEVICES");
    for (int d = 0; list && *list && d < gpus; d++) {
      devs[d] = atoi(list);
      const char *c = strchr(list, ',');
      list = c ? c + 1 : NULL;
    }
    int rc = vs_node_create(devs, gpus, &node);
    if (rc != VS_OK) {
      fprintf(stderr, "vs_batch: cannot open %d GPU devices: %s\n", gpus, vs_strerror(rc));
      return 1;
    }
    const char *a = getenv("VS_ARITH");
    if (a && strcmp(a, "fma") == 0) vs_node_set_arith(node, VS_ARITH_FMA);
  }
  const int hb = vs_cli_header_bytes();
  size_t remaining = n_jobs, launches = 0;
  vs_lane *lanes = (vs_lane *)malloc(n_jobs * sizeof(vs_lane));
  size_t *index = (size_t *)malloc(n_jobs * sizeof(size_t));
  while (remaining && lanes && index) {
    /* one launch per distinct sample count */
    uint64_t ns = 0;
    size_t m = 0;
    for (size_t k = 0; k < n_jobs; k++) {
      if (jobs[k].done) continue;
      if (m == 0) ns = jobs[k].n_samples;
      if (jobs[k].n_samples == ns) {
        lanes[m] = jobs[k].lane;
        index[m++] = k;
      }
    }
    sink sk;
    sk.jobs = jobs;
    sk.index = index;
    sk.ns = ns;
    sk.header_bytes = hb;
    sk.failed = 0;
    int rc = node ? vs_node_synth_rows(node, lanes, m, (size_t)ns, write_rows, &sk)
                  : vs_synth_rows(ctx, lanes, m, (size_t)ns, write_rows, &sk);
    if (rc != VS_OK || sk.failed) {
      if (!sk.failed) {
        /* with --gpus the HIP error lives in the failing shard's context */
        int hip = ctx ? vs_ctx_last_hip_error(ctx) : 0, shard = -1;
        for (int d = 0; node && d < gpus && hip == 0; d++) {
          vs_ctx *c = NULL;
          if (vs_node_ctx(node, d, &c) == VS_OK && vs_ctx_last_hip_error(c) != 0) {
            hip = vs_ctx_last_hip_error(c);
            shard = d;
          }
        }
        if (shard >= 0) fprintf(stderr, "vs_batch: synthesis failed: %s (hip %d on shard %d)\n", vs_strerror(rc), hip, shard);
        else fprintf(stderr, "vs_batch: synthesis failed: %s (hip %d)\n", vs_strerror(rc), hip);
      }
      return 1;
    }
    for (size_t q = 0; q < m; q++) {
      if (!jobs[index[q]].done) {
        fprintf(stderr, "vs_batch: %s was not delivered\n", jobs[index[q]].path);
        return 1;
      }
    }
    remaining -= m;
    launches++;
  }
  printf("vs_batch: %zu utterances in %zu launch(es)\n", n_jobs, launches);
  if (ctx) vs_ctx_destroy(ctx);
  if (node) vs_node_destroy(node);
  return 0;
}
